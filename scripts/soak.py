"""One-off soak: HIP path vs the CPU oracle over thousands of DEM sub-steps (film step at 8000, diagnostics at
4000 and 8000, Verlet rebuild every 100) on a mid-size lattice; compares every bit at several checkpoints.
   python scripts/soak.py [lx ly ngrains nsteps]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, samples
pkg = ge.load_package(); po = ge.load_oracle()
lx, ly, n, nsteps = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (768, 640, 1800, 8400)))
vscale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
r, x, y = samples.row_packing(lx, ly, n, seed=99); r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(lx, ly, r, x1, x2); ora = po.Oracle(lx, ly, r, x1, x2, fast=False)
rng = np.random.default_rng(5)
k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.02 * vscale, 0.02 * vscale, 10.0 * vscale]
sim.kinematics = k; ora.set_kinematics(k)
done = 0; t0 = time.time()
for stop in (1200, 3999, 4000, 4001, 6000, 7999, 8000, 8001, 12000, 15999, 16000, 16001, nsteps):
    if stop > nsteps or stop <= done: continue
    sim.renderScene(stop - done); ora.steps(stop - done); done = stop
    ok = (np.array_equal(sim.kinematics, ora.get_grains()[:, :9]) and np.array_equal(sim.fhf, ora.get_fhf())
          and np.array_equal(sim.obst, ora.get_obst()) and np.array_equal(sim.f, ora.get_f()))
    extra = ""
    if stop % 4000 == 0:
        tg, to = sim.grain_table(), ora.get_grains()
        cols = [c for c in range(30) if c not in (18, 20)]  # fm, ifr are formed by write_DEM (main.c:388,409), which the oracle never calls
        bad = [c for c in cols if not np.array_equal(tg[:, c], to[:, c])]
        extra = " grain table: differing columns %s; nonzero slip/rw/fr/ice: %s" % (bad, [int((to[:, c] != 0).sum()) for c in (26, 27, 19, 25)])
        ok = ok and not bad
    print(f"step {stop}: bit-equal {ok}{extra}  anomalies {ora.act_anomalies()}  [{time.time()-t0:.0f} s]", flush=True)
    if not ok: sys.exit(1)
print("SOAK OK")
