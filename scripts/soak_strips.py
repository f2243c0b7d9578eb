"""One-off soak of the strip path: several strips on ONE GPU stepped in lock-step (tests/strip_backends.py), fast
grains so that some cross the cuts (ownership of rasterisation and forces changes hands), thousands of sub-steps,
bit-equality with the CPU oracle.   python scripts/soak_strips.py [world lx ly ngrains nsteps vscale]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge, samples
from strip_backends import LoopbackComm, lockstep_render
pkg = ge.load_package(); po = ge.load_oracle(); strips = pkg.strips_module()
a = sys.argv[1:]
world, lx, ly, n, nsteps = (int(v) for v in (a[:5] if len(a) >= 5 else (3, 640, 384, 700, 3000)))
vscale = float(a[5]) if len(a) > 5 else 6.0
r, x, y = samples.row_packing(lx, ly, n, seed=17); r, x1, x2 = samples.to_metres(r, x, y)
cfg = pkg.derive(lx, ly, r); halo = strips.halo_rows(float(r.max()), cfg.dx)
rng = np.random.default_rng(8)
k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05 * vscale, 0.02 * vscale, 10.0]
runners = []
for rank, strip in enumerate(strips.partition(lx, world)):
    be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, halo, 0)
    be.sim.kinematics = k
    runners.append(strips.StripRunner(be, LoopbackComm(), rank, world))
ora = po.Oracle(lx, ly, r, x1, x2); ora.set_kinematics(k)
cuts = [s[0] for s in strips.partition(lx, world)][1:]
xc0 = (x1 - cfg.Mgx) / cfg.dx
done = 0; t0 = time.time()
for stop in range(500, nsteps + 1, 500):
    lockstep_render(runners, stop - done); ora.steps(stop - done); done = stop
    got = np.full((lx, ly, 9), np.nan)
    for R in runners: R.b.sim.download_f_into(got)
    g = ora.get_grains()
    eq = lambda u, v: np.array_equal(u, v, equal_nan=True)
    ok = eq(got, ora.get_f()) and all(eq(R.b.sim.kinematics, g[:, :9]) and eq(R.b.sim.fhf, ora.get_fhf()) for R in runners)
    if not np.isfinite(g[:, :9]).all(): print("  (the oracle's own state is no longer finite: the packing blew up)")
    xc = (g[:, 0] - cfg.Mgx) / cfg.dx
    crossed = int(sum(((xc0 < c) != (xc < c)).sum() for c in cuts))
    print(f"step {stop}: bit-equal {ok}; grains that changed owner so far: {crossed}; anomalies {ora.act_anomalies()} [{time.time()-t0:.0f} s]", flush=True)
    if not ok: sys.exit(1)
print("STRIP SOAK OK")
