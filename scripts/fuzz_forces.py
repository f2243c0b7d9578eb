"""Randomised parity sweep of the whole coupled step, aimed at the link-sum table route of the hydrodynamic forces:
packings with different radius ranges (lattice lines per grain), touching / overlapping / wall-clipped grains, fast
grains (many nodes change owner per step), for a few dozen coupled steps each, every bit of f, obst, fhf and the grain
kinematics against the CPU oracle; reports how many grains took the table and how many the gather queue.
   python scripts/fuzz_forces.py [ncases] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as ge, samples
pkg = ge.load_package(); po = ge.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
bad = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    lx = int(rng.choice([384, 512, 640])); ly = int(rng.choice([256, 320, 448]))
    rmin = float(rng.choice([0.3, 0.5, 0.7])); rmax = rmin + float(rng.choice([0.1, 0.4, 0.8]))
    overlap = float(rng.choice([4e-3, 0.05, 0.3]))           # up to 0.3 mm: reduced discs of neighbours overlap
    n = int(rng.integers(200, 1500))
    r, x, y = samples.row_packing(lx, ly, n, seed=seed0 + case, rmin=rmin, rmax=rmax, touch_prob=float(rng.uniform(0.2, 0.9)),
                                  max_overlap=overlap, margin=float(rng.choice([0.0, 0.05, 0.3])))
    if len(r) < 3:
        continue
    # some grains pushed into the walls so that the lattice-interior clamp clips their discs
    k_wall = rng.integers(0, len(r), 6)
    x[k_wall[:3]] = rng.choice([0.35, 0.1 * lx - 0.35], 3); y[k_wall[3:]] = rng.choice([0.35, 0.1 * ly - 0.35], 3)
    r, x1, x2 = samples.to_metres(r, x, y)
    try:
        sim = pkg.LbmDem(lx, ly, r, x1, x2); ora = po.Oracle(lx, ly, r, x1, x2)
    except pkg.LbmDemError as e:
        print(f"case {case}: skipped ({e})"); continue
    vs = float(rng.choice([0.02, 0.2, 1.0]))
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [vs, vs, 50 * vs]
    sim.kinematics = k; ora.set_kinematics(k)
    nsteps = int(rng.integers(2, 5)) * sim.cfg.npDEM + int(rng.integers(0, sim.cfg.npDEM))
    tab = gat = 0
    ok = True
    done = 0
    try:
        while done < nsteps and ok:
            step = min(sim.cfg.npDEM, nsteps - done)
            sim.renderScene(step); ora.steps(step); done += step
            a, g = sim.force_stats(); tab += a; gat += g
            ok = (np.array_equal(sim.fhf, ora.get_fhf()) and np.array_equal(sim.kinematics, ora.get_grains()[:, :9]))
        ok = ok and np.array_equal(sim.f, ora.get_f()) and np.array_equal(sim.obst, ora.get_obst())
    except pkg.LbmDemError as e:
        print(f"case {case}: library error after {done} sub-steps: {e}"); bad += 1; continue
    print(f"case {case}: {lx}x{ly}, {len(r)} grains r {rmin:.1f}-{rmax:.1f} mm, overlap <= {overlap} mm, v ~ {vs}, {nsteps} sub-steps: "
          f"{'bit-equal' if ok else 'DIFFERENT'}; grain-steps from table {tab}, gathered {gat}; act anomalies {ora.act_anomalies()}", flush=True)
    bad += 0 if ok else 1
    del sim, ora
print(f"{'FUZZ OK' if bad == 0 else 'FUZZ FAILED: %d cases' % bad} [{time.time() - t0:.0f} s]")
sys.exit(1 if bad else 0)
