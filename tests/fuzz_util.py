"""Randomised coupled-step cases for the parity sweep (scripts/fuzz_forces.py) and its short version in the GPU suite
(tests/test_gpu_fuzz.py): packings with different radius ranges, touching / overlapping / wall-clipped grains, fast
grains, a few dozen sub-steps each."""
import numpy as np

import samples


def run_case(pkg, po, seed):
    """Returns (description, ok, grain-steps served by the table, by the gather queue, oracle's act anomalies); ok is
    None when the case could not be built."""
    rng = np.random.default_rng(seed)
    lx = int(rng.choice([384, 512, 640])); ly = int(rng.choice([256, 320, 448]))
    rmin = float(rng.choice([0.3, 0.5, 0.7])); rmax = rmin + float(rng.choice([0.1, 0.4, 0.8]))
    overlap = float(rng.choice([4e-3, 0.05, 0.3]))           # up to 0.3 mm: reduced discs of neighbours overlap
    n = int(rng.integers(200, 1500))
    r, x, y = samples.row_packing(lx, ly, n, seed=seed, rmin=rmin, rmax=rmax, touch_prob=float(rng.uniform(0.2, 0.9)),
                                  max_overlap=overlap, margin=float(rng.choice([0.0, 0.05, 0.3])))
    if len(r) < 3:
        return "too few grains", None, 0, 0, 0
    # some grains pushed into the walls so that the lattice-interior clamp clips their discs
    k_wall = rng.integers(0, len(r), 6)
    x[k_wall[:3]] = rng.choice([0.35, 0.1 * lx - 0.35], 3); y[k_wall[3:]] = rng.choice([0.35, 0.1 * ly - 0.35], 3)
    r, x1, x2 = samples.to_metres(r, x, y)
    sim = pkg.LbmDem(lx, ly, r, x1, x2); ora = po.Oracle(lx, ly, r, x1, x2)
    vs = float(rng.choice([0.02, 0.2, 1.0]))
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [vs, vs, 50 * vs]
    sim.kinematics = k; ora.set_kinematics(k)
    nsteps = int(rng.integers(2, 5)) * sim.cfg.npDEM + int(rng.integers(0, sim.cfg.npDEM))
    desc = (f"{lx}x{ly}, {len(r)} grains r {rmin:.1f}-{rmax:.1f} mm, overlap <= {overlap} mm, v ~ {vs}, "
            f"{nsteps} sub-steps")
    tab = gat = done = 0
    ok = True
    while done < nsteps and ok:
        step = min(sim.cfg.npDEM, nsteps - done)
        sim.renderScene(step); ora.steps(step); done += step
        a, g = sim.force_stats(); tab += a; gat += g
        ok = (np.array_equal(sim.fhf, ora.get_fhf()) and np.array_equal(sim.kinematics, ora.get_grains()[:, :9]))
    ok = ok and np.array_equal(sim.f, ora.get_f()) and np.array_equal(sim.obst, ora.get_obst())
    return desc, bool(ok), tab, gat, ora.act_anomalies()
