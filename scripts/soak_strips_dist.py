"""One-off soak of the DISTRIBUTED-grain strip path: several strips on ONE GPU stepped in lock-step
(tests/strip_backends.py), grains every rank does not integrate poisoned with NaN after each sub-step, fast grains so
that many cross the cuts (migration), thousands of sub-steps, bit-equality of every owned row and owned grain with
the CPU oracle.   python scripts/soak_strips_dist.py [world lx ly ngrains nsteps vscale]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge, samples
from strip_backends import LoopbackComm, lockstep_render_dist
pkg = ge.load_package(); po = ge.load_oracle(); strips = pkg.strips_module()
a = sys.argv[1:]
world, lx, ly, n, nsteps = (int(v) for v in (a[:5] if len(a) >= 5 else (4, 1280, 256, 1200, 6000)))
vscale = float(a[5]) if len(a) > 5 else 6.0
r, x, y = samples.row_packing(lx, ly, n, seed=17); r, x1, x2 = samples.to_metres(r, x, y)
cfg = pkg.derive(lx, ly, r)
margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
parts = strips.partition(lx, world)
assert min(b - a_ for a_, b in parts) >= margin, (parts, margin)
rng = np.random.default_rng(8)
k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05 * vscale, 0.02 * vscale, 10.0]
runners = []
for rank, strip in enumerate(parts):
    be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin, poison=True)
    be.sim.kinematics = k
    runners.append(strips.DistStripRunner(be, LoopbackComm(), rank, world))
ora = po.Oracle(lx, ly, r, x1, x2); ora.set_kinematics(k)
cuts = [s[0] for s in parts][1:]
xc0 = (x1 - cfg.Mgx) / cfg.dx
done = 0; t0 = time.time()
print(f"{world} strips of {parts[0][1] - parts[0][0]} rows, margin {margin}, {len(r)} grains, {lx}x{ly}", flush=True)
for stop in range(500, nsteps + 1, 500):
    lockstep_render_dist(runners, stop - done); ora.steps(stop - done); done = stop
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.sync(); R.b.sim.download_f_into(got)
    g = ora.get_grains()[:, :9]; fh = ora.get_fhf()
    xc = (g[:, 0] - cfg.Mgx) / cfg.dx
    ok = np.array_equal(got, ora.get_f())
    for R, (lo, hi) in zip(runners, parts):
        own = ((lo == 0) | (xc >= lo)) & ((hi == lx) | (xc < hi))
        ok = ok and np.array_equal(R.b.sim.kinematics[own], g[own]) and np.array_equal(R.b.sim.fhf[own], fh[own])
    crossed = int(sum(((xc0 < c) != (xc < c)).sum() for c in cuts))
    print(f"step {stop}: bit-equal {ok}; grains that changed owner so far: {crossed}; anomalies {ora.act_anomalies()} [{time.time()-t0:.0f} s]", flush=True)
    if not ok: sys.exit(1)
print("DISTRIBUTED STRIP SOAK OK")
