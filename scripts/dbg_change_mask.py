import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import samples, __graft_entry__ as ge
pkg = ge.load_package()
lx, ly = 320, 250
r, x, y = samples.row_packing(lx, ly, 230, seed=3)
r, x1, x2 = samples.to_metres(r, x, y)
a = pkg.LbmDem(lx, ly, r, x1, x2); a.set_change_mask(2)
b = pkg.LbmDem(lx, ly, r, x1, x2); b.set_change_mask(0)
rng = np.random.default_rng(11)
k = a.kinematics; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * (1.5, 1.5, 80.0)
a.kinematics = k; b.kinematics = k
n = a.cfg.npDEM
dx = a.cfg.dx
for s in range(60):
    ka = a.kinematics.copy()
    a.renderScene(n); b.renderScene(n)
    d = (a.f != b.f).any(axis=2)
    used, hidden = a.change_mask_stats()
    if d.any() or hidden:
        xs, ys = np.nonzero(d)
        print("step", a.nbsteps, "used", used, "hidden bits", hidden & 0xFFFFFFFF, "output mismatches", hidden >> 32, "nodes", list(zip(xs.tolist(), ys.tolist()))[:10], "maps equal", np.array_equal(a.obst, b.obst))
        kb = a.kinematics
        for (px, py) in list(zip(xs.tolist(), ys.tolist()))[:3]:
            cx = kb[:, 0] / dx; cy = kb[:, 1] / dx
            j = np.argsort((cx - px) ** 2 + (cy - py) ** 2)[:3]
            for g in j:
                print("  node", px, py, "grain", g, "centre now", cx[g], cy[g], "before", ka[g, 0] / dx, ka[g, 1] / dx, "r", r[g] / dx, "v", kb[g, 3:5])
        break
else:
    print("no difference in 60 steps; stats", a.change_mask_stats(), a.obst_stats())
