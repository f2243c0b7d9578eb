import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["LBMDEM_HIP_LIBRARY"] = os.path.join(ROOT, "2d-lbm-dem_amd", "liblbmdem_hip_ab.so")
import __graft_entry__ as ge, samples
pkg = ge.load_package()
lx = ly = 4096
r, x, y = samples.row_packing(lx, ly, 50000, seed=1234)
r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(lx, ly, r, x1, x2)
sim.renderScene(3 * sim.cfg.npDEM); sim.lbm_step()
print(sim.force_stats())
q = (C.c_int * 64)()
L = pkg.load_library()
n = L.lbmdem_debug_gather_queue(sim._h, q, 64)
k = sim.kinematics
dx = sim.cfg.dx
for i in list(q)[:n]:
    xc, yc = k[i, 0] / dx, k[i, 1] / dx
    print(i, "xc", repr(xc), "yc", repr(yc), "r/dx", r[i] / dx, "rLB", 0.85 * r[i] / dx, "v", k[i, 3:6])
