"""Which route does the parity force kernel take? (table built by the fused kernel vs gather)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as ge, samples
pkg = ge.load_package()
for (lx, ly, n) in ((512, 512, 700), (4096, 4096, 50000)):
    r, x, y = samples.row_packing(lx, ly, n, seed=1234)
    r, x1, x2 = samples.to_metres(r, x, y)
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    sim.renderScene(3 * sim.cfg.npDEM)
    sim.lbm_step()
    print(lx, ly, n, "from_table, gathered =", sim.force_stats(), flush=True)
    sim.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        sim.lbm_step()
    sim.sync()
    print("lbm_step ms", 1e3 * (time.perf_counter() - t0) / 20)
