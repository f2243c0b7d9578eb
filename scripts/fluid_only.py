import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
lx = ly = 4096
for n, label in ((1, "1 grain"),):
    sim = pkg.LbmDem(lx, ly, [0.5e-3], [2e-3], [2e-3])
    for _ in range(3): sim.lbm_step()
    sim.sync(); sim.profile_enable(True)
    for _ in range(20): sim.lbm_step()
    ms, cnt = sim.profile_read()
    print(f"variant {os.environ.get('LBMDEM_CS_VARIANT','default')}: {label}: k_collide_stream {ms:.4f} ms  -> {148*lx*ly/ms/1e6:.1f} GB/s  frac {148*lx*ly/ms/1e6/8000:.3f}")
