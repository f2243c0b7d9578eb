"""Which way the discs go in the rasterisation at the end of a run of sub-steps (experiment build, LBMDEM_HIP_LIBRARY =
liblbmdem_hip_ab.so): still (no node can have changed sides: skipped), ring scan, box scan (first picture), partners near
(compare-and-swap hand-over). The bench packing, 60 coupled steps. The counters come out on stderr."""
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
r, x, y = samples.row_packing(4096, 4096, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(4096, 4096, r, x1, x2)
L = sim._L; L.lbmdem_debug_chain_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros((782, 16), dtype=np.int64)
L.lbmdem_debug_chain_times(sim._h, buf.ctypes.data, 782)
sim.renderScene(12 * 60); sim.sync()
L.lbmdem_debug_chain_times(sim._h, buf.ctypes.data, 782)
print("obst map (updated in place, cleared and repainted):", sim.obst_stats(), "; rasterisations by runs:", sim.dem_chain_paints())
