"""DEM-only sub-step rate of the bench packing (run_dem, list rebuilds included) with the library LBMDEM_HIP_LIBRARY points at."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
r, x, y = samples.row_packing(4096, 4096, n, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(4096, 4096, r, x1, x2)
if len(sys.argv) > 2: sim.set_dem_chain(int(sys.argv[2]))
sim.run_dem(200); sim.sync(); t0 = time.perf_counter(); sim.run_dem(2400); sim.sync()
print(round(2400 / (time.perf_counter() - t0)), "sub-steps/s", sim.dem_chain_stats())
