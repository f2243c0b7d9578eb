import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
def run(label, r, x1, x2, lx=4096, ly=4096, n=2000):
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    sim.initVerlet()
    _, neigh, _ = sim.verlet()
    for _ in range(50): sim.dem_substep()
    sim.sync(); t0 = time.perf_counter()
    for _ in range(n): sim.dem_substep()
    sim.sync(); t1 = time.perf_counter()
    print(f"{label}: {len(r)} grains, {len(neigh)} pairs: {1e6*(t1-t0)/n:.2f} us per sub-step (wall, pipelined launches)")
r, x, y = samples.row_packing(4096, 4096, 50000); r, x1, x2 = samples.to_metres(r, x, y)
run("dense 50k", r, x1, x2)
# same grains spread on a coarse grid: no pairs within the Verlet distance
g = int(np.ceil(np.sqrt(len(r)))); xs = (np.arange(len(r)) % g) * 1.75e-3 + 1e-3; ys = (np.arange(len(r)) // g) * 1.75e-3 + 1e-3
run("no pairs 50k", np.full(len(r), 0.5e-3), xs, ys)
run("dense 6k", r[:6250], x1[:6250], x2[:6250])
run("dense 1k", r[:1000], x1[:1000], x2[:1000])
