"""DEM sub-step variants: LBMDEM_DEM_VARIANT=0 (one thread per grain) / 1 (one lane per list entry),
LBMDEM_DEM_MULTI=0/1 (resident multi-sub-step kernel). Prints time per sub-step and checks equality."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
r, x, y = samples.row_packing(4096, 4096, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
os.environ["LBMDEM_DEM_MULTI"] = "0"
res = {}
for multi in ("0", "1"):
    os.environ["LBMDEM_DEM_MULTI"] = multi
    sim = pkg.LbmDem(4096, 4096, r, x1, x2)
    sim.lbm_step(); sim.run_dem(300)
    sim.sync(); t0 = time.perf_counter()
    reps = 200
    for _ in range(reps): sim.run_dem(12)
    sim.sync(); t1 = time.perf_counter()
    print(f"variant={os.environ.get('LBMDEM_DEM_VARIANT','1')} multi={multi} (batched {sim.dem_batched_substeps}): {1e6*(t1-t0)/reps/12:.2f} us per sub-step")
    res[multi] = (sim.kinematics, sim.grain_pressure)
    del sim
np.save("gpurun_out/dem_state_v%s.npy" % os.environ.get('LBMDEM_DEM_VARIANT','1'), res["0"][0])
print("multi == single:", all(np.array_equal(a, b) for a, b in zip(res["0"], res["1"])))
