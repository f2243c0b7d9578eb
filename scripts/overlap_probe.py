"""Does a DEM chain hide under the fused kernel? TIMING ONLY (two independent simulations of the bench workload on one GPU).
Handle A runs fluid steps (rasteriser + fused kernel + force table), handle B runs DEM sub-steps, each on its own stream:
  (1) A alone, N fluid steps              (2) B alone, N x npDEM sub-steps
  (3) both enqueued together (one host thread per handle)      -> if t3 ~ t1 the DEM chain is free under the fluid step.
Also the force-table kernel of B under the fused kernel of A (forces_fluid alone on B).
usage: python scripts/overlap_probe.py [steps]"""
import os, sys, json, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = bench.workload("default")
(r, x1, x2), _ = bench.make_sample(w)
lx, ly = w["lx"], w["ly"]
A = pkg.LbmDem(lx, ly, r, x1, x2)
B = pkg.LbmDem(lx, ly, r, x1, x2)
npdem = 12
for s in (A, B):
    s.initVerlet()
    s.renderScene(2 * npdem)   # warm: both have a rasterised map, tables, lists
    s.sync()

def timed(fa, fb):
    torch.cuda.synchronize()
    A.sync(); B.sync()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=f) for f in (fa, fb) if f]
    for t in ths: t.start()
    for t in ths: t.join()
    A.sync(); B.sync()
    return (time.perf_counter() - t0) * 1e3 / N

def fluid():
    for _ in range(N): A.lbm_step()
def dem():
    for _ in range(N): B.run_dem(npdem - 1)   # 11 per round: stays clear of the Verlet rebuild cadence bookkeeping differences
def table():
    for _ in range(N): B.forces_fluid()
def cs_only():
    for _ in range(N): A.collision_streaming()

out = {"steps": N}
for rep in range(2):
    out[f"fluid_alone_ms_{rep}"] = round(timed(fluid, None), 4)
    out[f"dem_alone_ms_{rep}"] = round(timed(dem, None), 4)
    out[f"fluid_and_dem_ms_{rep}"] = round(timed(fluid, dem), 4)
    out[f"table_alone_ms_{rep}"] = round(timed(table, None), 4)
    out[f"fluid_and_table_ms_{rep}"] = round(timed(fluid, table), 4)
    out[f"cs_alone_ms_{rep}"] = round(timed(cs_only, None), 4)
    out[f"cs_and_dem_ms_{rep}"] = round(timed(cs_only, dem), 4)
print(json.dumps(out))
