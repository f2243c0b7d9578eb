"""Where a wavefront of k_cs_march spends its cycles: experiment builds with -DMARCH_TIMING (s_memtime between the phases
of an iteration, summed over all wavefronts). LBMDEM_HIP_LIBRARY must point at such a build:
  make -C 2d-lbm-dem_amd/csrc AB=1 ABFLAGS=-DMARCH_TIMING=1   (=2: plus a full vmcnt(0) wait at the top of the iteration)
The read of s_memtime waits for lgkmcnt(0), so the phases are slightly serialised against the LDS traffic."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import bench
import __graft_entry__ as ge

pkg = ge.load_package()
w = bench.workload("metric")
(r, x1, x2), _ = bench.make_sample(w)
sim = pkg.LbmDem(w["lx"], w["ly"], r, x1, x2)
npdem = sim.cfg.npDEM
sim.renderScene(5 * npdem); sim.sync()
L = pkg.load_library()
fn = L.lbmdem_ab_march_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
assert fn(out) == 0          # clears
steps = 20
sim.profile_enable(True)
sim.renderScene(steps * npdem); sim.sync()
kernel_ms, launches = sim.profile_read()
assert fn(out) == 0
t = [int(v) for v in out]
names = ["wait for the populations of row x+1", "reinit + collide of row x+1", "gathers + issue of the row prefetch",
         "act of row x+1 (lane masks)", "DPP shifts + classification (edge rows: + their stores)",
         "after the links: the row's nine stores (results merged in)", "ring put + rotate", "loop head", "", "",
         "links compacted into LDS slots", "links evaluated (LDS reads, bounce-back, result, table store)"]
tot = t[8]
phase = t[:8] + [0, 0] + t[10:12]
res = {"fused_kernel_ms": round(kernel_ms, 4), "waves": t[9] // steps,
       "cycles_per_wave": round(tot / max(t[9], 1)), "sum_of_phases_frac": round(sum(phase) / tot, 4),
       "phases_frac_of_wave_time": {n: round(v / tot, 4) for n, v in zip(names, phase) if n}}
res["phases_frac_of_wave_time"] = {n: v for n, v in res["phases_frac_of_wave_time"].items() if v}
print(json.dumps(res))
