"""When and where the rows of k_cs_march are slow: an experiment build with -DMARCH_TRACE
  make -C 2d-lbm-dem_amd/csrc AB=1 ABTAG=_trace ABFLAGS=-DMARCH_TRACE
logs the constant-rate clock (100 MHz, common to all XCDs) at the head of every row iteration of every wavefront of ONE launch.
  LBMDEM_HIP_LIBRARY=.../liblbmdem_hip_ab_trace.so [LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=138 | LBMDEM_TAPER=r16,r8] python scripts/march_trace.py [out.npz]
Prints one JSON line: how the launch's wave slots (256 CUs x 4 SIMDs x 2) were used, iteration time by row index, by the moment
of the launch, by XCD, by window (lattice column band)."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch
import bench
import __graft_entry__ as ge

pkg = ge.load_package()
w = bench.workload("metric")
(r, x1, x2), _ = bench.make_sample(w)
sim = pkg.LbmDem(w["lx"], w["ly"], r, x1, x2)
npdem = sim.cfg.npDEM
sim.renderScene(5 * npdem); sim.sync()
L = pkg.load_library()
fn = L.lbmdem_ab_march_trace
fn.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
rows = int(os.environ.get("LBMDEM_CS_ROWS", "32"))
nstrips = (w["ly"] + 59) // 60   # MARCH_WW of lbm_fused.hip
cap = nstrips * (w["lx"] // 8 + 8)              # room for any mix of segments of >= 8 rows
stride = rows + 8
buf = torch.zeros(cap * stride, dtype=torch.int32, device="cuda")
assert fn(buf.data_ptr(), cap, stride) == 0
sim.profile_enable(True)
sim.renderScene(3 * npdem); sim.sync()
kernel_ms, launches = sim.profile_read()
assert fn(None, 0, 0) == 0
t = buf.cpu().numpy().view(np.uint32).reshape(cap, stride).astype(np.int64)
t = t[t[:, 3] > 0]
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], trace=t.astype(np.uint32), nstrips=nstrips, rows=rows)
W = len(t)
n = t[:, 3]
rel = ((t[:, 0] - t[:, 0].min()) & 0xFFFFFFFF) * 0.01          # us since the first wave started
stamps = t[:, 4:4 + rows + 1] * 0.01
k = np.arange(rows)[None, :]
valid = k < n[:, None]                                           # iteration k of wave w exists
dt = np.where(valid, np.diff(stamps, axis=1), np.nan)
life = stamps[np.arange(W), n]
end = rel + life
span = float(end.max())
xcc = (t[:, 1] >> 16) & 15
hw = t[:, 1] & 0xFFFF
strip = t[:, 1] >> 20
slot = (xcc << 16) | hw
res = {"rows_per_wave": {int(v): int((n == v).sum()) for v in np.unique(n)}, "waves": W, "fused_kernel_ms": round(kernel_ms, 4),
       "traced_span_us": round(span, 1), "wave_slots_seen": int(len(np.unique(slot)))}
# slot-time accounting
gaps = 0.0; tail = 0.0; head = 0.0
for s in np.unique(slot):
    i = np.where(slot == s)[0]; o = i[np.argsort(rel[i])]
    head += rel[o[0]]; tail += span - end[o[-1]]; gaps += float((rel[o[1:]] - end[o[:-1]]).sum())
tot = len(np.unique(slot)) * span
res["slot_time"] = {"busy": round(float(life.sum()) / tot, 4), "between_waves": round(gaps / tot, 4), "tail": round(tail / tot, 4),
                    "before_first_wave": round(head / tot, 4)}
res["wave_lifetime_us"] = {int(v): round(float(life[n == v].mean()), 1) for v in np.unique(n)}
res["iteration_us"] = {"mean": round(float(np.nanmean(dt)), 3), "p10": round(float(np.nanpercentile(dt, 10)), 2),
                       "p50": round(float(np.nanpercentile(dt, 50)), 2), "p90": round(float(np.nanpercentile(dt, 90)), 2)}
g = max(1, rows // 16)
full = n == rows
res["iteration_us_by_row_index (groups of %d, waves of %d rows)" % (g, rows)] = [round(float(np.nanmean(dt[full][:, j:j + g])), 3) for j in range(0, rows, g)]
t_abs = rel[:, None] + stamps[:, :rows]
dec = np.minimum((t_abs / span * 10).astype(int), 9)
res["iteration_us_by_launch_decile"] = [round(float(np.nanmean(dt[valid & (dec == d)])), 3) if (valid & (dec == d)).any() else None for d in range(10)]
res["waves_in_flight_by_launch_decile"] = [round(float(np.nansum(dt[valid & (dec == d)]) / (span / 10)), 0) for d in range(10)]
res["rows_per_us_by_launch_decile"] = [round(float((valid & (dec == d)).sum() / (span / 10)), 1) for d in range(10)]
res["iteration_us_by_xcd"] = {int(v): round(float(np.nanmean(dt[xcc == v])), 3) for v in np.unique(xcc)}
res["iteration_us_by_window (8 column bands, low y first)"] = [round(float(np.nanmean(dt[(strip * 8) // nstrips == b])), 3) for b in range(8)]
res["lifetime_us_of_first_and_last_window (full waves)"] = [round(float(life[full & (strip == 0)].mean()), 1), round(float(life[full & (strip == nstrips - 1)].mean()), 1)]
print(json.dumps(res))
