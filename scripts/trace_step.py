"""Per-launch durations of one bench run, in launch order, from a rocprofv3 kernel trace (csv).
usage: python scripts/trace_step.py <kernel_trace.csv> [first_step last_step] -> one line per fluid step:
       rasteriser, fused kernel, force table, gather queue, the DEM sub-steps (sum, min, max), gaps between launches."""
import csv, sys, json

def short(n):
    for k in ("k_cs_march", "k_forces_table", "k_forces_gather_queue", "k_obst_paint", "k_obst_fill", "k_dem_entries",
              "k_verlet_scan", "k_cell_count", "k_cell_scatter"):
        if k in n:
            return k
    return "other"

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
steps, cur = [], None
for s, e, k in rows:
    if k == "k_obst_paint":
        cur = {"t0": s, "k": []}
        steps.append(cur)
    if cur is not None:
        cur["k"].append((k, s, e))
out = []
for i, st in enumerate(steps[:-1]):
    d = {}
    busy = 0
    for k, s, e in st["k"]:
        d.setdefault(k, []).append((e - s) / 1e3)
        busy += e - s
    span = (steps[i + 1]["t0"] - st["t0"]) / 1e3
    dem = d.get("k_dem_entries", [])
    out.append({"step": i, "span_us": round(span, 1), "idle_us": round(span - busy / 1e3, 1),
                "paint": round(sum(d.get("k_obst_paint", [0])), 1), "fused": round(sum(d.get("k_cs_march", [0])), 1),
                "table": round(sum(d.get("k_forces_table", [0])), 1), "queue": round(sum(d.get("k_forces_gather_queue", [0])), 1),
                "dem_sum": round(sum(dem), 1), "dem_n": len(dem), "dem_min": round(min(dem), 1) if dem else 0,
                "dem_max": round(max(dem), 1) if dem else 0, "n_launch": len(st["k"]),
                "other": round(sum(sum(v) for k, v in d.items() if k in ("other", "k_verlet_scan", "k_cell_count", "k_cell_scatter", "k_obst_fill")), 1)})
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(out))
for o in out[lo:hi]:
    print(json.dumps(o))
