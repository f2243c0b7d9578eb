"""GPU-busy time of ONE rank's period from a rocprofv3 kernel trace of scripts/strip_proxy.py (the last 100 periods of the
trace = the interior rank stepped alone): sum of kernel durations and number of launches per period, per kernel.
usage: python scripts/strip_proxy_busy.py <kernel_trace.csv>"""
import collections, csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_dist_classify" in r["Kernel_Name"]]
a, b = idx[-101], idx[-1]
tot = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:40]
    tot[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[n][1] += 1
out = {"busy_us_per_period": round(sum(v[0] for v in tot.values()) / 100 / 1e3, 1),
       "launches_per_period": round(sum(v[1] for v in tot.values()) / 100, 1),
       "kernels_us_per_period": {n: round(d / 100 / 1e3, 1) for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]},
       "kernels_launches_per_period": {n: round(c / 100, 2) for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]}}
print(json.dumps(out))
