#!/bin/bash
# the launches of the coupled steps of a bench run in order, with the idle time before each (rocprofv3 kernel trace):
# scripts/trace_gaps.sh [bench args] -> per kernel: mean duration and mean gap since the previous kernel's end, over the timed steps
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/tg; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tg -o tg -- python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline "$@" > gpurun_out/tg.json 2>gpurun_out/tg.err
f=$(find gpurun_out/tg -name "*kernel_trace*" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
def short(n):
    for k in ("k_cs_march","k_forces_table","k_forces_gather_queue","k_dem_chain","k_verlet_scan","k_cell_count","k_cell_scatter","k_tile_halo","k_obst","scan"):
        if k in n: return k
    return n[:30]
# steps = from one k_cs_march to the next; keep the steps with exactly 4 launches (no rebuild) in the last third of the run
idx=[i for i,r in enumerate(rows) if "k_cs_march" in r[2]]
acc=collections.defaultdict(lambda:[0,0.0,0.0]); spans=[]
for a,b in zip(idx[len(idx)//2:-1], idx[len(idx)//2+1:]):
    if b-a!=4: continue
    spans.append((rows[b][0]-rows[a][0])/1e3)
    for j in range(a,b):
        k=short(rows[j][2]); gap=(rows[j][0]-rows[j-1][1])/1e3
        acc[k][0]+=1; acc[k][1]+=(rows[j][1]-rows[j][0])/1e3; acc[k][2]+=gap
print("steps without a rebuild:",len(spans),"mean span us",round(sum(spans)/max(1,len(spans)),1))
for k,(n,d,g) in acc.items(): print(f"{k:26s} n={n:4d} mean us {d/n:8.1f}  mean gap before {g/n:6.2f}")
m=idx[len(idx)//3]
for j in range(m,m+14): print(short(rows[j][2]), "dur", round((rows[j][1]-rows[j][0])/1e3,1), "gap before", round((rows[j][0]-rows[j-1][1])/1e3,2))
PY
rm -rf gpurun_out/tg
