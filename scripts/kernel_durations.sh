#!/bin/bash
# per-launch durations of one kernel, in launch order: scripts/kernel_durations.sh <kernel substring> [bench args]
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
K=$1; shift
rm -rf gpurun_out/kd; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kd -o kd -- python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline "$@" > gpurun_out/kd.json 2>gpurun_out/kd.err
f=$(find gpurun_out/kd -name "*kernel_trace*" | head -1)
python - "$f" "$K" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[]
prev_end=None
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if sys.argv[2] in r["Kernel_Name"]:
        seq.append(round(d,1))
print(len(seq),"launches; last 120 (us):")
print(seq[-120:])
PY
rm -rf gpurun_out/kd
