#!/bin/bash
# same-box A/B of library builds for the force-table kernel: scripts/ab_forces_table.sh lib1.so lib2.so ...
# (each twice, interleaved; k_forces_table's average from rocprofv3 --kernel-trace --stats, the step time from bench.py)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)_$rep
    rm -rf gpurun_out/ft_$tag
    LBMDEM_HIP_LIBRARY=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ft_$tag -o ft -- python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ft_$tag.json 2>gpurun_out/ft_$tag.err
    f=$(find gpurun_out/ft_$tag -name "*kernel_stats*" | head -1)
    python - "$f" "$lib" gpurun_out/ft_$tag.json <<'PY'
import csv,sys,json
rows=list(csv.reader(open(sys.argv[1])))[1:]
out={}
for r in rows:
    for k in ("k_forces_table","k_forces_gather_queue","k_cs_march","k_dem_chain"):
        if k in r[0]: out[k]=(int(r[1]),round(float(r[3])/1e3,2),round(float(r[5])/1e3,1))
try: ms=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])["ms_per_step"]
except Exception as e: ms=None
print(sys.argv[2],"ms/step(traced)",ms,out,flush=True)
PY
    rm -rf gpurun_out/ft_$tag
  done
done
