# usage: bash scripts/trace_step.sh <tag> [bench args]  -> gpurun_out/trace_<tag>.jsonl (per-step launch durations)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_${tag}_bench.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/trace_step.py $f > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.jsonl
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$tag
head -40 $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.jsonl
