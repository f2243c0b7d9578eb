# usage: bash scripts/prof_kernels.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ (kernel stats csv)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench.log | cut -c1-300
