# whole-step A/B of several libraries inside one gpurun call (bench.py wall clock, no profiler), each lib twice:
# bash scripts/ab_bench.sh <lib1> <lib2> ...
for rep in 1 2; do
  for lib in "$@"; do
    LBMDEM_HIP_LIBRARY=$lib python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], d['ms_per_step'], 'fused', d['collide_stream_kernel_ms'], 'lbm_only', d['lbm_step_only_ms'])"
  done
done
