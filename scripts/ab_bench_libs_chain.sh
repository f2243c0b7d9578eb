#!/bin/bash
# same-box A/B of library builds on the default bench: scripts/ab_bench_libs_chain.sh lib1.so lib2.so ... (each twice, interleaved)
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    LBMDEM_HIP_LIBRARY=$PWD/$lib python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_lib.json 2>gpurun_out/ab_lib.err
    python -c "
import json;d=json.load(open('gpurun_out/ab_lib.json'));print('$lib', 'ms/step', d['ms_per_step'], 'dem_only/s', d['dem_only_steps_per_s'], 'fused ms', d['collide_stream_kernel_ms'], 'lbm only ms', d['lbm_step_only_ms'])"
  done
done
