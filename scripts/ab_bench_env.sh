#!/bin/bash
# same-box A/B of environment switches of the experiment build on the default bench: scripts/ab_bench_env.sh "VAR=1" "" ...
# ("" = no switch; each variant twice, interleaved). LIB=<library> to choose the build (default: liblbmdem_hip_ab.so)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
export LBMDEM_HIP_LIBRARY=$PWD/${LIB:-2d-lbm-dem_amd/liblbmdem_hip_ab.so}
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    env $v python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_env.json 2>gpurun_out/ab_env.err
    python -c "
import json;d=json.load(open('gpurun_out/ab_env.json'));print('[$v]', 'ms/step', d['ms_per_step'], 'fused ms', d['collide_stream_kernel_ms'], 'lbm only ms', d['lbm_step_only_ms'])"
  done
done
