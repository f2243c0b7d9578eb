#!/bin/bash
# same-box A/B of bench.py flag sets, each twice, interleaved: scripts/ab_bench_flags.sh "--obst-update 1" "--obst-update 0" ...
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
mkdir -p gpurun_out
for rep in 1 2; do
  for fl in "$@"; do
    python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline $fl > gpurun_out/ab_flags.json 2>gpurun_out/ab_flags.err
    python -c "
import json;d=json.load(open('gpurun_out/ab_flags.json'));print('[$fl]', 'ms/step', d['ms_per_step'], 'unsettled', d['ms_per_step_unsettled'], 'dem_only/s', d['dem_only_steps_per_s'], 'fused ms', d['collide_stream_kernel_ms'], 'lbm only ms', d['lbm_step_only_ms'], 'copy', d['hbm_copy_gbs'])"
  done
done
