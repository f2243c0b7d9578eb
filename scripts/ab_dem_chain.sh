#!/bin/bash
# A/B on one GPU box: the default bench with runs of DEM sub-steps in one launch (k_dem_chain) and with one launch per
# sub-step, interleaved. Prints ms per coupled step, DEM-only sub-steps/s, fused kernel ms, fluid-only ms.
mkdir -p gpurun_out
for c in ${CHAINS:-128 0 128 0}; do
  python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --dem-chain $c > gpurun_out/bench_chain_$c.json 2>gpurun_out/bench_chain_$c.err
  python -c "
import json;d=json.load(open('gpurun_out/bench_chain_$c.json'));print('chain', $c, 'ms/step', d['ms_per_step'], 'dem_only/s', d['dem_only_steps_per_s'], 'fused ms', d['collide_stream_kernel_ms'], 'lbm only ms', d['lbm_step_only_ms'], d['config']['dem_chain'])"
done
