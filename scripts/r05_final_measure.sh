# Round-5 evidence in ONE gpurun call (one GPU): bench lines (headline, A/B legs of this round's changes), rocprofv3 kernel
# stats, PMC traffic of the fused kernel (tied to the source hash), the DEM chain's hand-off micro-benchmark and per-tile
# timeline, the C transport's strip period (8 processes on this GPU through tests/rccl_shim). Output: gpurun_out/r05/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05}; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
# what this round's changes are worth on this very GPU, interleaved: runs of sub-steps / one launch per sub-step; map updated in
# place by the runs / cleared and repainted / runs without the rasterisation
for rep in 1 2; do
  for fl in "" "--dem-chain 0 --obst-update 0" "--obst-update 0" "--dem-chain -1" "--change-mask 0"; do
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline $fl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'flags': '$fl', 'ms_per_step': d['ms_per_step'], 'unsettled': d['ms_per_step_unsettled'], 'dem_only_steps_per_s': d['dem_only_steps_per_s'], 'fused_ms': d['collide_stream_kernel_ms'], 'lbm_step_only_ms': d['lbm_step_only_ms'], 'hbm_copy_gbs': d['hbm_copy_gbs'], 'dem_chain': d['config']['dem_chain'], 'obst_map': d['config']['obst_map']}))" >> $O/ab_round5_changes.jsonl
  done
done
python bench.py --workload real50k --no-cpu-baseline > $O/bench_real50k.json 2>/dev/null
python bench.py --workload configs4 --no-cpu-baseline > $O/bench_configs4_one_gpu.json 2>/dev/null
python bench.py --precision f32 > $O/bench_f32.json 2>/dev/null
bash scripts/prof_kernels.sh r05_final > $O/prof_kernels.log 2>&1
cp $(find gpurun_out/prof_r05_final -name "*kernel_stats*" | head -1) $O/kernel_stats_final.csv 2>/dev/null
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/ 2>/dev/null
[ -x scripts/micro/handoff_xcd ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/micro/handoff_xcd scripts/micro/handoff_xcd.hip
timeout 120 scripts/micro/handoff_xcd > $O/handoff_xcd_micro.txt 2>&1
CT=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_ct.so
if [ -f $CT ]; then
  for n in 50000 12000 1000; do LBMDEM_HIP_LIBRARY=$CT python scripts/dem_chain_times.py $n 96 2>/dev/null >> $O/dem_chain_tiles.txt; echo >> $O/dem_chain_tiles.txt; done
fi
LBMDEM_RCCL_LIBRARY=$PWD/tests/rccl_shim/librccl.so.1 LBMDEM_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_two_ranks_one_gpu.json
bash scripts/strip_proxy_c.sh r05_final 8 4096 > $O/strip_proxy_c_4096.log 2>&1
bash scripts/strip_proxy_c.sh r05_final8k 8 8192 > $O/strip_proxy_c_8192.log 2>&1
LBMDEM_DEM_CHAIN=0 bash scripts/strip_proxy_c.sh r05_final_nochain 8 4096 > $O/strip_proxy_c_4096_one_launch_per_substep.log 2>&1
cp gpurun_out/proxyc_r05_final*.json $O/ 2>/dev/null
tail -1 $O/bench_final.json | cut -c1-400
