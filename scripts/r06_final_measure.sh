# Round-6 evidence in ONE gpurun call (one GPU). Output: gpurun_out/${TAG:-r06}/ (copied to profiles/r06_*).
# Needs, next to liblbmdem_hip.so: liblbmdem_hip_r5.so (the round-5 product library, for the A/B), the per-phase timer builds
# liblbmdem_hip_ab_t<k>.so (make AB=1 ABTAG=_t<k> ABFLAGS=-DMARCH_TIMING=<k>, k = 0 1 2 3 4 10 11 5 6 7) and the ablation builds
# liblbmdem_hip_ab_{full,NOTAB,NOPASS,NOEVALPASS,NOCOLLIDE,NOREINIT}.so (make AB=1 ABTAG=_X ABFLAGS=-DMARCH_ABL_X; full: none).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r06}; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
for k in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/driver_form_three_runs.jsonl; done
python bench.py --workload real50k --no-cpu-baseline --real-steps 0 > $O/bench_real50k.json 2>/dev/null
python bench.py --workload configs4 --no-cpu-baseline > $O/bench_configs4_one_gpu.json 2>/dev/null
python bench.py --precision f32 --real-steps 0 > $O/bench_f32.json 2>/dev/null
bash scripts/prof_kernels.sh r06_final > $O/prof_kernels.log 2>&1
cp $(find gpurun_out/prof_r06_final -name "*kernel_stats*" | head -1) $O/kernel_stats_final.csv 2>/dev/null
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/ 2>/dev/null
bash scripts/sq_counters.sh r06_final > /dev/null 2>&1; cp gpurun_out/r06_final_sq_counters.txt $O/sq_counters.txt 2>/dev/null
# where a wavefront's time goes (one build per phase), what a phase is worth to the kernel (ablation builds, wrong results),
# and the kernel against round 5's on this very GPU
bash scripts/march_timing_all.sh > $O/march_phase_timers.jsonl 2>/dev/null
REPS=4 bash scripts/ab_fused_min.sh _ab_full _ab_NOTAB _ab_NOPASS _ab_NOEVALPASS _ab_NOCOLLIDE _ab_NOREINIT > $O/fused_ablations.txt 2>&1
REPS=6 bash scripts/ab_fused_min.sh _r5 "" > $O/ab_vs_round5.txt 2>&1
bash scripts/ab_real50k.sh > $O/ab_vs_round5_real50k.txt 2>&1
# the DEM run with its tiles by index / as patches of the packing, both packings
for w in bench real; do for t in 0 1; do python scripts/dem_rate.py $w $t 2>/dev/null | tail -3 >> $O/dem_rate_tiles.txt; done; done
CT=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_ct.so   # make AB=1 ABTAG=_ct ABFLAGS=-DLBMDEM_CHAIN_TIMING: where a tile's sub-step goes
if [ -f $CT ]; then
  LBMDEM_HIP_LIBRARY=$CT python scripts/dem_chain_times.py 50000 96 2>/dev/null > $O/dem_chain_tiles.txt
  echo >> $O/dem_chain_tiles.txt; LBMDEM_HIP_LIBRARY=$CT python scripts/dem_chain_times.py 50000 96 real 2>/dev/null >> $O/dem_chain_tiles.txt
fi
LBMDEM_RCCL_LIBRARY=$PWD/tests/rccl_shim/librccl.so.1 LBMDEM_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_two_ranks_one_gpu.json
bash scripts/strip_proxy_c.sh r06_final 8 4096 > $O/strip_proxy_c_4096.log 2>&1
bash scripts/strip_proxy_c.sh r06_final8k 8 8192 > $O/strip_proxy_c_8192.log 2>&1
cp gpurun_out/proxyc_r06_final*.json $O/ 2>/dev/null
tail -1 $O/bench_final.json | cut -c1-400
