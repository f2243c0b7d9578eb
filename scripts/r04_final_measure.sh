# Round-4 evidence in ONE gpurun call (one GPU): bench lines, rocprofv3 kernel stats, PMC traffic (tied to the source hash),
# the C transport's strip period (8 processes on this GPU through tests/rccl_shim). Output: gpurun_out/r04/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04}; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --workload real50k --no-cpu-baseline > $O/bench_real50k.json 2>/dev/null
python bench.py --workload configs4 --no-cpu-baseline > $O/bench_configs4_one_gpu.json 2>/dev/null
python bench.py --precision f32 > $O/bench_f32.json 2>/dev/null
bash scripts/prof_kernels.sh r04_final > $O/prof_kernels.log 2>&1
cp $(find gpurun_out/prof_r04_final -name "*kernel_stats*" | head -1) $O/kernel_stats_final.csv 2>/dev/null
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/ 2>/dev/null
LBMDEM_RCCL_LIBRARY=$PWD/tests/rccl_shim/librccl.so.1 LBMDEM_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_two_ranks_one_gpu.json 2>/dev/null
bash scripts/strip_proxy_c.sh r04_final 8 4096 > $O/strip_proxy_c_4096.log 2>&1
bash scripts/strip_proxy_c.sh r04_final8k 8 8192 > $O/strip_proxy_c_8192.log 2>&1
cp gpurun_out/proxyc_r04_final*.json $O/ 2>/dev/null
# where the launch's wave slots go (experiment build with -DMARCH_TRACE): uniform 32-row segments vs the product's tapered plan
TR=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_trace.so
if [ -f $TR ]; then
  LBMDEM_HIP_LIBRARY=$TR LBMDEM_CS_VARIANT=25 python scripts/march_trace.py 2>/dev/null | tail -1 > $O/march_trace_uniform32.json
  LBMDEM_HIP_LIBRARY=$TR python scripts/march_trace.py 2>/dev/null | tail -1 > $O/march_trace_product_plan.json
  LBMDEM_HIP_LIBRARY=$TR LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=138 LBMDEM_PLAN= python scripts/march_trace.py 2>/dev/null | tail -1 > $O/march_trace_one_round_138.json
fi
tail -1 $O/bench_final.json | cut -c1-400
