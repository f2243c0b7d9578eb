# Round-3 evidence in ONE gpurun call (one GPU): bench lines, rocprofv3 kernel stats, PMC traffic, the per-phase timers
# of the marching kernel, the issue-rate micro-benchmark and the A/B of the experiment kernels. Output: gpurun_out/r03/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --workload real50k --no-cpu-baseline > $O/bench_real50k.json 2>/dev/null
python bench.py --workload configs4 --no-cpu-baseline > $O/bench_configs4_one_gpu.json 2>/dev/null
python bench.py --precision f32 > $O/bench_f32.json 2>/dev/null
bash scripts/prof_kernels.sh r03_final > $O/prof_kernels.log 2>&1
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/traffic/pmc_traffic.json $O/ 2>/dev/null
for v in 0 1 2 3 4 5 6 7; do LBMDEM_HIP_LIBRARY=$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab_t$v.so timeout 300 python scripts/march_timing.py 2>/dev/null | tail -1; done > $O/march_phase_timers.jsonl
timeout 120 scripts/micro/issue_latency.bin > $O/issue_rate_micro.txt 2>&1
bash scripts/ab_march_libs.sh "_ab:LBMDEM_MARCH=2" "_ab_recip:LBMDEM_MARCH=2" "_ab_branchy:LBMDEM_MARCH=2" "_ab:LBMDEM_MARCH_DYNLDS=24000" "_ab:LBMDEM_MARCH=3" "_ab:LBMDEM_MARCH=21" "_ab_m3u:LBMDEM_MARCH=3" "_ab_m3u:LBMDEM_MARCH=21" "_ab_m3m:LBMDEM_MARCH=3" "_ab_m3m:LBMDEM_MARCH=21" > $O/march_ab.txt 2>&1
bash scripts/ab_march_libs.sh "_ab:LBMDEM_MARCH=2" "_ab_recip:LBMDEM_MARCH=2" "_ab_branchy:LBMDEM_MARCH=2" "_ab_m3m:LBMDEM_MARCH=3" >> $O/march_ab.txt 2>&1
( python bench.py --steps 25000 --warmup 50 --no-cpu-baseline > /dev/null 2>&1 & for i in 1 2 3 4 5 6 7 8 9 10; do sleep 3; echo "t=$((i*3))s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed "s/.*: //" | tr "\n" " ")"; done; wait ) > $O/clocks_power.txt 2>&1
tail -1 $O/bench_final.json | cut -c1-400
