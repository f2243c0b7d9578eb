# bench.py A/B of experiment-build libraries (args = suffixes after liblbmdem_hip), interleaved; REPS, STEPS, WORKLOAD from the env
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "fused_ms", d["collide_stream_kernel_ms"], "frac", d["roofline"]["frac"])'
for rep in $(seq ${REPS:-3}); do for lib in "$@"; do
  echo "[$lib] $(LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip$lib.so python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --workload ${WORKLOAD:-metric} 2>/dev/null | tail -1 | python -c "$P")"
done; done
