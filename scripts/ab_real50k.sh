# interleaved A/B of library builds on the reference's own geometry (bench.py --workload real50k): fused kernel ms, step ms, DEM-only rate
# usage: bash scripts/ab_real50k.sh [<suffix> ...]   (default: the round-5 library against the product; REPS from the env)
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["collide_stream_kernel_ms"], d["ms_per_step"], d["dem_only_steps_per_s"])'
if [ $# -eq 0 ]; then set -- _r5 ""; fi
for rep in $(seq ${REPS:-3}); do for lib in "$@"; do
  echo "[$lib] $(LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip$lib.so python bench.py --steps 40 --warmup 5 --settle 20 --no-cpu-baseline --workload real50k --long-steps 0 --real-steps 0 2>/dev/null | tail -1 | python -c "$P")"
done; done
