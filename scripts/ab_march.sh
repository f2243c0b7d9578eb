# A/B of the marching kernels inside ONE gpurun call (same GPU), experiment build (make AB=1):
# usage: bash scripts/ab_march.sh "<env1>" "<env2>" ...   each arg = env assignments for one bench run
cd $GRAFT_REPO_ROOT
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
P='import sys,json; d=json.loads(sys.stdin.read()); print("MLUPS", d["value"], "ms/step", d["ms_per_step"], "fused_ms", d["collide_stream_kernel_ms"], "frac", d["roofline"]["frac"])'
for rep in 1 2; do
  for e in "$@"; do
    echo "[$e] $(env $e python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")"
  done
done
