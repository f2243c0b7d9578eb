# A/B of several experiment builds inside ONE gpurun call: each arg = "<lib suffix>:<env assignments>", e.g.
#   bash scripts/ab_march_libs.sh "_ab:LBMDEM_MARCH=2" "_ab_bperm:LBMDEM_MARCH=2" "_ab_m3a:LBMDEM_MARCH=3"
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print("MLUPS", d["value"], "ms/step", d["ms_per_step"], "fused_ms", d["collide_stream_kernel_ms"], "frac", d["roofline"]["frac"])'
for rep in 1 2; do
  for a in "$@"; do
    lib=${a%%:*}; e=${a#*:}
    echo "[$lib $e] $(env LBMDEM_HIP_LIBRARY=$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip$lib.so $e python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")"
  done
done
