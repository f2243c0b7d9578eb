# A/B inside ONE gpurun call: the current build vs scripts/dbg_libs/liblbmdem_hip_$1.so. Extra env for the current build: $2
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["collide_stream_kernel_ms"], d["roofline"]["frac"])'
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/cur.so
for rep in 1 2 3; do
  cp scripts/dbg_libs/liblbmdem_hip_$1.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "$1:      $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
  cp /tmp/cur.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "current: $(env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
done
