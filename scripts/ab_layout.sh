# A/B of the population layout inside ONE gpurun call: tiles of 16 y-nodes (default build) vs nine planes
# (scripts/dbg_libs/liblbmdem_hip_planes.so, built with -DLBMDEM_F_TILES=0).
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["collide_stream_kernel_ms"], d["roofline"]["frac"])'
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/tiles.so
for rep in 1 2 3; do
  cp scripts/dbg_libs/liblbmdem_hip_planes.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "planes: $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
  cp /tmp/tiles.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "tiles:  $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
done
