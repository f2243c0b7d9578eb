# bench A/B of experiment builds with the DEM rate: args = library suffixes
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print("MLUPS", d["value"], "ms/step", d["ms_per_step"], "fused_ms", d["collide_stream_kernel_ms"], "dem_only/s", d["dem_only_steps_per_s"])'
for rep in 1 2 3; do for lib in "$@"; do
 echo "[$lib] $(LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip$lib.so python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")"
done; done
