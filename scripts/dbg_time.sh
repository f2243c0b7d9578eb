# timing-only ablation: swap in a debug library (grain logic partly disabled; results are WRONG) and bench
cd $GRAFT_REPO_ROOT
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/good.so
for D in NO_IBB NO_FEQ NO_IBB_NO_FEQ; do
  cp scripts/dbg_libs/liblbmdem_hip_$D.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "$D: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["collide_stream_kernel_ms"])')"
done
cp /tmp/good.so 2d-lbm-dem_amd/liblbmdem_hip.so
echo "FULL: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["collide_stream_kernel_ms"])')"
