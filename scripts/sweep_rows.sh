# usage: ROWS="32 34 ..." bash scripts/sweep_rows.sh   -- fused kernel with run-time rows per wave (variant 28 = LX 0 + XCD remap)
# The LBMDEM_CS_* / LBMDEM_MARCH knobs only exist in the experiment build (make -C 2d-lbm-dem_amd/csrc AB=1): the product
# library ignores them, so without this line every "variant" below would silently be the same kernel.
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
[ -f "$LBMDEM_HIP_LIBRARY" ] || { echo "experiment build $LBMDEM_HIP_LIBRARY not found: make -C 2d-lbm-dem_amd/csrc AB=1"; exit 1; }
cd $GRAFT_REPO_ROOT
echo "compiled LX=32: $(LBMDEM_CS_VARIANT=25 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["collide_stream_kernel_ms"], d["roofline"]["frac"])')"
for r in $ROWS; do
  echo "rows $r: $(LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=$r python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["collide_stream_kernel_ms"], d["roofline"]["frac"])')"
done
