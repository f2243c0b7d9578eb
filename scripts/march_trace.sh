# the row-time trace of the fused kernel for several segment lengths (experiment build with -DMARCH_TRACE), one gpurun call
export LBMDEM_HIP_LIBRARY=$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab_trace.so
[ -f "$LBMDEM_HIP_LIBRARY" ] || { echo "build it: make -C 2d-lbm-dem_amd/csrc AB=1 ABTAG=_trace ABFLAGS=-DMARCH_TRACE"; exit 1; }
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== compiled LX=32 (the product's kernel)"; python scripts/march_trace.py gpurun_out/trace_lx32.npz 2>&1 | tail -1
for r in ${ROWS:-32 70 138}; do
  echo "== run-time rows $r"; LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=$r python scripts/march_trace.py gpurun_out/trace_rows$r.npz 2>&1 | tail -1
done
