cd $GRAFT_REPO_ROOT
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/good.so
for D in NO_IBB NO_FEQ NO_IBB_NO_FEQ; do
  cp scripts/dbg_libs/liblbmdem_hip_$D.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "$D: $(bash scripts/pmc.sh dbg_$D 25 'SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES' | grep march | cut -c60-)"
done
cp /tmp/good.so 2d-lbm-dem_amd/liblbmdem_hip.so
echo "FULL: $(bash scripts/pmc.sh dbg_FULL 25 'SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES' | grep march | cut -c60-)"
