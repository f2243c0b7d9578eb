# SQ and L2 (TCC) counters of the final fused kernel, and the write-request mix of 62- vs 60-lane windows (experiment build,
# LBMDEM_CS_VARIANT=30 = 62 lanes with the product work order). Output: gpurun_out/r04g_counters.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g_counters.txt; : > $O
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITEBACK_sum"; do
  echo "[product, 60 lanes] $c" >> $O
  bash scripts/pmc_kernel.sh g k_cs_march "$c" >> $O 2>&1
done
for c in "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  for v in 30 28; do
    echo "[experiment build, LBMDEM_CS_VARIANT=$v (30 = 62 lanes, 28 = 60 lanes)] $c" >> $O
    LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab.so LBMDEM_CS_VARIANT=$v bash scripts/pmc_kernel.sh g$v k_cs_march "$c" >> $O 2>&1
  done
done
cat $O
