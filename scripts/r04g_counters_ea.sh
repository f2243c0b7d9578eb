# L2 -> memory request mix of the fused kernel: 62-lane (LBMDEM_CS_VARIANT=30) vs 60-lane (=28) windows, experiment build,
# product work order; separate passes. Output: gpurun_out/r04g_counters_ea.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g_counters_ea.txt; : > $O
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_WRITE_SECTORS_sum TCC_READ_SECTORS_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum"; do
  for v in 30 28; do
    echo "[LBMDEM_CS_VARIANT=$v] $c" >> $O
    LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab.so LBMDEM_CS_VARIANT=$v bash scripts/pmc_kernel.sh e$v k_cs_march "$c" >> $O 2>&1
  done
done
cat $O
