# SQ and L2 (TCC) counters of the round's final fused kernel (both instantiations: ", t" = with the change bits, the steady
# state; ", f" = the first steps of a run, both maps read everywhere). Output: gpurun_out/${1:-r06}_sq_counters.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06}_sq_counters.txt; : > $O
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITEBACK_sum"; do
  echo "[$c]" >> $O
  bash scripts/pmc_kernel.sh s k_cs_march "$c" >> $O 2>&1
done
cat $O
