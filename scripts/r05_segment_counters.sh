# Why are long row segments of the fused kernel slow even when a queue balances them (round-4 open question)?
# One counter pass per main segment length 32 / 64 / 128 rows (experiment build: LBMDEM_CS_ROWS, chunk = 128 so that every
# length divides it; tail levels as in the product) over the L2 / memory-side and address-translation counters, plus the
# kernel's duration from the same runs. Output: gpurun_out/r05_segment_length_counters.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_segment_length_counters.txt; : > $O
export LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab.so
for seg in 32 64 128; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_DRAM_sum" "TCC_EA_RD_UNCACHED_32B_sum TCC_EA_WR_UNCACHED_32B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    echo "[main segments of $seg rows, chunk 128] $c" >> $O
    LBMDEM_CS_ROWS=$seg LBMDEM_CHUNK=128 bash scripts/pmc_kernel.sh seg$seg k_cs_march "$c" >> $O 2>&1
  done
  echo "[main segments of $seg rows, chunk 128] kernel time" >> $O
  LBMDEM_CS_ROWS=$seg LBMDEM_CHUNK=128 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('fused kernel ms', d['collide_stream_kernel_ms'], 'step ms', d['ms_per_step'])" >> $O
done
cat $O
