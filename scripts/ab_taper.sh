# schedules of the fused kernel (experiment build): untraced kernel time (bench.py) for each environment string given
# ("LBMDEM_CS_ROWS=64 LBMDEM_PLAN=32:64,16:64,8:64" ...: main segment length and tail levels seg:rows of every XCD band; an empty
# LBMDEM_PLAN= is the untapered band schedule), interleaved and repeated, next to the compiled 32-row kernel with uniform
# segments (LBMDEM_CS_VARIANT=25, round 3's product); TRACE=1 adds the slot-time accounting of scripts/march_trace.py for each
cd $GRAFT_REPO_ROOT
AB=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab.so; TR=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_trace.so
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "fused_ms", d["collide_stream_kernel_ms"], "frac", d["roofline"]["frac"])'
for rep in $(seq ${REPS:-2}); do
  echo "[uniform 32-row segments, compiled] $(LBMDEM_HIP_LIBRARY=$AB LBMDEM_CS_VARIANT=25 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --workload ${WORKLOAD:-metric} 2>/dev/null | tail -1 | python -c "$P")"
  for t in "$@"; do
    echo "[$t] $(env LBMDEM_HIP_LIBRARY=$AB $t python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --workload ${WORKLOAD:-metric} 2>/dev/null | tail -1 | python -c "$P")"
  done
done
if [ -n "$TRACE" ]; then
  for t in "$@"; do
    echo "== trace [$t]"; env LBMDEM_HIP_LIBRARY=$TR $t python scripts/march_trace.py 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("fused_kernel_ms","slot_time","wave_lifetime_us","waves_in_flight_by_launch_decile","rows_per_us_by_launch_decile")})'
  done
fi
