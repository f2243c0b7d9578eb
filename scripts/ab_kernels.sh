# per-kernel times (rocprofv3 kernel stats) of several experiment builds: args = library suffixes
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for lib in "$@"; do
  O=$GRAFT_REPO_ROOT/gpurun_out/abk$lib; rm -rf $O
  LBMDEM_HIP_LIBRARY=$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip$lib.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python - "$lib" $(find $O -name "*kernel_stats*" | head -1) <<'PY'
import csv,sys,re
out=[]
for r in csv.DictReader(open(sys.argv[2])):
    m=re.search(r'(k_[a-z_0-9]+)', r['Name'])
    if m and m.group(1) in ('k_cs_march','k_dem_entries','k_forces_table','k_obst_paint','k_forces_gather_queue') and int(r['Calls'])>20:
        out.append(f"{m.group(1)} {float(r['AverageNs'])/1e3:.1f}")
print(f"[{sys.argv[1]}] " + "  ".join(out))
PY
done; done
