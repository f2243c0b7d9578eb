# A/B inside ONE gpurun call (same GPU): per-kernel times from rocprofv3 for the experiment build with
# several settings. usage: bash scripts/ab_forces.sh "<env1>" "<env2>" ...   (each arg = env assignments, may be empty)
cd /tmp && export TMPDIR=/tmp
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
k=0
for e in "$@"; do
  k=$((k+1)); O=$GRAFT_REPO_ROOT/gpurun_out/abf_$k; rm -rf $O
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O.log 2>&1
  echo "== [$e]  $(tail -1 $O.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "fused", d["collide_stream_kernel_ms"], "lbm_only", d["lbm_step_only_ms"])' 2>/dev/null)"
  python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    n=row["Name"]
    if any(t in n for t in ("k_cs_march","k_forces","k_obst_paint","k_fill_u64","k_dem_entries","k_obst_fill")):
        print("   %-28s calls %5s avg %9.1f us" % (n.split("(")[0].split("::")[-1][:28], row["Calls"], float(row["AverageNs"])/1e3))
PY
done
