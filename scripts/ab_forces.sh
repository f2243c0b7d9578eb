# forces-kernel time (rocprofv3) for the current build and scripts/dbg_libs/liblbmdem_hip_<x>.so, in ONE call.  LIBS="a b"
cd $GRAFT_REPO_ROOT
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/cur.so
one() { bash scripts/prof_kernels.sh $1 > /dev/null 2>&1; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$1/$1_kernel_stats.csv")))
print("$1", {r["Name"].split("(")[0].split("::")[-1]: round(float(r["AverageNs"])/1e3,1) for r in rows if "k_forces" in r["Name"] or "k_cs_march" in r["Name"]})
PY
}
for rep in 1 2; do
  cp /tmp/cur.so 2d-lbm-dem_amd/liblbmdem_hip.so; one current$rep
  for l in $LIBS; do cp scripts/dbg_libs/liblbmdem_hip_$l.so 2d-lbm-dem_amd/liblbmdem_hip.so; one $l$rep; done
done
cp /tmp/cur.so 2d-lbm-dem_amd/liblbmdem_hip.so
