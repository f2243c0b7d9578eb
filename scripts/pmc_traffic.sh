# HBM traffic of the fused kernel from PMC counters, calibrated on copy kernels of known size.
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots). Output: gpurun_out/traffic/*.csv + summary.
# (the calibration binary is git-ignored: built where it is missing)
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
[ -x $GRAFT_REPO_ROOT/scripts/micro/stream_pattern ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $GRAFT_REPO_ROOT/scripts/micro/stream_pattern $GRAFT_REPO_ROOT/scripts/micro/stream_pattern.hip
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/cal_$c -o cal -- $GRAFT_REPO_ROOT/scripts/micro/stream_pattern > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/run_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,json
O="$O"
def mean(path, key):
    f=glob.glob(path+"/**/*counter_collection.csv", recursive=True)[0]
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v)/len(v) for k,v in acc.items() if key in k}
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    cal=mean(f"{O}/cal_{c}","k_copy8"); cal16=mean(f"{O}/cal_{c}","k_copy16"); run=mean(f"{O}/run_{c}","k_cs_march")
    # (two instantiations of the kernel: the steady state's reads the previous map only where the maps differ -- ", true>")
    steady=[v for k,v in run.items() if ", true>" in k or ",true>" in k or "(bool)1" in k]
    out[c]={"copy8_raw":list(cal.values())[0],"copy16_raw":list(cal16.values())[0],"march_raw":steady[0] if steady else list(run.values())[0],
            "march_kernels":sorted(run)}
known=9*4096*4096*8   # bytes read (and written) by the copy kernels
for c in out:
    d=out[c]; d["bytes_per_unit_copy8"]=known/d["copy8_raw"]; d["bytes_per_unit_copy16"]=known/d["copy16_raw"]
    d["march_bytes_calibrated_8B"]=d["march_raw"]*d["bytes_per_unit_copy8"]
print(json.dumps(out, indent=1))
json.dump(out, open(f"{O}/summary.json","w"), indent=1)
# the file bench.py reads (copy to profiles/rNN_pmc_traffic.json)
fb=out["FETCH_SIZE"]["march_bytes_calibrated_8B"]; wb=out["WRITE_SIZE"]["march_bytes_calibrated_8B"]; alg=148*4096*4096
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
json.dump({"source_sha256": bench.library_source_sha256(), "_what": "HBM traffic of k_cs_march per launch, 4096x4096/50k grains (bench.py default workload), rocprofv3 PMC FETCH_SIZE and "
           "WRITE_SIZE in separate passes (scripts/pmc_traffic.sh), calibrated on copy kernels of known size",
           "fetch_bytes": fb, "write_bytes": wb, "fetch_bytes_per_unit": out["FETCH_SIZE"]["bytes_per_unit_copy8"],
           "write_bytes_per_unit": out["WRITE_SIZE"]["bytes_per_unit_copy8"], "traffic_bytes_per_launch": fb+wb,
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fb+wb)/alg}, open(f"{O}/pmc_traffic.json","w"), indent=1)
PY
