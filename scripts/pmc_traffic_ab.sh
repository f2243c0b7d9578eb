# HBM traffic of the fused kernel for several env settings of the experiment build (calibration as pmc_traffic.sh)
# usage: bash scripts/pmc_traffic_ab.sh "<env1>" "<env2>" ...
# (the calibration binary is git-ignored: built where it is missing)
[ -x $GRAFT_REPO_ROOT/scripts/micro/stream_pattern ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $GRAFT_REPO_ROOT/scripts/micro/stream_pattern $GRAFT_REPO_ROOT/scripts/micro/stream_pattern.hip
cd /tmp && export TMPDIR=/tmp
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
O=$GRAFT_REPO_ROOT/gpurun_out/traffic_ab; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/cal_$c -o cal -- $GRAFT_REPO_ROOT/scripts/micro/stream_pattern > /dev/null 2>&1
  k=0
  for e in "$@"; do
    k=$((k+1))
    env $e rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/run${k}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  done
done
python - "$@" <<PY
import csv,glob,collections,json,sys
O="$O"
def mean(path, key):
    f=glob.glob(path+"/**/*counter_collection.csv", recursive=True)[0]
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v)/len(v) for k,v in acc.items() if key in k}
known=9*4096*4096*8
unit={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    cal=mean(f"{O}/cal_{c}","k_copy8"); unit[c]=known/list(cal.values())[0]
print("bytes per counter unit", unit)
for k,e in enumerate(sys.argv[1:],1):
    for pat in ("k_cs_march","k_forces","k_obst_paint"):
        r=mean(f"{O}/run{k}_FETCH_SIZE",pat); w=mean(f"{O}/run{k}_WRITE_SIZE",pat)
        for name in r:
            print("[%s] %-24s read %.3f GB  write %.3f GB" % (e, name.split("(")[0][-24:], r[name]*unit["FETCH_SIZE"]/1e9, w.get(name,0)*unit["WRITE_SIZE"]/1e9))
PY
