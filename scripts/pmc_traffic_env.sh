# HBM traffic (FETCH_SIZE 2048 B/unit, WRITE_SIZE 1024 B/unit on gfx950, see profiles/r01_d_pmc_traffic.json) of the fused
# kernel under several environment settings of the experiment build.
#   usage: bash scripts/pmc_traffic_env.sh "LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=32" "LBMDEM_CS_VARIANT=28 LBMDEM_CS_ROWS=138"
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trafficenv; rm -rf $O; mkdir -p $O
i=0
for e in "$@"; do for c in FETCH_SIZE WRITE_SIZE; do
  env $e rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/v${i}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done; i=$((i+1)); done
python - "$@" <<PY
import csv,glob,sys
for i,e in enumerate(sys.argv[1:]):
    res={}
    for c,unit in (("FETCH_SIZE",2048),("WRITE_SIZE",1024)):
        f=glob.glob(f"$O/v{i}_{c}/**/*counter_collection.csv", recursive=True)[0]
        vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_cs_march" in r["Kernel_Name"] or "k_collide_stream" in r["Kernel_Name"]]
        res[c]=sum(vals)/len(vals)*unit/1e9
    print(f"[{e}] fetch {res['FETCH_SIZE']:.3f} GB  write {res['WRITE_SIZE']:.3f} GB  total {res['FETCH_SIZE']+res['WRITE_SIZE']:.3f} GB")
PY
