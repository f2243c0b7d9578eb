# HBM traffic (FETCH_SIZE 2048 B/unit, WRITE_SIZE 1024 B/unit on gfx950, see profiles/r01_d_pmc_traffic.json) of the fused
# kernel for several LBMDEM_CS_VARIANT values.   usage: VARS="25 29 30" bash scripts/pmc_traffic_variants.sh
# The LBMDEM_CS_* / LBMDEM_MARCH knobs only exist in the experiment build (make -C 2d-lbm-dem_amd/csrc AB=1): the product
# library ignores them, so without this line every "variant" below would silently be the same kernel.
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
[ -f "$LBMDEM_HIP_LIBRARY" ] || { echo "experiment build $LBMDEM_HIP_LIBRARY not found: make -C 2d-lbm-dem_amd/csrc AB=1"; exit 1; }
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trafficv; mkdir -p $O
for v in $VARS; do for c in FETCH_SIZE WRITE_SIZE; do
  LBMDEM_CS_VARIANT=$v rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/v${v}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done; done
python - <<PY
import csv,glob,collections
for v in "$VARS".split():
    res={}
    for c,unit in (("FETCH_SIZE",2048),("WRITE_SIZE",1024)):
        f=glob.glob(f"$O/v{v}_{c}/**/*counter_collection.csv", recursive=True)[0]
        vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_cs_march" in r["Kernel_Name"] or "k_collide_stream" in r["Kernel_Name"]]
        res[c]=sum(vals)/len(vals)*unit/1e9
    print(f"variant {v}: fetch {res['FETCH_SIZE']:.3f} GB  write {res['WRITE_SIZE']:.3f} GB  total {res['FETCH_SIZE']+res['WRITE_SIZE']:.3f} GB")
PY
