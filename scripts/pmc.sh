# usage: bash scripts/pmc.sh <tag> <variant> "<counters>"
# The LBMDEM_CS_* / LBMDEM_MARCH knobs only exist in the experiment build (make -C 2d-lbm-dem_amd/csrc AB=1): the product
# library ignores them, so without this line every "variant" below would silently be the same kernel.
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
[ -f "$LBMDEM_HIP_LIBRARY" ] || { echo "experiment build $LBMDEM_HIP_LIBRARY not found: make -C 2d-lbm-dem_amd/csrc AB=1"; exit 1; }
tag=$1; var=$2; ctrs=$3
cd /tmp && export TMPDIR=/tmp
LBMDEM_CS_VARIANT=$var rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    cnt[(k,row["Counter_Name"])]+=1
for k,d in acc.items():
    if "collide" in k or "march" in k or "forces" in k:
        print(k, {c: round(v/cnt[(k,c)],1) for c,v in d.items()})
PY
