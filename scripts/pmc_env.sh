# PMC counters of the fused kernel under several settings of the experiment build, one pass per counter group.
#   usage: bash scripts/pmc_env.sh "<counters group 1>;<group 2>;..." "<env 1>" "<env 2>" ...
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcenv; rm -rf $O; mkdir -p $O
groups=$1; shift
i=0
for e in "$@"; do
  g=0
  echo "$groups" | tr ';' '\n' | while read ctrs; do
    env $e rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/v${i}_g$g -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    g=$((g+1))
  done
  i=$((i+1))
done
python - "$@" <<PY
import csv,glob,sys,collections
for i,e in enumerate(sys.argv[1:]):
    res=collections.OrderedDict()
    for f in sorted(glob.glob(f"$O/v{i}_g*/**/*counter_collection.csv", recursive=True)):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_cs_march" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c,v in acc.items(): res[c]=sum(v)/len(v)
    print(f"[{e}] " + "  ".join(f"{c} {v/1e6:.2f}M" for c,v in res.items()))
PY
