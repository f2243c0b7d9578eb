# usage: bash scripts/pmc_kernel.sh <tag> <kernel-substring> "<counters>"
export LBMDEM_BENCH_NO_LEGS=1   # bench.py: no 200-step / real50k legs behind the timed region
tag=$1; pat=$2; ctrs=$3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag   # (a pass whose counters do not exist must not read the previous pass's file)
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    if "$pat" in row["Kernel_Name"]: acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,d in acc.items(): print(k, {c: round(sum(v)/len(v),1) for c,v in d.items()}, "n=", len(list(d.values())[0]))
PY
