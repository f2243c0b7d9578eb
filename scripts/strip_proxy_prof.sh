# usage: bash scripts/strip_proxy_prof.sh <tag> [world] [lx]  -> gpurun_out/proxy_<tag>.json (+ _busy.json from the kernel trace)
tag=$1; world=${2:-8}; lx=${3:-4096}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/proxy_$tag
rocprofv3 --kernel-trace --output-format csv -d $O/proxy_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/strip_proxy.py $world $lx > $O/proxy_$tag.log 2>&1
grep "^{" $O/proxy_$tag.log | tail -1 > $O/proxy_$tag.json
python $GRAFT_REPO_ROOT/scripts/strip_proxy_busy.py $(find $O/proxy_$tag -name "*kernel_trace.csv" | head -1) > $O/proxy_${tag}_busy.json
cat $O/proxy_$tag.json; cat $O/proxy_${tag}_busy.json
find $O/proxy_$tag -name "*kernel_trace.csv" -delete
