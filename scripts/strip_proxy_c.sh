# usage: bash scripts/strip_proxy_c.sh <tag> [world] [lx]  -> gpurun_out/proxyc_<tag>.json (+ _busy.json from the kernel trace
# of the interior rank's process). Needs `make -C 2d-lbm-dem_amd/csrc AB=1` and tests/rccl_shim built.
tag=$1; world=${2:-8}; lx=${3:-4096}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/proxyc_$tag
rocprofv3 --kernel-trace --output-format csv -d $O/proxyc_$tag/%pid% -o t -- python $GRAFT_REPO_ROOT/scripts/strip_proxy_c.py $world $lx > $O/proxyc_$tag.log 2>&1
grep "^{" $O/proxyc_$tag.log | tail -1 > $O/proxyc_$tag.json
pid=$(python -c "import json,sys; print(json.load(open('$O/proxyc_$tag.json'))['mid_rank']['pid'])")
trace=$(find $O/proxyc_$tag/$pid -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/strip_proxy_busy.py $trace > $O/proxyc_${tag}_busy.json
cat $O/proxyc_$tag.json; cat $O/proxyc_${tag}_busy.json
rm -rf $O/proxyc_$tag
