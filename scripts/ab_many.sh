# usage: LIBS="a b c" bash scripts/ab_many.sh  -- current build vs scripts/dbg_libs/liblbmdem_hip_<x>.so, interleaved, in ONE gpurun call
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["collide_stream_kernel_ms"], d["roofline"]["frac"])'
cp 2d-lbm-dem_amd/liblbmdem_hip.so /tmp/cur.so
for rep in 1 2; do
  cp /tmp/cur.so 2d-lbm-dem_amd/liblbmdem_hip.so
  echo "current: $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
  for l in $LIBS; do
    cp scripts/dbg_libs/liblbmdem_hip_$l.so 2d-lbm-dem_amd/liblbmdem_hip.so
    echo "$l: $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P")"
  done
done
cp /tmp/cur.so 2d-lbm-dem_amd/liblbmdem_hip.so
