# A/B of the strip proxy for several libraries inside one gpurun call: bash scripts/ab_proxy.sh <lib1> <lib2> ...
k=0
for lib in "$@"; do
  k=$((k+1))
  LBMDEM_HIP_LIBRARY=$lib bash $GRAFT_REPO_ROOT/scripts/strip_proxy_prof.sh ab$k > /dev/null 2>&1
  echo "== $lib"; cut -c1-330 $GRAFT_REPO_ROOT/gpurun_out/proxy_ab${k}_busy.json
done
