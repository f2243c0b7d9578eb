# the strip proxy under several environment settings (experiment build), one gpurun call:
# bash scripts/ab_proxy_env.sh "<env1>" "<env2>" ...
export LBMDEM_HIP_LIBRARY=${LBMDEM_HIP_LIBRARY:-$GRAFT_REPO_ROOT/2d-lbm-dem_amd/liblbmdem_hip_ab.so}
k=0
for e in "$@"; do
  k=$((k+1))
  env $e bash $GRAFT_REPO_ROOT/scripts/strip_proxy_prof.sh abe$k > /dev/null 2>&1
  echo "== [$e]"; cut -c1-140 $GRAFT_REPO_ROOT/gpurun_out/proxy_abe${k}_busy.json
done
