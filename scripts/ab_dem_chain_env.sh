#!/bin/bash
# A/B of experiment switches of k_dem_chain on one GPU box (experiment build of the library): DEM only, the bench packing.
# usage: scripts/ab_dem_chain_env.sh "VAR1=1" "VAR2=1" ...   ("" = no switch); each variant twice, interleaved
export LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_ct.so   # make AB=1 ABTAG=_ct ABFLAGS=-DLBMDEM_CHAIN_TIMING
for rep in 1 2; do
  for v in "$@"; do
    echo -n "[$v] "; env $v python scripts/dem_chain_times.py ${GRAINS:-50000} 96 2>/dev/null | grep "grains,\|per tile, us"
  done
done
