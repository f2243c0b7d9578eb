# the per-phase timers of the fused kernel's iteration, one experiment build per phase (liblbmdem_hip_ab_t<k>.so =
# make AB=1 ABTAG=_t<k> ABFLAGS=-DMARCH_TIMING=<k>), one GPU: scripts/march_timing.py for each
cd $GRAFT_REPO_ROOT
for k in 0 1 2 3 4 10 11 5 6 7; do
  LBMDEM_HIP_LIBRARY=$PWD/2d-lbm-dem_amd/liblbmdem_hip_ab_t$k.so python scripts/march_timing.py 2>/dev/null | tail -1
done
