# Long runs of the library's own transport (lbmdem_comm_run through strips.CCommRunner) with several ranks on ONE GPU through
# the test-only RCCL stand-in, against the CPU oracle (tests/multi_gpu_check.py MODE=ccomm SHARED_GPU=1).
# usage: bash scripts/soak_c_transport.sh "<worlds>" "<nsteps list>"      -> stdout
cd $GRAFT_REPO_ROOT
export LBMDEM_RCCL_LIBRARY=$PWD/tests/rccl_shim/librccl.so.1 RCCL_SHIM_TIMEOUT_S=120 SHARED_GPU=1 MODE=ccomm
for w in ${1:-2 3 4}; do for n in ${2:-4003 8005}; do
  r=$(NSTEPS=$n timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$w --master-addr 127.0.0.1 --master-port $((29700 + w)) tests/multi_gpu_check.py 2>&1 | grep -o "MULTI-GPU-OK.*\|MISMATCH.*\|Error.*" | head -1)
  echo "world $w, $n sub-steps: ${r:-FAILED (no result line)}"
done; done
