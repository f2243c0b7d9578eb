"""GPU, only where the box has at least two GPUs (the gpurun boxes have one: skipped there): the real multi-GPU
strip path -- one process per GPU under torch.distributed.run, RCCL -- bit-equal to the CPU oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["distributed", "ccomm", "replicated"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_strips_over_rccl_on_real_gpus(world, mode):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MODE=mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world), os.path.join(ROOT, "tests", "multi_gpu_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "MULTI-GPU-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
