"""CPU: the host logic of bench.py that decides HOW the benchmark is started -- no GPU needed (on a box without one every
path must end in one clean message, never in a CPU fallback or a traceback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("LBMDEM_BENCH_DEVICES", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def test_single_gpu_run_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this box has a GPU")
    out = _run(["--steps", "2", "--warmup", "1"])
    assert out.returncode != 0 and "no HIP device visible (there is no CPU fallback)" in out.stderr
    assert "Traceback" not in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_multi_gpu_request_without_enough_gpus_is_one_clean_message():
    """`python bench.py --gpus 2` as the driver starts it (no launcher around it): the self-launch checks the device count
    FIRST -- exit code 2 and one line, not a torch.distributed.run error report."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two GPUs")
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert out.returncode == 2 and f"needs 2 GPUs, this node has {have}" in out.stderr, (out.returncode, out.stderr[-500:])
    assert "Traceback" not in out.stderr and "ChildFailedError" not in out.stderr


def test_launcher_world_size_must_match_gpus():
    out = _run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=4 does not match --gpus 2" in out.stderr


def test_device_map_and_port_helpers():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    os.environ.pop("LBMDEM_BENCH_DEVICES", None)
    assert bench.device_map(2) is None
    os.environ["LBMDEM_BENCH_DEVICES"] = "0,0,1"
    try:
        assert bench.device_map(3) == [0, 0, 1]
        try:
            bench.device_map(4)
            assert False
        except SystemExit as e:
            assert "names 3 devices for 4 ranks" in str(e)
    finally:
        os.environ.pop("LBMDEM_BENCH_DEVICES", None)
    p = bench.free_port()
    assert 1024 < p < 65536
