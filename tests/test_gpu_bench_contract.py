"""GPU: bench.py's output contract, single process and through the torch.distributed launcher (one
rank: the same strip-driver + RCCL plumbing the multi-GPU runs use)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-400:]
    return json.loads(lines[0])


def test_single_process_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-500:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "f64"
    assert d["unit"] == "MLUPS" and d["value"] > 1000 and d["vs_baseline"] is None and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["warmup"] == 1 and d["config"]["settle_steps"] == 40      # the untimed steps before the warm-up are reported
    # comparable across boxes and rounds: the driver's own protocol (K steps straight after W warm-up steps) next to the
    # headline, and a plain copy of the same traffic on the same GPU as the yardstick
    assert d["ms_per_step_unsettled"] > 0 and 2000 < d["hbm_copy_gbs"] < 8000
    assert abs(r["achieved_over_copy"] - r["achieved"] / d["hbm_copy_gbs"]) < 1e-3
    assert r["frac_of_copy"] is None or 0.3 < r["frac_of_copy"] < 1.2
    assert d["config"]["dem_chain"]["launches"] > 0 and d["config"]["dem_chain"]["resident"] == d["config"]["dem_chain"]["tile_slots"]
    # what the README quotes is in the driver-timed record itself: SURVEY 8-d's 200-step window of the same run (the fused
    # kernel averaged over 25 timed launches), the reference's own 50000.data geometry, and the whole step priced like the kernel
    assert 0.3 < d["ms_per_step_200"] < 5.0 and d["launches_timed_200"] >= 25 and 0.2 < d["collide_stream_kernel_ms_200"] < d["ms_per_step_200"]
    assert 0.3 < d["real50k_ms_per_step"] < 5.0 and d["real50k_mlups"] > 1000
    assert abs(r["step_frac"] - d["value"] * 1e6 * 148.0 / 8e12) < 2e-3 and 0.05 < r["step_frac"] < r["frac"]
    assert r["frac_200"] > 0.2 and r["step_frac_200"] < r["frac_200"]
    assert d["dem_chain_recoveries"] == 0


def test_settle_steps_can_be_switched_off():
    """--settle 0: no untimed steps before the W warm-up steps (the timed region is K steps either way)"""
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--settle", "0", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-500:]
    d = _last_json(out.stdout)
    assert d["steps"] == 3 and d["config"]["settle_steps"] == 0 and d["value"] > 1000
    assert d["ms_per_step_unsettled"] == d["ms_per_step"]            # nothing to tell apart


def test_launcher_path_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--strips", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 1000


def test_plain_python_strips_form_one_rank():
    """`python bench.py --gpus 1 --strips` without any launcher: the strip driver (C transport, one rank) in a plain process"""
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--strips", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 1000 and d["config"]["driver"].startswith("C (lbmdem_comm_run")
