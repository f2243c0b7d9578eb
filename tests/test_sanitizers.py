"""CPU: one AddressSanitizer + UndefinedBehaviorSanitizer run of the host-side C -- the oracle (oracle/lbmdem_oracle.c
through oracle/sanitize_check.c) and the C host driver (2d-lbm-dem_amd/host/main.c: argument parsing, the sample reader,
the decomposition check, the fork/wait logic of --gpus N up to the point where a GPU is needed). The reference has no
sanitizer configuration."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _sample(path, po):
    # grains on all four lattice edges (clipped discs), one beyond the lattice, two one fluid node apart, some in contact
    r = [0.7, 0.6, 0.8, 0.55, 0.6, 0.7, 0.62, 0.58, 0.9, 0.5]
    x = [0.5, 9.3, 4.0, 5.42, 2.0, 2.0 + 1.31, 7.0, 7.0, 4.6, 12.5]
    y = [3.0, 3.5, 0.55, 6.9, 5.0, 5.0, 2.0, 3.19, 3.6, 3.0]
    po.write_sample(str(path), np.array(r), np.array(x), np.array(y), comment="#sanitizer packing")


def test_oracle_under_asan_and_ubsan(po, tmp_path):
    exe = tmp_path / "sanitize_check"
    cc = subprocess.run(["gcc", "-std=gnu99", "-O1", *SAN, "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"),
                         os.path.join(ROOT, "oracle", "sanitize_check.c"), os.path.join(ROOT, "oracle", "lbmdem_oracle.c"),
                         "-lm", "-o", str(exe)], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    sample = tmp_path / "pack.data"
    _sample(sample, po)
    out = subprocess.run([str(exe), str(sample)], capture_output=True, text=True, env=ENV, timeout=300)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
    assert "sanitize_check: 10 grains, 130 steps" in out.stdout and "BAD" not in out.stdout


def test_host_driver_under_asan_and_ubsan(po, pkg, tmp_path):
    """The C driver instrumented: usage errors, a missing sample, a truncated sample, a decomposition whose strips are
    narrower than the margin (refused BEFORE anything is forked), and a good command line up to lbmdem_create -- which
    needs a GPU: here it must fail cleanly with the library's message (no CPU fallback), on a GPU box it runs 30 steps."""
    exe = tmp_path / "lbmdem_san"
    lib_dir = os.path.join(ROOT, "2d-lbm-dem_amd")
    cc = subprocess.run(["gcc", "-std=gnu99", "-O1", *SAN, os.path.join(lib_dir, "host", "main.c"), "-L" + lib_dir,
                         "-llbmdem_hip", "-Wl,-rpath," + lib_dir, "-lm", "-o", str(exe)], capture_output=True, text=True)
    if cc.returncode != 0 and "cannot find" in cc.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0")     # the HIP runtime keeps process-lifetime allocations
    sample = tmp_path / "pack.data"
    _sample(sample, po)

    def run(*args):
        out = subprocess.run([str(exe), *args], capture_output=True, text=True, env=env, cwd=tmp_path, timeout=300)
        assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
        return out
    assert run().returncode != 0                                            # usage
    assert run(str(tmp_path / "missing.data")).returncode != 0              # the reference segfaults here (main.c:610)
    short = tmp_path / "short.data"
    short.write_text("#c\n5\n1 2 3\n")
    assert run(str(short)).returncode != 0
    out = run(str(sample), "--lx", "96", "--ly", "72", "--gpus", "4")        # 24-row strips, margin > 300 rows
    assert out.returncode != 0 and "narrower than the margin" in out.stderr
    out = run(str(sample), "--lx", "96", "--ly", "72", "--steps", "30")
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0 and "final_density:" in out.stderr, out.stderr[-800:]
    else:
        assert out.returncode != 0 and "create" in out.stderr, out.stderr[-800:]
