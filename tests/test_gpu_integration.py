"""GPU: the binding INTEGRATION.md documents, compiled and RUN. oracle/_ref/ref_hip_256x200 is the reference's own
src/main.c -- its main(), reader, console lines, write_DEM and write_forces -- with the five documented edits applied
(oracle/make_integration_check.py, build container only) and linked against liblbmdem_hip.so. Its files after 4000
renderScene() calls must be the files the UNMODIFIED reference wrote (tests/golden/dem_G6_4000steps/), and its
`final_density:` line the oracle's."""
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(gu.HERE), "oracle", "_ref", "ref_hip_256x200")
REF_DIR = os.path.join(gu.HERE, "golden", "dem_G6_4000steps")


@pytest.mark.skipif(not os.path.exists(EXE), reason="integration binary not built (needs /root/reference at build time)")
def test_reference_main_with_the_hip_library_writes_the_reference_s_files(po, tmp_path):
    z = np.load(os.path.join(REF_DIR, "inputs_and_table.npz"))
    sample = tmp_path / "g6.data"
    po.write_sample(str(sample), z["r_mm"], z["x_mm"], z["y_mm"], comment="#Length =25 Height =20")
    out = subprocess.run([EXE, str(sample)], capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert out.returncode == 0, (out.stdout[-400:], out.stderr[-400:])
    n = len(z["r_mm"])
    assert "Nb grains %d" % n in out.stdout or str(n) in out.stdout
    assert "Iteration Number 0, Total density in the system" in out.stdout
    # write_DEM of the REFERENCE on the globals refreshed from the device
    got = open(tmp_path / "DEM000000.dat").read().splitlines()
    want = open(os.path.join(REF_DIR, "DEM000000.dat")).read().splitlines()
    assert got == want
    sg = open(tmp_path / "stats.data").read().splitlines()
    sw = open(os.path.join(REF_DIR, "stats.data")).read().split()
    assert sg[0].startswith("#1_t 2_xfront") and sg[-1].split() == sw
    # write_forces of the reference: every well-defined line (it also prints g[nbgrains], one past the array)
    lines = open(tmp_path / "DEM000000.ps", "rb").read().split(b"\n")
    kept = [lines[0], lines[4]] + lines[5:5 + n] + lines[5 + n + 1:]
    assert kept == open(os.path.join(REF_DIR, "DEM000000.ps"), "rb").read().split(b"\n")
    # final_density: the string the reference's benchmark parses
    r, x1, x2 = po.read_sample(str(sample))
    ora = po.Oracle(256, 200, r, x1, x2)
    ora.steps(4000)
    assert re.search(r"final_density: ([0-9.]+)", out.stderr).group(1) == "%f" % ora.total_density()
