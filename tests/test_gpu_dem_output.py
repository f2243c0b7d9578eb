"""GPU: write_DEM (main.c:340-438) -- DEM%06d.dat and stats.data -- against the files the reference wrote
(tests/golden/dem_G6_4000steps/), and the per-grain contact diagnostics against the oracle (which is
pinned to the reference for all 30 grain fields). The four fields that depend on the reference's serial
carries through the contact loop (fr, ice, slip, rw) and the statistics summed from them are written as 0
and excluded here."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REF_DIR = os.path.join(gu.HERE, "golden", "dem_G6_4000steps")
UNPINNED_DEM_COLS = {17, 19, 20, 21}          # 0-based: fr, ice, slip, rw
UNPINNED_STATS_COLS = {17, 19, 20, 21}        # WF, INCE, TSLIP, TRW


def _inputs():
    z = np.load(os.path.join(REF_DIR, "inputs_and_table.npz"))
    return z["r_mm"] * 1e-3, z["x_mm"] * 1e-3, z["y_mm"] * 1e-3, z["grains"]


def test_dem_file_and_stats_line_match_the_reference(pkg, tmp_path):
    r, x1, x2, _ = _inputs()
    sim = pkg.LbmDem(256, 200, r, x1, x2)
    sim.renderScene(4000)
    sim.write_DEM(str(tmp_path), 0)
    got = open(tmp_path / "DEM000000.dat").read().splitlines()
    want = open(os.path.join(REF_DIR, "DEM000000.dat")).read().splitlines()
    assert len(got) == len(want) == len(r)
    for lg, lw in zip(got, want):
        cg, cw = lg.split("\t"), lw.split("\t")
        assert len(cg) == len(cw) == 28
        for k in range(28):
            if k not in UNPINNED_DEM_COLS:
                assert cg[k] == cw[k], (cg[0], k, cg[k], cw[k])
    # write_forces (main.c:440-478): every well-defined line of the reference's PostScript file
    sim.write_forces(str(tmp_path), 0)
    got = open(tmp_path / "DEM000000.ps", "rb").read().split(b"\n")
    want = open(os.path.join(REF_DIR, "DEM000000.ps"), "rb").read().split(b"\n")
    assert got[1].startswith(b"%%BoundingBox: ") and got[2].startswith(b"%%Creator") and got[3].startswith(b"%%Title")
    assert [got[0]] + got[4:] == want      # the fixture has the three undefined header lines removed
    assert sum(l.startswith(b"stroke") for l in got) >= 20
    sg = open(tmp_path / "stats.data").read().split()
    sw = open(os.path.join(REF_DIR, "stats.data")).read().split()
    assert len(sg) == len(sw) == 22
    for k in range(22):
        if k not in UNPINNED_STATS_COLS:
            assert sg[k] == sw[k], (k, sg[k], sw[k])


def test_grain_table_matches_reference_dump_and_oracle(pkg, po):
    r, x1, x2, ref_table = _inputs()
    cols = [po.COL[c] for c in "x1 x2 x3 v1 v2 v3 a1 a2 a3 r m It p s f1 f2 ifm M11 M12 M21 M22 z zz".split()]
    sim = pkg.LbmDem(256, 200, r, x1, x2)
    ora = po.Oracle(256, 200, r, x1, x2)
    sim.set_diagnostics(True)
    most = 0
    for n in (1, 1, 1, 22, 75, 300):        # step 0 is a film step; wall contacts in the first steps
        sim.renderScene(n); ora.steps(n)
        tg, to = sim.grain_table(), ora.get_grains()
        for c in cols:
            assert np.array_equal(tg[:, c], to[:, c]), (sim.nbsteps, c)
        most = max(most, int((to[:, po.COL["z"]] > 0).sum()))
    assert most > 10
    sim.set_diagnostics(False)
    sim.renderScene(4000 - sim.nbsteps); ora.steps(4000 - ora.nbsteps)
    tg = sim.grain_table()                   # produced automatically by the sub-step reaching 4000
    for c in cols + [po.COL["fm"]]:
        assert np.array_equal(tg[:, c], ref_table[:, c]), c
    with pytest.raises(pkg.LbmDemError):
        sim.renderScene(1); sim.grain_table()
