"""GPU: write_DEM (main.c:340-438) -- DEM%06d.dat and stats.data -- against the files the reference wrote
(tests/golden/dem_G6_4000steps/), and the per-grain contact diagnostics against the oracle (which is
pinned to the reference for all 30 grain fields), including the four that depend on the reference's serial
"previous contact" carries through the contact loop (fr, ice, slip, rw)."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REF_DIR = os.path.join(gu.HERE, "golden", "dem_G6_4000steps")
UNPINNED_DEM_COLS = set()
UNPINNED_STATS_COLS = set()


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
    cols = [po.COL[c] for c in "x1 x2 x3 v1 v2 v3 a1 a2 a3 r m It p s f1 f2 ifm M11 M12 M21 M22 z zz "
                               "fr ice slip rw".split()]
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
    assert (ref_table[:, po.COL["slip"]] != 0).sum() >= 5   # the carry chain is exercised at that step
    with pytest.raises(pkg.LbmDemError):
        sim.renderScene(1); sim.grain_table()


def test_order_dependent_diagnostics_with_walls_and_film(pkg, po):
    """fr, ice, slip, rw against the oracle while grains are pressed into all four walls (bottom and left
    feed `fr` through the list-position-indexed update of main.c:1462/1490, top and left feed `ic`), with the
    film law every 5th sub-step (it credits both partners of a contact) and Verlet rebuilds in between."""
    import ctypes
    import golden_util as gu
    c = gu.ALL_CASES["G5_dem_64x48"]
    r, x1, x2 = gu.inputs_m("G5_dem_64x48")
    phys = pkg.Physics()
    pkg.load_library().lbmdem_physics_defaults(ctypes.byref(phys))
    names = [f[0] for f in pkg.Physics._fields_ if f[1] is ctypes.c_double]
    phys.updateVerlet, phys.stepFilm = 13, 5
    sim = pkg.LbmDem(c["lx"], c["ly"], r, x1, x2, physics=phys)
    ora = po.Oracle(c["lx"], c["ly"], r, x1, x2)
    ora.set_physics([getattr(phys, nme) for nme in names], phys.updateVerlet, phys.stepFilm)
    k = gu.mg.dem_initial_kinematics(c)
    sim.kinematics = k; ora.set_kinematics(k)
    sim.set_diagnostics(True)
    cols = [po.COL[n] for n in "fr ice slip rw p s z".split()]
    seen = {n: 0 for n in "fr ice slip rw".split()}
    for n in (1, 1, 1, 2, 3, 7, 11, 30, 60):
        sim.renderScene(n); ora.steps(n)
        tg, to = sim.grain_table(), ora.get_grains()
        for cc in cols:
            assert np.array_equal(tg[:, cc], to[:, cc]), (sim.nbsteps, cc)
        for nme in seen:
            seen[nme] = max(seen[nme], int((to[:, po.COL[nme]] != 0).sum()))
    assert all(v > 0 for v in seen.values()), seen


@pytest.mark.parametrize("start", [0, 3, 26, 150, 199, 200, 341, 342, 400])
def test_carries_are_exact_from_any_sub_step(pkg, po, start):
    """The "previous contact" carries (pft, pff, pf: main.c:130-131) as the reference holds them after `start` ordinary
    sub-steps, without the diagnostic pipeline having run before: every sub-step leaves per-tile records of its last
    grain / bottom / left / right contact, resolved when the first table sub-step comes. In this case the contacts are
    intermittent: nothing touches in sub-steps 198 and 340, so the table of sub-step 199 / 341 reads carries that
    are dozens of sub-steps old (which the round-1 scheme -- hand-over from the sub-step before -- got wrong)."""
    import ctypes
    c = gu.ALL_CASES["G5_dem_64x48"]
    r, x1, x2 = gu.inputs_m("G5_dem_64x48")
    sim = pkg.LbmDem(c["lx"], c["ly"], r, x1, x2)
    ora = po.Oracle(c["lx"], c["ly"], r, x1, x2)
    k = gu.mg.dem_initial_kinematics(c)
    sim.kinematics = k; ora.set_kinematics(k)
    cols = [po.COL[n] for n in "fr ice slip rw p s z".split()]
    if start:
        sim.renderScene(start); ora.steps(start)
        if start in (199, 341):
            assert ora.get_grains()[:, po.COL["z"]].sum() == 0      # nothing touched in the sub-step before
    sim.set_diagnostics(True)
    for n in (1, 1, 3):
        sim.renderScene(n); ora.steps(n)
        tg, to = sim.grain_table(), ora.get_grains()
        for cc in cols:
            assert np.array_equal(tg[:, cc], to[:, cc]), (sim.nbsteps, cc)
    if start in (199, 341):
        assert (to[:, po.COL["slip"]] != 0).any() or (to[:, po.COL["fr"]] != 0).any()
    # back to ordinary sub-steps, then a table again: records younger than the last table win, older ones do not
    sim.set_diagnostics(False)
    sim.renderScene(17); ora.steps(17)
    sim.set_diagnostics(True)
    sim.renderScene(1); ora.steps(1)
    tg, to = sim.grain_table(), ora.get_grains()
    for cc in cols:
        assert np.array_equal(tg[:, cc], to[:, cc]), (sim.nbsteps, cc)
