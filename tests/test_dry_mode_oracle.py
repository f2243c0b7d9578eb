"""CPU: the reference's DEM-only mode -- compiled without `#define _FLUIDE_` (main.c:16): no fluid step (main.c:1709-1719), no
VTK frames (main.c:1768-1770), hydrodynamic forces 0. The oracle's dry stepping against the fixture the reference built that
way wrote (tests/golden/dem_dry_G6_4000steps/, generator tests/golden/make_golden.py --dry) and, where /root/reference
exists, against that build live."""
import os

import numpy as np
import pytest

import golden_util as gu

DRY_DIR = os.path.join(gu.HERE, "golden", "dem_dry_G6_4000steps")
COLS = ("x1 x2 x3 v1 v2 v3 a1 a2 a3 r m It p s f1 f2 ifm M11 M12 M21 M22 z zz fr ice slip rw").split()   # (fm is filled in by write_DEM, not by the step)


def _inputs():
    z = np.load(os.path.join(DRY_DIR, "inputs_and_table.npz"))
    return z["r_mm"] * 1e-3, z["x_mm"] * 1e-3, z["y_mm"] * 1e-3, z["grains"]


def test_oracle_dry_steps_reproduce_the_dry_reference_table(po):
    r, x1, x2, ref_table = _inputs()
    ora = po.Oracle(256, 200, r, x1, x2)
    ora.steps_dry(4000)
    got = ora.get_grains()
    for c in COLS:
        assert np.array_equal(got[:, po.COL[c]], ref_table[:, po.COL[c]]), c
    assert np.all(ora.get_fhf() == 0.0)
    # the fluid is never touched: the lattice still holds init_density's weights (main.c:716-724)
    f = ora.get_f()
    assert np.all(f[..., 0] == 4. / 9) and np.all(f[..., 1] == 1. / 36) and np.all(f[..., 2] == 1. / 9)
    # free fall without buoyancy: the run differs from the coupled one
    wet = np.load(os.path.join(gu.HERE, "golden", "dem_G6_4000steps", "inputs_and_table.npz"))["grains"]
    assert not np.array_equal(wet[:, po.COL["x2"]], ref_table[:, po.COL["x2"]])


def test_oracle_dry_steps_equal_the_live_dry_reference(po, tmp_path):
    if not po.reference_available():
        pytest.skip("/root/reference is not present here")
    import multiprocessing as mp
    r, x1, x2, _ = _inputs()
    sample = tmp_path / "s.data"
    po.write_sample(str(sample), r * 1e3, x1 * 1e3, x2 * 1e3)

    def run(q):      # the reference keeps its state in globals: one build per process
        R = po.Reference(256, 200, str(sample), dry=True)
        R.steps(230)
        q.put((R.get_grains(), R.get_fhf(), R.nbsteps))
    q = mp.Queue()
    p = mp.Process(target=run, args=(q,))
    p.start()
    ref_g, ref_fhf, n = q.get(timeout=120)
    p.join()
    assert n == 230 and np.all(ref_fhf == 0.0)
    rr, xx1, xx2 = po.read_sample(str(sample))
    ora = po.Oracle(256, 200, rr, xx1, xx2)
    ora.steps_dry(230)
    got = ora.get_grains()
    for c in COLS:
        assert np.array_equal(got[:, po.COL[c]], ref_g[:, po.COL[c]]), c
