"""GPU: the float build of the library (liblbmdem_hip_sp.so; `real` = float, the reference's -DSINGLE_PRECISION mode,
main.c:34-40) against golden vectors dumped from the reference compiled -DSINGLE_PRECISION (tests/golden/*_f32.npz,
made by tests/golden/make_golden.py --f32). Exact equality: the float build mirrors where the reference's C promotes a
sub-expression to double (a `1.` / `4.5` literal, sqrt(), fabs(), the explicit `double fn, ft` of force_grains) and
where it does not. Pinned: populations, obstacle map, hydrodynamic forces, grain kinematics, time-step derivation, the
sample reader, the serial total density. Not offered by the float build: file writers, the write_DEM diagnostics table,
checkpoints, strips."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp(pkg):
    if not os.path.exists(pkg.SP_LIB_PATH):
        pytest.skip("liblbmdem_hip_sp.so not built (make -C 2d-lbm-dem_amd/csrc SP=1)")
    return pkg


class F32Adapter(gu.GpuAdapter):
    """the case's grains as the float build's reader parses the sample text the reference was given"""

    def __init__(self, pkg, po, name, tmp_path):
        c = gu.CASES[name]
        sample = tmp_path / (name + ".data")
        po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"], comment=f"#golden {name}")
        r, x1, x2 = pkg.read_sample(str(sample), "f32")
        assert all(np.array_equal(a, a.astype(np.float32).astype(np.float64)) for a in (r, x1, x2))
        self.sim = pkg.LbmDem(c["lx"], c["ly"], r, x1, x2, precision="f32")


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_float_build_matches_the_single_precision_reference(sp, po, name, tmp_path):
    sim = F32Adapter(sp, po, name, tmp_path)
    g = gu.load(name + "_f32")
    cfg = sim.sim.cfg
    got = [cfg.dx, cfg.dtLB, cfg.dt, cfg.dt2, cfg.c, float(cfg.npDEM)]
    assert np.array_equal(np.array(got), g["scalars"][:6]), "time-step derivation in float"
    res = gu.run_case(sim, name)
    for v in res.values():          # everything the float build hands out is a float
        v = np.asarray(v)
        if v.dtype == np.float64:
            assert np.array_equal(v, v.astype(np.float32).astype(np.float64), equal_nan=True)
    gu.compare(name + "_f32", res, grain_cols=list(range(9)))


def test_float_build_on_the_reference_s_sample_a08d83(sp):
    """bin/a08d83.data @ 600x500, 1 / 10 / 20 coupled steps: SHA-256 of populations, obstacle map, hydrodynamic forces
    and kinematics, and the total density as the float reference's serial chain adds it."""
    name = "real_a08d83_600x500_f32"
    g = gu.load(name)
    lx, ly = 600, 500
    sim = sp.LbmDem(lx, ly, g["r"], g["x1"], g["x2"], precision="f32")
    npdem = int(g["npDEM"])
    assert sim.cfg.npDEM == npdem and sim.cfg.dx == float(g["dx"]) and sim.cfg.c == float(g["c"])
    done = 0
    for k in (1, 10, 20):
        sim.renderScene((k - done) * npdem)
        done = k
        kin = sim.kinematics
        assert np.array_equal(kin[0], g[f"grain0_{k}"]), (k, "grain 0")
        assert gu.sha(kin) == str(g[f"sha_kin_{k}"]), (k, "kinematics")
        assert gu.sha(sim.fhf) == str(g[f"sha_fhf_{k}"]), (k, "hydrodynamic forces")
        assert gu.sha(sim.obst.astype(np.int32)) == str(g[f"sha_obst_{k}"]), (k, "obstacle map")
        assert gu.sha(sim.f) == str(g[f"sha_f_{k}"]), (k, "populations")
        assert sim.final_density() == float(g[f"mass_{k}"]), (k, "total density (a float accumulator)")


def test_float_build_refuses_what_it_does_not_offer(sp, tmp_path):
    sim = sp.LbmDem(64, 48, [0.5e-3], [1.2e-3], [1.1e-3], precision="f32")
    sim.renderScene(3)
    for call in (lambda: sim.write_DEM(str(tmp_path), 0), lambda: sim.checkpoint_save(str(tmp_path / "c")),
                 lambda: sim.dist_enable(0), lambda: sim.grain_table()):
        with pytest.raises(sp.LbmDemError):
            call()
