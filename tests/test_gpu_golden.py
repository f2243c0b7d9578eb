"""GPU: the HIP path (through the C ABI) directly against the golden vectors dumped from the
unmodified reference. Exact equality."""
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_hip_matches_reference_dump(pkg, name):
    sim = gu.GpuAdapter(pkg, name)
    res = gu.run_case(sim, name)
    gu.compare(name, res, grain_cols=list(range(9)))


@pytest.mark.parametrize("name", sorted(gu.REAL_CASES))
def test_hip_on_the_reference_s_own_samples(pkg, name):
    """The reference's shipped inputs at BASELINE.json's sizes -- a08d83 @ 600x500, a08_a4b4r18_7000 @ 2048^2
    (configs[2]), 50000.data @ 4096^2 (configs[3], 49 987 grains), 50000-test.data @ 3072^2 (configs[0], 47 980
    grains) and 50000.data @ 8192x4096 (configs[4]'s lattice on one GPU) -- through the C ABI: populations, obstacle
    map, hydrodynamic forces and grain kinematics after whole coupled steps hash (SHA-256 of the host-layout
    buffers) to what the unmodified reference produced."""
    class Sim:
        def __init__(self, lx, ly, r, x1, x2): self.s = pkg.LbmDem(lx, ly, r, x1, x2)
        def steps(self, n): self.s.renderScene(n)
    # ... and the total density the reference itself printed for that state (check_density, main.c:1249-1261)
    gu.check_real_case(name, Sim, lambda s: (s.s.f, s.s.obst, s.s.fhf, s.s.kinematics, s.s.final_density()))


def test_vtk_files_byte_identical_to_the_reference(pkg, tmp_path):
    """write_vtk (main.c:237-338): the five binary legacy-VTK files for case G5 after 25 renderScene
    calls must equal, byte for byte, the files the reference wrote (tests/golden/vtk_G5_25steps/)."""
    import os
    sim = gu.GpuAdapter(pkg, "G5_dem_64x48")
    sim.set_kinematics(gu.mg.dem_initial_kinematics(gu.CASES["G5_dem_64x48"]))
    sim.steps(25)
    sim.sim.write_vtk(str(tmp_path), 3)
    ref_dir = os.path.join(gu.HERE, "golden", "vtk_G5_25steps")
    names = sorted(os.listdir(ref_dir))
    assert len(names) == 5
    for name in names:
        got = open(tmp_path / name, "rb").read()
        want = open(os.path.join(ref_dir, name), "rb").read()
        assert got == want, f"{name}: {len(got)} vs {len(want)} bytes, first diff at " \
                            f"{next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)}"


def test_grain_pressure_matches_oracle(pkg, po):
    """g.p (sum of normal contact forces of the last sub-step, main.c:776-777,830,880,912,938)."""
    import numpy as np
    name = "G5_dem_64x48"
    c = gu.CASES[name]
    r, x1, x2 = gu.inputs_m(name)
    sim = pkg.LbmDem(c["lx"], c["ly"], r, x1, x2); ora = po.Oracle(c["lx"], c["ly"], r, x1, x2)
    k = gu.mg.dem_initial_kinematics(c)
    sim.kinematics = k; ora.set_kinematics(k)
    seen = 0
    for n in (1, 1, 23, 100):   # step 0 uses the film law
        sim.renderScene(n); ora.steps(n)
        assert np.array_equal(sim.grain_pressure, ora.get_grains()[:, po.COL["p"]])
        seen = max(seen, int((sim.grain_pressure != 0).sum()))
    assert seen >= 4            # the four grains pressed into the walls
