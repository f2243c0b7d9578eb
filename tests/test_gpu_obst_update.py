"""The obstacle map updated in place (k_obst_update, lbmdem_set_obst_update): obst_construction (main.c:991-1065) clears
the map and paints every disc again; the HIP path compares each disc's footprint at the centre it was last painted at in
that map buffer with its footprint now and writes only the nodes whose owner changes. The maps -- and with them f, the
hydrodynamic forces and the trajectories -- have to be the ones clear + repaint gives, and the oracle's.

Cases: a packing under gravity (most grains alone: plain stores of the changed nodes); grains fast enough to move more
than a node per fluid step; reduced discs that overlap and separate again (reductionR close to and above 1: the
compare-and-swap hand-over between partners, the overlap flags and lowest-cover records from the partners' disc tests);
grains that leave and enter the lattice; the fall-backs (first step, upload of positions)."""
import ctypes

import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu


def packing(lx, ly, n, seed):
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    return samples.to_metres(r, x, y)


def physics(pkg, reduction):
    phys = pkg.Physics()
    pkg.load_library().lbmdem_physics_defaults(ctypes.byref(phys))
    phys.reductionR = reduction
    return phys


def trio(pkg, po, lx, ly, r, x1, x2, reduction=None, vel=None, seed=1):
    kw = {} if reduction is None else {"physics": physics(pkg, reduction)}
    a = pkg.LbmDem(lx, ly, r, x1, x2, **kw)
    a.set_obst_update(True)
    b = pkg.LbmDem(lx, ly, r, x1, x2, **kw)
    b.set_obst_update(False)
    ora = po.Oracle(lx, ly, r, x1, x2)
    if reduction is not None:
        ora.set_reduction(reduction)
    if vel is not None:
        rng = np.random.default_rng(seed)
        k = a.kinematics
        k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * vel
        a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    return a, b, ora


def check(a, b, ora, what):
    oa = a.obst
    assert np.array_equal(oa, b.obst), f"{what}: map differs from clear + repaint at {int((oa != b.obst).sum())} nodes"
    assert np.array_equal(oa, ora.get_obst()), f"{what}: map differs from the oracle's"
    assert np.array_equal(a.fhf, b.fhf) and np.array_equal(a.fhf, ora.get_fhf()), what
    assert np.array_equal(a.kinematics, b.kinematics) and np.array_equal(a.kinematics, ora.get_grains()[:, :9]), what


@pytest.mark.parametrize("vel", [(0.02, 0.02, 5.0), (0.25, 0.25, 50.0)])
def test_update_equals_clear_and_repaint_and_the_oracle(pkg, po, vel):
    """(0.25 m/s: grains knocking each other about, a few hundredths of a node per fluid step; 0.02: a settling packing)"""
    lx, ly = 320, 240
    r, x1, x2 = packing(lx, ly, 220, 3)
    a, b, ora = trio(pkg, po, lx, ly, r, x1, x2, vel=vel, seed=5)
    n = a.cfg.npDEM
    for k in range(12):
        a.renderScene(3 * n); b.renderScene(3 * n); ora.steps(3 * n)
        check(a, b, ora, f"vel {vel}, step {a.nbsteps}")
    assert np.array_equal(a.f, b.f) and np.array_equal(a.f, ora.get_f())
    up, rep = a.obst_stats()
    assert up >= 33 and rep <= 3, (up, rep)          # the first steps have no list / no picture to start from
    assert b.obst_stats()[0] == 0
    a.close(); b.close()


@pytest.mark.parametrize("reduction", [0.98, 1.0, 1.08])
def test_overlapping_reduced_discs_that_move(pkg, po, reduction):
    """reduced discs of touching grains share nodes (reductionR near / above 1): partners hand nodes over by
    compare-and-swap, `act` needs the lowest-cover records, the force kernel the overlap flags -- all from the update"""
    lx, ly = 200, 150
    r, x1, x2 = packing(lx, ly, 60, 8)
    a, b, ora = trio(pkg, po, lx, ly, r, x1, x2, reduction=reduction, vel=(0.4, 0.4, 30.0), seed=2)
    n = a.cfg.npDEM
    for k in range(10):
        a.renderScene(2 * n + 1); b.renderScene(2 * n + 1); ora.steps(2 * n + 1)
        check(a, b, ora, f"reductionR {reduction}, step {a.nbsteps}")
    assert np.array_equal(a.f, ora.get_f())
    assert a.obst_stats()[0] > 10
    a.close(); b.close()


def test_fast_grains_and_grains_crossing_the_lattice_edge(pkg, po):
    """more than a node per fluid step (the old and the new footprint of a disc do not even overlap), and grains that
    leave the lattice through an edge or come in from outside (no DEM walls there: the fluid domain is a tenth of the box).
    Grains this fast outrun the pair list (more than distVerlet / 2 between two rebuilds): the update notices and the
    library clears and repaints with atomics until the next rebuild -- the maps stay the oracle's throughout."""
    lx, ly = 256, 192
    r = np.array([0.7e-3, 0.8e-3, 0.6e-3, 0.9e-3, 0.75e-3, 0.65e-3])
    x1 = np.array([5.0e-3, 12.0e-3, 24.3e-3, 18.0e-3, 26.5e-3, 9.0e-3])      # 0.1 mm per node: lattice = 25.6 x 19.2 mm
    x2 = np.array([5.0e-3, 9.0e-3, 10.0e-3, 18.4e-3, 4.0e-3, 15.0e-3])
    a, b, ora = trio(pkg, po, lx, ly, r, x1, x2)
    k = a.kinematics
    k[:, 3] = [9.0, -12.0, 6.0, 0.0, -7.0, 0.5]     # m/s; c = 7.5 m/s is one node per fluid step
    k[:, 4] = [0.0, 3.0, 0.0, 8.0, 0.0, -11.0]
    a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    n = a.cfg.npDEM
    for s in range(14):
        a.renderScene(n); b.renderScene(n); ora.steps(n)
        oa = a.obst
        assert np.array_equal(oa, b.obst) and np.array_equal(oa, ora.get_obst()), a.nbsteps
    assert np.array_equal(a.kinematics, ora.get_grains()[:, :9])
    up, rep = a.obst_stats()
    assert up >= 1 and rep >= 8, (up, rep)
    a.close(); b.close()


def test_upload_of_positions_falls_back_and_recovers(pkg, po):
    lx, ly = 160, 120
    r, x1, x2 = packing(lx, ly, 40, 21)
    a, b, ora = trio(pkg, po, lx, ly, r, x1, x2, vel=(0.05, 0.05, 5.0))
    n = a.cfg.npDEM
    a.renderScene(3 * n); b.renderScene(3 * n); ora.steps(3 * n)
    k = a.kinematics
    k[:, 0] += 0.31e-3; k[:, 1] += 0.17e-3           # everything moved by hand: the pair list no longer tracks the positions
    a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    before = a.obst_stats()
    a.lbm_step(); b.lbm_step(); ora.lbm_steps(1)
    assert a.obst_stats()[1] == before[1] + 1        # clear + repaint
    assert np.array_equal(a.obst, ora.get_obst()) and np.array_equal(a.f, ora.get_f())
    a.nbsteps = 0; b.nbsteps = 0; ora.set_nbsteps(0)   # renderScene rebuilds the list at a multiple of 100
    a.renderScene(4 * n); b.renderScene(4 * n); ora.steps(4 * n)
    check(a, b, ora, "after the upload")
    assert a.obst_stats()[0] > before[0]
    a.close(); b.close()
