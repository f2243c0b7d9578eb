"""The fused kernel reads the previous obstacle map only where it differs from the current one (lbmdem_set_change_mask).

reinit_obst_density (main.c:966-986) re-initialises the nodes a grain has just left or entered from the owner in the map
of the step before; between two fluid steps a few thousand of the 16.8 M nodes of the bench lattice change hands. The
rasterisation at the end of a run of sub-steps (k_dem_chain, in place) leaves one bit per lattice row and 64-column window
of the fused kernel where the picture it paints differs from the one in the OTHER map buffer; the fused kernel takes the
previous owner from the current map in all other rows. A bit that is missing shows as a wrong population at once -- every
case here compares with a library that reads both maps everywhere and with the CPU oracle, bit for bit -- and mode 2
checks every (row, window) pair against the two maps before each use.

Cases: a settling and an agitated packing; grains faster than half a node per fluid step (the conservative path: all rows
of a disc's boxes); reduced discs that overlap (partners hand nodes over: conservative as well); a lattice whose height is
no multiple of the window width; a checkpoint-free fall-back (upload of positions: the first steps after it read both maps)."""
import ctypes

import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu


def packing(lx, ly, n, seed):
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    return samples.to_metres(r, x, y)


def pair(pkg, po, lx, ly, r, x1, x2, vel, seed, reduction=None):
    kw = {}
    if reduction is not None:
        phys = pkg.Physics()
        pkg.load_library().lbmdem_physics_defaults(ctypes.byref(phys))
        phys.reductionR = reduction
        kw["physics"] = phys
    a = pkg.LbmDem(lx, ly, r, x1, x2, **kw); a.set_change_mask(2)
    b = pkg.LbmDem(lx, ly, r, x1, x2, **kw); b.set_change_mask(0)
    ora = po.Oracle(lx, ly, r, x1, x2)
    if reduction is not None:
        ora.set_reduction(reduction)
    rng = np.random.default_rng(seed)
    k = a.kinematics
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * vel
    a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    return a, b, ora


def same(a, b, ora, what):
    fa = a.f
    assert np.array_equal(fa, b.f), f"{what}: f differs from the two-map kernel's at {int((fa != b.f).sum())} values"
    assert np.array_equal(fa, ora.get_f()), f"{what}: f differs from the oracle's"
    assert np.array_equal(a.obst, ora.get_obst()), what
    assert np.array_equal(a.fhf, ora.get_fhf()), what
    assert np.array_equal(a.kinematics, ora.get_grains()[:, :9]), what


@pytest.mark.parametrize("vel", [(0.02, 0.02, 5.0), (0.25, 0.25, 50.0)])
def test_one_map_where_the_bits_are_clear(pkg, po, vel):
    """0.02 m/s: most discs still (no node changes, no bits); 0.25: rings (a few hundredths of a node per fluid step)"""
    lx, ly = 320, 250      # (250 = 4 windows of 60 columns + 10: the last window is cut by the lattice edge)
    r, x1, x2 = packing(lx, ly, 230, 3)
    a, b, ora = pair(pkg, po, lx, ly, r, x1, x2, vel, seed=11)
    n = a.cfg.npDEM
    for k in range(14):
        a.renderScene(3 * n); b.renderScene(3 * n); ora.steps(3 * n)
        same(a, b, ora, f"vel {vel}, step {a.nbsteps}")
    used, hidden = a.change_mask_stats()
    assert hidden == 0, hidden
    assert used >= 30, used            # all but the first steps (no pair list / no picture in both buffers yet)
    assert b.change_mask_stats()[0] == 0
    a.close(); b.close()


def test_grains_faster_than_a_quarter_node_per_step(pkg, po):
    """0.15 ... 0.65 nodes per fluid step (7.5 m/s = one node): beyond half a node from either painted centre a disc takes
    the conservative path (every row of its three boxes); the fastest outrun the pair list now and then (clear + repaint,
    both maps read) -- f, maps and forces stay the oracle's throughout, the bits are verified at every use"""
    lx, ly = 256, 192
    r = np.array([0.7e-3, 0.8e-3, 0.6e-3, 0.9e-3, 0.75e-3, 0.65e-3])
    x1 = np.array([5.0e-3, 12.0e-3, 21.3e-3, 18.0e-3, 22.5e-3, 9.0e-3])      # 0.1 mm per node: lattice = 25.6 x 19.2 mm
    x2 = np.array([5.0e-3, 9.0e-3, 10.0e-3, 15.4e-3, 4.0e-3, 15.0e-3])
    a = pkg.LbmDem(lx, ly, r, x1, x2); a.set_change_mask(2)
    b = pkg.LbmDem(lx, ly, r, x1, x2); b.set_change_mask(0)
    ora = po.Oracle(lx, ly, r, x1, x2)
    k = a.kinematics
    k[:, 3] = [2.4, -3.1, 1.1, 0.0, -4.9, 0.5]
    k[:, 4] = [0.0, 1.2, 0.0, -3.4, 0.0, -2.0]
    a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    n = a.cfg.npDEM
    for s in range(24):
        a.renderScene(n); b.renderScene(n); ora.steps(n)
        same(a, b, ora, f"step {a.nbsteps}")
    used, hidden = a.change_mask_stats()
    assert hidden == 0 and used >= 3, (used, hidden)   # (most steps here clear and repaint: a grain has outrun the list)
    a.close(); b.close()


@pytest.mark.parametrize("reduction", [0.98, 1.08])
def test_overlapping_reduced_discs(pkg, po, reduction):
    lx, ly = 200, 150
    r, x1, x2 = packing(lx, ly, 60, 8)
    a, b, ora = pair(pkg, po, lx, ly, r, x1, x2, (0.4, 0.4, 30.0), seed=2, reduction=reduction)
    n = a.cfg.npDEM
    for k in range(20):
        a.renderScene(2 * n); b.renderScene(2 * n); ora.steps(2 * n)
        same(a, b, ora, f"reduction {reduction}, step {a.nbsteps}")
    used, hidden = a.change_mask_stats()
    # (reductionR >= 1 runs the LDS-tile kernel, which reads both maps: lbm_fused.hip, collide_stream_fills_slots)
    assert hidden == 0 and (used >= 30 if reduction < 1 else used == 0), (used, hidden)
    a.close(); b.close()


def test_upload_of_positions_falls_back_to_both_maps(pkg, po):
    lx, ly = 256, 200
    r, x1, x2 = packing(lx, ly, 120, 5)
    a, b, ora = pair(pkg, po, lx, ly, r, x1, x2, (0.2, 0.2, 20.0), seed=4)
    n = a.cfg.npDEM
    a.renderScene(6 * n); b.renderScene(6 * n); ora.steps(6 * n)
    same(a, b, ora, "before the upload")
    used0 = a.change_mask_stats()[0]
    k = a.kinematics
    k[:, 0] += 3e-4 * np.sin(np.arange(len(r)))        # a third of a node, every grain its own way
    a.kinematics = k; b.kinematics = k; ora.set_kinematics(k)
    a.renderScene(n); b.renderScene(n); ora.steps(n)
    same(a, b, ora, "the step after the upload")
    assert a.change_mask_stats()[0] == used0           # ... read both maps
    a.renderScene(8 * n); b.renderScene(8 * n); ora.steps(8 * n)
    same(a, b, ora, "after the upload")
    used, hidden = a.change_mask_stats()
    assert hidden == 0 and used > used0, (used, used0, hidden)
    a.close(); b.close()


def test_bench_lattice_bits_cover_every_difference(pkg):
    """4096 x 4096, 50 000 grains: 69 windows, rows of 64-row segments and of the tapered tail -- verified per use, and the
    same populations as with both maps read everywhere"""
    lx, ly = 4096, 4096
    r, x, y = samples.row_packing(lx, ly, 50000, seed=1234)
    r, x1, x2 = samples.to_metres(r, x, y)
    a = pkg.LbmDem(lx, ly, r, x1, x2); a.set_change_mask(2)
    b = pkg.LbmDem(lx, ly, r, x1, x2); b.set_change_mask(0)
    n = a.cfg.npDEM
    for k in range(3):
        a.renderScene(6 * n); b.renderScene(6 * n)
        assert np.array_equal(a.f, b.f), a.nbsteps
    used, hidden = a.change_mask_stats()
    assert hidden == 0 and used >= 12, (used, hidden)
    a.close(); b.close()
