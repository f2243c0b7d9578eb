"""GPU: BASELINE.json's full-size configurations.

C3 shape (2048 x 2048, ~7000 grains) and C4 (4096 x 4096, 50 000 grains, the bench workload): the
first fluid step(s) are compared with the CPU oracle exactly (the oracle needs a few seconds per
fluid step at these sizes, its O(N^2) Verlet build ~7 s at 50 k grains), then size-independent
properties are checked over more steps: determinism (two runs, same bits), mass balance, finite and
positive densities, the Verlet pair set, strip-decomposed == single-domain."""
import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu


def packing(lx, ly, n, seed=1234):
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    return samples.to_metres(r, x, y)


def test_c3_2048_7000_grains_two_fluid_steps_exact(pkg, po):
    lx = ly = 2048
    r, x1, x2 = packing(lx, ly, 7000, seed=99)
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    assert sim.fused_work_order()["levels"] == 0          # too few rows for the taper: uniform short segments
    ora = po.Oracle(lx, ly, r, x1, x2)
    n = sim.cfg.npDEM + 1   # fluid steps at nbsteps 0 and npDEM
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.fhf, ora.get_fhf())
    assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9])
    fg = sim.f
    assert np.array_equal(fg, ora.get_f())
    cg, ng, _ = sim.verlet(); co, no, _, _ = ora.verlet()
    assert np.array_equal(cg, co) and np.array_equal(ng, no[:len(ng)])


@pytest.mark.parametrize("lx,ly,n", [(4000, 2046, 12000), (3000, 2790, 12000)])
def test_tapered_work_order_on_row_counts_that_are_no_multiple_of_its_bands(pkg, po, lx, ly, n):
    """Large row ranges run the fused kernel with the tapered work order (eight XCD bands of interleaved 64-row chunks, 64-row
    segments first, 32-, 16- and 8-row segments last: lbm_fused.hip march_plan). With lx = 4000 / 3000 the bands (8 x 512 / 8 x 384
    rows) reach beyond the lattice: items that start past the last row must do nothing, the item that straddles it must stop
    there. Two fluid steps and the sub-steps between them, every array against the CPU oracle."""
    r, x1, x2 = packing(lx, ly, n, seed=7)
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    wo = sim.fused_work_order()
    assert wo["levels"] == 4 and wo["segment_rows"] == [64, 32, 16, 8] and wo["chunk_rows"] == 64
    assert 8 * wo["band_rows"] > lx and sum(wo["level_rows"]) == wo["band_rows"]     # the bands overshoot the lattice
    ora = po.Oracle(lx, ly, r, x1, x2)
    k = sim.cfg.npDEM + 1   # fluid steps at nbsteps 0 and npDEM
    sim.renderScene(k); ora.steps(k)
    assert ora.act_anomalies() == 0
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.f, ora.get_f())
    assert np.array_equal(sim.fhf, ora.get_fhf())
    assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9])


def test_c4_4096_50k_first_step_exact_then_properties(pkg, po):
    lx = ly = 4096
    r, x1, x2 = packing(lx, ly, 50000)
    assert len(r) == 50000
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    assert sim.fused_work_order()["levels"] == 4          # the headline lattice runs the tapered work order
    ora = po.Oracle(lx, ly, r, x1, x2)
    # one renderScene: fluid step + O(N^2) Verlet build + DEM sub-step on the CPU (~10 s)
    sim.renderScene(1); ora.steps(1)
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.fhf, ora.get_fhf())
    assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9])
    cg, ng, _ = sim.verlet(); co, no, _, _ = ora.verlet()
    assert np.array_equal(cg, co) and np.array_equal(ng, no[:len(ng)]) and len(ng) > 50000
    f1 = sim.f
    assert np.array_equal(f1, ora.get_f())
    solid = float((ora.get_obst() >= 0).mean())
    assert 0.25 < solid < 0.45          # same regime as bin/50000.data (34 % solid nodes)
    del ora
    # properties over 4 more fluid steps
    npdem = sim.cfg.npDEM
    sim.renderScene(4 * npdem)
    f5 = sim.f
    k5 = sim.kinematics
    assert np.isfinite(f5).all() and (f5.sum(-1) > 0.5).all()
    m1, m5 = f1.sum(), f5.sum()
    assert abs(m5 - m1) / m1 < 1e-4      # moving walls exchange mass with the fluid; no blow-up
    assert abs(sim.total_density_tree() - m5) <= 1e-10 * m5
    # check_density / final_density: the reference's SERIAL chain (np.cumsum adds strictly in order) -- it drifts away
    # from the true sum by ~1e-9 relative (every addition rounds to the quantum of the running sum), and the HIP path
    # reproduces exactly that number
    serial = 0.0
    for x in range(lx):
        serial = float(np.cumsum(np.concatenate(([serial], f5[x].ravel())))[-1])
    assert sim.final_density() == serial
    assert sim.density_rows_replayed < 40     # the chain took the per-row integer shortcut nearly everywhere
    # determinism: a second simulation gives the same bits
    sim2 = pkg.LbmDem(lx, ly, r, x1, x2)
    sim2.renderScene(1 + 4 * npdem)
    assert np.array_equal(sim2.kinematics, k5)
    out = np.empty_like(f5); sim2.download_f_into(out)
    assert np.array_equal(out, f5)


def test_strips_equal_single_domain_at_1024(pkg):
    """Strip-decomposed (4 strips on one GPU) == single domain, bit for bit, on a lattice large enough
    for many grains to straddle the cuts."""
    import torch
    from strip_backends import LoopbackComm, lockstep_render
    strips = pkg.strips_module()
    lx, ly = 1024, 768
    r, x1, x2 = packing(lx, ly, 3000, seed=7)
    cfg = pkg.derive(lx, ly, r)
    halo = strips.halo_rows(float(r.max()), cfg.dx)
    runners = []
    for rank, strip in enumerate(strips.partition(lx, 4)):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, halo, 0)
        runners.append(strips.StripRunner(be, LoopbackComm(), rank, 4))
    n = 3 * cfg.npDEM + 1
    lockstep_render(runners, n)
    single = pkg.LbmDem(lx, ly, r, x1, x2)
    single.renderScene(n)
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.download_f_into(got)
    assert np.array_equal(got, single.f)
    for R in runners:
        assert np.array_equal(R.b.sim.kinematics, single.kinematics)
        assert np.array_equal(R.b.sim.fhf, single.fhf)


def test_two_strips_of_the_headline_lattice_equal_one_domain(pkg):
    """bench.py --gpus 2 cuts 4096^2 into two strips of 2048 rows: their interior rows (2044, starting at local row 4) are a
    LARGE row range and run the fused kernel's tapered work order with a row offset and bands that overshoot the range. Two
    strips with distributed grains on this GPU (loop-back messages) against one domain, bit for bit, over three periods."""
    import torch
    from strip_backends import LoopbackComm, lockstep_render_dist
    strips = pkg.strips_module()
    lx = ly = 4096
    r, x1, x2 = packing(lx, ly, 50000)
    cfg = pkg.derive(lx, ly, r)
    parts = strips.partition(lx, 2)
    margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
    runners = []
    for rank, strip in enumerate(parts):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin, poison=True)
        runners.append(strips.DistStripRunner(be, LoopbackComm(), rank, 2))
    assert runners[0].b.sim.fused_work_order()["levels"] == 4      # (whole strip; its interior rows take the same order)
    n = 3 * cfg.npDEM + 1
    lockstep_render_dist(runners, n)
    for R in runners:
        R.b.sim.sync()
    single = pkg.LbmDem(lx, ly, r, x1, x2)
    single.renderScene(n)
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.download_f_into(got)
    fs = single.f
    assert np.array_equal(got, fs)
    del got, fs
    ks = single.kinematics
    xc = ks[:, 0] / cfg.dx
    for R, (a, b) in zip(runners, parts):
        own = ((a == 0) | (xc >= a)) & ((b == lx) | (xc < b))
        assert np.array_equal(R.b.sim.kinematics[own], ks[own])
        assert np.array_equal(R.b.sim.fhf[own], single.fhf[own])


def test_c2_1024_fluid_only_rho_u(pkg, po):
    """BASELINE config 2 shape: 1024 x 1024, fluid only. The reference cannot run with 0 grains
    (main.c:220 reads g[0]) and has no lid (its lid terms are commented out, main.c:1125-1130), so the
    case is: one small grain in a corner, all four edges resting walls, f = equilibrium of a smooth
    vortex field u = U (sin 2pi x cos 2pi y, -cos 2pi x sin 2pi y), U = 0.02 in lattice units.
    10 fluid steps; every population, and rho, u, must equal the oracle's."""
    lx = ly = 1024
    r, x1, x2 = np.array([0.5e-3]), np.array([1.2e-3]), np.array([1.1e-3])
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    ora = po.Oracle(lx, ly, r, x1, x2)
    X = (np.arange(lx) / lx)[:, None]; Y = (np.arange(ly) / ly)[None, :]
    U = 0.02
    ux = U * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    uy = -U * np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)
    ex = np.array([0, -1, -1, -1, 0, 1, 1, 1, 0.0]); ey = np.array([0, 1, 0, -1, -1, -1, 0, 1, 1.0])
    w = np.array([4 / 9, 1 / 36, 1 / 9, 1 / 36, 1 / 9, 1 / 36, 1 / 9, 1 / 36, 1 / 9])
    eu = ux[..., None] * ex + uy[..., None] * ey
    f0 = w * (1 + 3 * eu + 4.5 * eu ** 2 - 1.5 * (ux ** 2 + uy ** 2)[..., None])
    sim.f = f0; ora.set_f(f0)
    for _ in range(10):
        sim.lbm_step(); ora.lbm_steps(1)
    fc = ora.get_f()
    assert np.array_equal(sim.f, fc)
    rho, jx, jy = sim.macro()
    assert np.allclose(rho, fc.sum(-1), rtol=1e-14, atol=0)
    assert np.allclose(jx, (fc * ex).sum(-1), rtol=0, atol=1e-15)
    assert np.allclose(jy, (fc * ey).sum(-1), rtol=0, atol=1e-15)
    assert np.abs(jx).max() > 0.5 * U      # the vortex is still there


def test_real_7000_grain_sample_30_coupled_steps_vs_oracle(pkg, po):
    """The reference's a08_a4b4r18_7000.data (6 355 grains, the packing with genuine bounce-back order hazards) on
    2048 x 2048 for 30 coupled steps = 360 DEM sub-steps with three Verlet rebuilds: every bit of f, obst, fhf
    and the kinematics equals the oracle's (which is pinned to the reference on this very sample)."""
    import golden_util as gu
    g = gu.load("real_7000_2048x2048")
    lx = ly = 2048
    sim = pkg.LbmDem(lx, ly, g["r"], g["x1"], g["x2"])
    ora = po.Oracle(lx, ly, g["r"], g["x1"], g["x2"], fast=False)
    n = 30 * int(g["npDEM"])
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9])
    assert np.array_equal(sim.fhf, ora.get_fhf())
    assert np.array_equal(sim.obst, ora.get_obst())
    assert np.array_equal(sim.f, ora.get_f())
    # (most of these fluid steps read the previous obstacle map only in the rows marked as changed: lbmdem_set_change_mask)
    assert sim.change_mask_stats()[0] >= 20, sim.change_mask_stats()


def test_real_50000_grain_sample_six_coupled_steps_vs_oracle(pkg, po):
    """BASELINE.json configs[3] on the reference's own bin/50000.data (49 987 grains, 4096 x 4096): six coupled
    steps (72 DEM sub-steps incl. the film step and a Verlet build) against the oracle, every bit."""
    import golden_util as gu
    g = gu.load("real_50000_4096x4096")
    lx = ly = 4096
    sim = pkg.LbmDem(lx, ly, g["r"], g["x1"], g["x2"])
    ora = po.Oracle(lx, ly, g["r"], g["x1"], g["x2"], fast=False)
    n = 6 * int(g["npDEM"])
    sim.renderScene(n); ora.steps(n)
    assert ora.act_anomalies() == 0
    assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9])
    assert np.array_equal(sim.fhf, ora.get_fhf())
    assert np.array_equal(sim.obst, ora.get_obst())
    got = sim.f
    assert np.array_equal(got, ora.get_f())
    assert sim.change_mask_stats()[0] >= 2, sim.change_mask_stats()   # (the steps after both map buffers hold a picture)


def test_configs4_8192x4096_as_eight_strips_against_the_reference_digests(pkg):
    """BASELINE.json configs[4]: 8192 x 4096, the reference's bin/50000.data, 1-D decomposition into 8 strips with
    the grains distributed and migrated by neighbour messages -- here the 8 ranks live on ONE GPU and are stepped in
    lock-step (a box with 8 GPUs runs the same runner under torch.distributed: tests/test_gpu_multi.py). After one and
    two coupled steps the gathered populations, obstacle map, forces and kinematics hash to what the unmodified
    reference produced on the whole lattice (tests/golden/real_50000_8192x4096.npz)."""
    import torch
    import golden_util as gu
    from strip_backends import LoopbackComm, lockstep_render_dist
    strips = pkg.strips_module()
    g = gu.load("real_50000_8192x4096")
    lx, ly, world = 8192, 4096, 8
    r, x1, x2 = g["r"], g["x1"], g["x2"]
    cfg = pkg.derive(lx, ly, r)
    assert int(g["npDEM"]) == cfg.npDEM
    margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
    parts = strips.partition(lx, world)
    runners = []
    for rank, strip in enumerate(parts):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin)
        runners.append(strips.DistStripRunner(be, LoopbackComm(), rank, world))
    f = np.empty((lx, ly, 9)); obst = np.empty((lx, ly), np.int32)
    done = 0
    for k in (1, 2):
        lockstep_render_dist(runners, (k - done) * cfg.npDEM)
        done = k
        kin = np.empty((len(r), 9)); fhf = np.empty((len(r), 3))
        for R, (a, b) in zip(runners, parts):
            s = R.b.sim
            s.sync()
            s.download_f_into(f)
            s._L.lbmdem_download_obst(s._h, obst.ctypes.data_as(__import__("ctypes").c_void_p))
            kk = s.kinematics
            xc = kk[:, 0] / cfg.dx     # Mgx = 0
            own = ((a == 0) | (xc >= a)) & ((b == lx) | (xc < b))
            kin[own] = kk[own]; fhf[own] = s.fhf[own]
        assert gu.sha(kin) == str(g[f"sha_kin_{k}"]), (k, "kinematics")
        assert gu.sha(fhf) == str(g[f"sha_fhf_{k}"]), (k, "hydrodynamic forces")
        assert gu.sha(obst) == str(g[f"sha_obst_{k}"]), (k, "obstacle map")
        assert gu.sha(f) == str(g[f"sha_f_{k}"]), (k, "populations")


def test_c2_1024_lid_driven_cavity(pkg, po):
    """BASELINE.json configs[1]: 1024 x 1024 lid-driven cavity, fluid only. The reference has no lid as it runs (its
    top-plate terms are commented out, main.c:1129-1130) and cannot run with 0 grains (main.c:220); the case is
    therefore: one small grain in a corner, lattice at rest, and the EXTENSION lbmdem_set_lid enabling exactly the two
    commented-out terms with uw_h = 0.05 (lattice units) -- against the CPU oracle carrying the same two terms
    (not reference-pinned). 40 fluid steps: every population equal, and the lid drags the fluid below it."""
    lx = ly = 1024
    r, x1, x2 = np.array([0.5e-3]), np.array([1.2e-3]), np.array([1.1e-3])
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    ora = po.Oracle(lx, ly, r, x1, x2)
    sim.set_lid(0.05); ora.set_lid(0.05)
    for _ in range(40):
        sim.lbm_step(); ora.lbm_steps(1)
    fc = ora.get_f()
    assert np.array_equal(sim.f, fc)
    rho, jx, jy = sim.macro()
    assert jx[lx // 2, ly - 2] > 1e-3 and abs(jx[lx // 2, ly // 2]) < 1e-6     # a shear layer under the lid only
    # the switch is off by default and off means the reference's bits
    sim0 = pkg.LbmDem(64, 64, r, x1, x2); ora0 = po.Oracle(64, 64, r, x1, x2)
    sim0.set_lid(0.0)
    for _ in range(5):
        sim0.lbm_step(); ora0.lbm_steps(1)
    assert np.array_equal(sim0.f, ora0.get_f())
