"""Runs of ordinary DEM sub-steps in ONE launch (k_dem_chain, lbmdem_set_dem_chain) against one launch per sub-step
and against the CPU oracle: the tiles of grains hand their drifted state (main.c:1748-1753) to each other through
tagged cache-line records inside the launch, so every bit has to be the one the per-sub-step kernels produce.

Cases: several tiles with halos staged in LDS; grain indices shuffled so that a tile's partners are spread over the
whole packing (halos larger than the staging: the per-entry reads from memory); the runs cut by fluid steps, list
rebuilds (every 100), the film-law sub-step (8000) and the write_DEM sub-step (4000); the last-contact records the
write_DEM diagnostics start from."""
import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu


def packing(lx, ly, n, seed, shuffle=False):
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    r, x1, x2 = samples.to_metres(r, x, y)
    if shuffle:
        p = np.random.default_rng(seed).permutation(len(r))
        r, x1, x2 = r[p], x1[p], x2[p]
    return r, x1, x2


def kick(sim_list, n, seed, scale=(0.05, 0.05, 30.0)):
    rng = np.random.default_rng(seed)
    k = sim_list[0].kinematics
    k[:, 3:6] = rng.normal(0, 1, (n, 3)) * scale
    for s in sim_list:
        s.kinematics = k
    return k


@pytest.mark.parametrize("tiles", [1, 0])
@pytest.mark.parametrize("shuffle", [False, True])
def test_chain_equals_one_launch_per_substep(pkg, shuffle, tiles):
    """(tiles: the kernel's workgroups hold patches of the packing -- consecutive grains along a space-filling curve, the
    default -- or 64 consecutive indices, lbmdem_set_dem_tiles: the same bits either way, and with any numbering)"""
    lx, ly = 1024, 640
    r, x1, x2 = packing(lx, ly, 2500, 5, shuffle)
    a = pkg.LbmDem(lx, ly, r, x1, x2)
    a.set_dem_tiles(tiles)
    b = pkg.LbmDem(lx, ly, r, x1, x2)
    b.set_dem_chain(0)
    kick([a, b], len(r), 17)
    for n in (37, 1, 12, 200, 95):     # runs of every length, across fluid steps and list rebuilds
        a.renderScene(n); b.renderScene(n)
        assert a.nbsteps == b.nbsteps
        assert np.array_equal(a.kinematics, b.kinematics), (shuffle, a.nbsteps)
        assert np.array_equal(a.grain_pressure, b.grain_pressure), (shuffle, a.nbsteps)
        assert np.array_equal(a.fhf, b.fhf), (shuffle, a.nbsteps)
    assert np.array_equal(a.f, b.f)
    assert np.array_equal(a.obst, b.obst)
    la, sa, slots, resident = a.dem_chain_stats()
    assert resident == slots and la >= 30 and sa >= 300, (la, sa, slots, resident)   # the chain did run
    assert b.dem_chain_stats()[:2] == (0, 0)
    a.close(); b.close()


def test_chain_dem_only_long_runs_match_oracle(pkg, po):
    """run_dem (the reference without _FLUIDE_): runs of 100 sub-steps between list rebuilds, shuffled indices."""
    lx, ly = 512, 384
    r, x1, x2 = packing(lx, ly, 900, 9, shuffle=True)
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    ora = po.Oracle(lx, ly, r, x1, x2)
    k = kick([sim], len(r), 3)
    ora.set_kinematics(k)
    for n in (250, 333):
        sim.run_dem(n); ora.steps_dry(n)
        assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9]), sim.nbsteps
    assert sim.dem_chain_stats()[1] >= 570      # all but a handful of the 583 sub-steps went through the chain
    sim.close()


def test_chain_across_film_and_table_substeps_matches_oracle(pkg, po):
    """sub-steps 3999 (feeds write_DEM: diagnostics, carries resolved from the chain's last-contact records) and 8000
    (film law) cut the runs; the 30-column grain table after 4000 and 8000 sub-steps equals the oracle's."""
    import golden_util as gu
    r, x1, x2 = gu.inputs_m("G4_coupled_256x200")
    sim = pkg.LbmDem(256, 200, r, x1, x2)
    ora = po.Oracle(256, 200, r, x1, x2)
    cols = [po.COL[c] for c in "x1 x2 x3 v1 v2 v3 a1 a2 a3 r m It p s f1 f2 ifm M11 M12 M21 M22 z zz "
                               "fr ice slip rw".split()]
    for n in (4000, 4000, 13):
        sim.renderScene(n); ora.steps(n)
        assert np.array_equal(sim.kinematics, ora.get_grains()[:, :9]), sim.nbsteps
        if sim.nbsteps % 4000 == 0:
            tg, to = sim.grain_table(), ora.get_grains()
            for c in cols:
                assert np.array_equal(tg[:, c], to[:, c]), (sim.nbsteps, c)
    assert np.array_equal(sim.f, ora.get_f())
    sim.close()


@pytest.mark.parametrize("shuffle", [False, True])
def test_rasterisation_by_the_tail_of_a_run_matches_the_rasteriser_and_the_oracle(pkg, po, shuffle):
    """A run that ends where a fluid step begins paints the reduced discs itself (obst_construction, main.c:1009-1032: the
    positions are in the tiles' on-chip memory); lbmdem_set_dem_chain(-1) leaves that to k_obst_paint. Same maps, same f,
    same forces -- against each other and against the oracle; shuffled indices: partners beyond the staged halo."""
    lx, ly = 512, 384
    r, x1, x2 = packing(lx, ly, 900, 13, shuffle)
    a = pkg.LbmDem(lx, ly, r, x1, x2)
    b = pkg.LbmDem(lx, ly, r, x1, x2)
    b.set_dem_chain(-1)
    ora = po.Oracle(lx, ly, r, x1, x2)
    k = kick([a, b], len(r), 23, scale=(0.2, 0.2, 30.0))
    ora.set_kinematics(k)
    n = a.cfg.npDEM
    for m in (2 * n, 5 * n + 3, 7 * n, 9 * n - 3):      # runs that end on and off the fluid steps
        a.renderScene(m); b.renderScene(m); ora.steps(m)
        oa = a.obst
        assert np.array_equal(oa, b.obst) and np.array_equal(oa, ora.get_obst()), (shuffle, a.nbsteps)
        assert np.array_equal(a.fhf, b.fhf) and np.array_equal(a.fhf, ora.get_fhf()), (shuffle, a.nbsteps)
        assert np.array_equal(a.kinematics, ora.get_grains()[:, :9]), (shuffle, a.nbsteps)
    assert np.array_equal(a.f, b.f) and np.array_equal(a.f, ora.get_f())
    assert a.dem_chain_paints() >= 15 and b.dem_chain_paints() == 0, (a.dem_chain_paints(), b.dem_chain_paints())
    a.close(); b.close()


def test_two_handles_on_their_own_streams_side_by_side(pkg):
    """Two launches of the multi-sub-step kernel must not share the GPU (each needs all its tiles resident at once; two
    half-resident ones would wait for each other until their spins run out): the library chains them by an event when
    they come from different streams. Two handles with their own streams, stepped alternately without a synchronisation in
    between, 50 000 grains each (782 of the 1024 workgroup slots: they cannot both be resident)."""
    lx, ly = 4096, 4096
    r, x, y = samples.row_packing(lx, ly, 50000, seed=1234)
    r, x1, x2 = samples.to_metres(r, x, y)
    a = pkg.LbmDem(lx, ly, r, x1, x2); a.use_own_stream()
    b = pkg.LbmDem(lx, ly, r, x1, x2); b.use_own_stream()
    for _ in range(40):
        a.run_dem(23); b.run_dem(23)
    a.sync(); b.sync()                       # (a tile that gave up would raise here)
    assert np.array_equal(a.kinematics, b.kinematics)
    assert a.dem_chain_stats()[3] == a.dem_chain_stats()[2] and a.dem_chain_stats()[0] >= 40
    a.close(); b.close()


def test_streams_that_come_and_go_between_launches(pkg):
    """The launches of the multi-sub-step kernel are chained across streams by an event recorded on the stream of the LAST
    launch when a launch from another stream arrives: a handle that was destroyed, or has changed its stream, in between
    must not be waited for (its stream may be gone). Three handles taking turns, one closed half way, one moved to the
    default stream and back; the survivors stay bit-equal to a handle that ran alone."""
    lx, ly = 768, 512
    r, x1, x2 = packing(lx, ly, 1500, 21)
    ref = pkg.LbmDem(lx, ly, r, x1, x2)
    a = pkg.LbmDem(lx, ly, r, x1, x2); a.use_own_stream()
    b = pkg.LbmDem(lx, ly, r, x1, x2); b.use_own_stream()
    c = pkg.LbmDem(lx, ly, r, x1, x2); c.use_own_stream()
    kick([ref, a, b, c], len(r), 8)
    for k in range(6):
        a.run_dem(17); b.run_dem(17); c.run_dem(17)
    c.close()                                    # the last launch came from c's stream
    for k in range(3):
        a.run_dem(17); b.run_dem(17)
    b.set_stream(None); b.run_dem(17); a.run_dem(17)      # b on the default stream, then back on its own
    b.use_own_stream(); b.run_dem(17); a.run_dem(17)
    ref.run_dem(17 * 11)
    a.sync(); b.sync()
    assert np.array_equal(a.kinematics, ref.kinematics) and np.array_equal(b.kinematics, ref.kinematics)
    a.close(); b.close(); ref.close()


def test_kernel_timing_of_every_nth_launch(pkg):
    """lbmdem_profile_enable(h, N): the HIP events go around every N-th launch of the fused kernel only"""
    lx, ly = 256, 192
    r, x1, x2 = packing(lx, ly, 120, 4)
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    n = sim.cfg.npDEM
    sim.renderScene(2 * n)
    sim.profile_enable(4)
    sim.renderScene(9 * n)
    ms, cnt = sim.profile_read()
    assert cnt == 3 and 0.0 < ms < 5.0, (ms, cnt)          # launches 0, 4 and 8 of the nine
    sim.profile_enable(True)
    sim.renderScene(3 * n)
    assert sim.profile_read()[1] == 3
    sim.profile_enable(False)
    sim.renderScene(n)
    assert sim.profile_read()[1] == 0
    sim.close()


def test_partners_beyond_the_staged_halo_inside_one_xcd_eighth(pkg):
    """A tile reads a partner that is not among its 256 staged halo grains straight from that grain's lines -- from the
    LOCAL copy when the partner's tile belongs to the same XCD eighth: the remote copy only exists for tiles with a partner
    in another eighth. Eight clusters of 512 grains, far apart, each numbered at random within its own index range
    (= one eighth of the 64 tiles): every tile has ~350 distinct partners outside itself, none of them in another eighth."""
    lxc, gap_mm = 600, 2.5
    rs, xs, ys = [], [], []
    y0 = 0.0
    for c in range(8):
        r, x, y = samples.row_packing(lxc, 4000, 512, seed=100 + c, rmin=0.2, rmax=0.3)   # small grains: ~12 list partners each
        assert len(r) == 512
        p = np.random.default_rng(c).permutation(512)
        rs.append(r[p]); xs.append(x[p]); ys.append(y[p] + y0)
        y0 += y.max() + 0.3 + gap_mm
    r, x, y = np.concatenate(rs), np.concatenate(xs), np.concatenate(ys)
    lx, ly = lxc, int(10 * (y0 + 1.0)) // 16 * 16 + 16
    r, x1, x2 = samples.to_metres(r, x, y)
    a = pkg.LbmDem(lx, ly, r, x1, x2)
    b = pkg.LbmDem(lx, ly, r, x1, x2)
    b.set_dem_chain(0)
    kick([a, b], len(r), 31)
    for n in (61, 160, 39):
        a.run_dem(n); b.run_dem(n)
        assert np.array_equal(a.kinematics, b.kinematics), a.nbsteps
    la, sa, slots, resident = a.dem_chain_stats()
    assert resident == slots == 64 and sa >= 250, (la, sa, slots, resident)
    assert a.dem_chain_recoveries() == 0
    # the case is what it says: a tile with more distinct partners outside itself than the 256 staged ones, all in its eighth
    cumul, nbrs, _ = a.verlet()             # the reference's form: pairs (i < j), cumul = running end offsets
    first = np.concatenate([[0], cumul[:-1]])
    counts = np.maximum(cumul - first, 0); counts[-1] = 0
    npairs = int(counts.sum())
    own = np.repeat(np.arange(len(r)), counts)
    pairs = np.concatenate([np.stack([own, nbrs[:npairs]], 1), np.stack([nbrs[:npairs], own], 1)])
    assert np.all(pairs[:, 0] // 512 == pairs[:, 1] // 512)
    off = pairs[pairs[:, 0] // 64 != pairs[:, 1] // 64]
    most = max(len(np.unique(off[off[:, 0] // 64 == t, 1])) for t in range(64))
    assert most > 256, most
    a.close(); b.close()


GIVEUP_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
mode = sys.argv[1]
lx, ly = 1024, 640
r, x, y = samples.row_packing(lx, ly, 2500, seed=5)
r, x1, x2 = samples.to_metres(r, x, y)
a = pkg.LbmDem(lx, ly, r, x1, x2)          # the multi-sub-step kernel, one launch made to give up
b = pkg.LbmDem(lx, ly, r, x1, x2); b.set_dem_chain(0)
k = a.kinematics
k[:, 3:6] = np.random.default_rng(17).normal(0, 1, (len(r), 3)) * (0.05, 0.05, 30.0)
a.kinematics = k; b.kinematics = k
step = (lambda s, n: s.run_dem(n)) if mode == "dem" else (lambda s, n: s.renderScene(n))
a.debug_chain_giveup(2 if mode == "download" else (9 if mode in ("inloop", "inloop_late") else 3))
runs = (37, 1, 12, 200, 95)
if mode == "inloop":                       # LBMDEM_CHAIN_CAP=2: the run loop drains the stream every second launch and finds it itself
    runs = (345,)
if mode == "inloop_late":                  # ... found by the loop of a LATER call than the launch's
    runs = (37, 1, 12, 36, 12, 24, 95)
if mode == "download":                     # found by the first call that is not a run: a download right behind the launch
    step(a, 37); step(b, 37)
    la = a.dem_chain_stats()[0]
    assert la >= 3, la
    assert np.array_equal(a.kinematics, b.kinematics)
    assert a.dem_chain_recoveries() == 1
    runs = (1, 12, 200, 95)
for n in runs:                             # nothing but runs in between: the failed launch is found at the end
    step(a, n); step(b, n)
assert a.nbsteps == b.nbsteps
assert np.array_equal(a.kinematics, b.kinematics)
assert a.dem_chain_recoveries() == 1, a.dem_chain_recoveries()
assert np.array_equal(a.grain_pressure, b.grain_pressure)
if mode != "dem":
    assert np.array_equal(a.fhf, b.fhf) and np.array_equal(a.obst, b.obst) and np.array_equal(a.f, b.f)
# the handle carries on one launch per sub-step, and takes the kernel back when asked to
la0 = a.dem_chain_stats()[0]
step(a, 40); step(b, 40)
assert a.dem_chain_stats()[0] == la0
a.set_dem_chain(128); a.debug_chain_giveup(-1)
step(a, 60); step(b, 60)
assert a.dem_chain_stats()[0] > la0 and a.dem_chain_recoveries() == 1
assert np.array_equal(a.kinematics, b.kinematics)
print("recovered:", mode, "launches", a.dem_chain_stats()[0])
"""


@pytest.mark.parametrize("mode", ["coupled", "dem", "download", "inloop", "inloop_late"])
def test_a_launch_that_gives_up_is_undone_and_the_run_goes_on(mode):
    """k_dem_chain needs all its workgroups resident at once; a launch that gives up (here: made to, in the experiment build,
    half way through its sub-steps) raises the handle's stop word, the kernels queued behind it do nothing, and the library
    goes back to the state before the launch and repeats the calls since with one launch per sub-step: bit-equal to a handle
    that never used the kernel -- fluid, maps, forces and grains, found at the end of a sequence of runs or by a download."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "2d-lbm-dem_amd", "liblbmdem_hip_ab.so")
    if not os.path.exists(lib):
        pytest.skip("the experiment build (make -C 2d-lbm-dem_amd/csrc AB=1) is not there")
    env = dict(os.environ, LBMDEM_HIP_LIBRARY=lib)
    if mode == "inloop": env["LBMDEM_CHAIN_CAP"] = "2"
    if mode == "inloop_late": env["LBMDEM_CHAIN_CAP"] = "5"
    out = subprocess.run([sys.executable, "-c", GIVEUP_SCRIPT, mode], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "recovered: " + mode in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
