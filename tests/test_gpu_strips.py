"""GPU: the strip path of the HIP library (owned-row kernels, halo pack/unpack, ownership masks,
bit-exact force combine) with 2 and 3 strips living on ONE GPU, stepped in lock-step by the same
StripRunner phases that run one-per-GPU under torch.distributed. Must equal the single-domain HIP
run and the CPU oracle bit for bit -- results do not depend on the number of strips."""
import numpy as np
import pytest

import samples
from strip_backends import LoopbackComm, lockstep_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_strips_on_one_gpu_bit_exact(pkg, po, world):
    import torch
    strips = pkg.strips_module()
    lx, ly = 256, 128
    r, x, y = samples.row_packing(lx, ly, 150, seed=13)
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    halo = strips.halo_rows(float(r.max()), cfg.dx)
    rng = np.random.default_rng(2)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.03, 0.03, 20.0]
    runners = []
    for rank, strip in enumerate(strips.partition(lx, world)):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, halo, 0)
        be.sim.kinematics = k
        runners.append(strips.StripRunner(be, LoopbackComm(), rank, world))
    nsteps = 4 * cfg.npDEM + 3
    lockstep_render(runners, nsteps)
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.download_f_into(got)
    ora = po.Oracle(lx, ly, r, x1, x2); ora.set_kinematics(k); ora.steps(nsteps)
    assert np.array_equal(got, ora.get_f())
    for R in runners:
        assert np.array_equal(R.b.sim.kinematics, ora.get_grains()[:, :9])
        assert np.array_equal(R.b.sim.fhf, ora.get_fhf())
    # the reference's total density is one serial chain over 9*lx*ly values: it runs through the strips in x order,
    # every strip continuing from its predecessor's sum -- the same bits as on one domain
    tot = 0.0
    for R in runners:
        tot = R.b.sim.final_density(tot)
    assert tot == ora.total_density()


def test_strip_needs_enough_halo(pkg):
    """Two rows for the fused kernel (the f row and the two obstacle rows it reads beyond a cut); with REPLICATED
    grains the owner of a grain also gathers over its whole footprint: 2 + the largest radius in nodes."""
    with pytest.raises(pkg.LbmDemError):
        pkg.LbmDem(128, 64, [0.8e-3], [3e-3], [3e-3], strip=(0, 64), halo=1)
    sim = pkg.LbmDem(128, 64, [0.8e-3], [3e-3], [3e-3], strip=(0, 64), halo=3)
    sim.obst_construction(); sim.collision_streaming()
    with pytest.raises(pkg.LbmDemError):
        sim.forces_fluid()


def test_split_collide_stream_equals_the_single_launch(pkg):
    """lbmdem_collide_stream_part: EDGES then INTERIOR produce the rows of one lbmdem_collide_stream; the
    lattice may not be read in between."""
    lx, ly = 256, 128
    r, x, y = samples.row_packing(lx, ly, 150, seed=13)
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    halo = pkg.strips_module().halo_rows(float(r.max()), cfg.dx)
    outs = []
    for split in (False, True):
        sim = pkg.LbmDem(lx, ly, r, x1, x2, strip=(64, 160), halo=halo)      # cuts on both sides
        for _ in range(3):
            sim.obst_construction()
            if split:
                sim.collision_streaming_edges()
                with pytest.raises(pkg.LbmDemError):
                    sim.forces_fluid()
                with pytest.raises(pkg.LbmDemError):
                    sim.collision_streaming()
                sim.collision_streaming_interior()
            else:
                sim.collision_streaming()
            sim.forces_fluid()
            sim.run_dem(cfg.npDEM)
        got = np.full((lx, ly, 9), np.nan)
        sim.download_f_into(got)
        outs.append((got[64:160].copy(), sim.fhf, sim.kinematics))
        with pytest.raises(pkg.LbmDemError):
            sim.collision_streaming_interior()          # nothing pending
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert not np.isnan(outs[0][0]).any()


_SELF_EXCHANGE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
strips = ge.load_package().strips_module()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
comm = strips.TorchComm(dist)
send = torch.arange(300000, dtype=torch.float64, device="cuda")
recv = torch.zeros_like(send)
big = torch.randn(2048, 2048, device="cuda")
for it in range(5):
    send.mul_(1.5).add_(it)                       # produced on the main stream just before the exchange
    pending = comm.exchange_begin([(0, send, recv)])
    for _ in range(6):
        big = (big @ big) * 1e-3                  # the "interior rows": main stream stays busy
    comm.exchange_end(pending)
    chk = recv.clone()                            # main stream, ordered after the transfer
    assert torch.equal(chk, send), it
assert comm._lanes["halo"]["side"] is not None
torch.cuda.synchronize()
dist.destroy_process_group()
print("SELF-EXCHANGE-OK")
"""


def test_torchcomm_overlapped_exchange_over_rccl(tmp_path):
    """TorchComm.exchange_begin/end (side stream, event dependency on the producer, current stream waits on
    completion) with the real RCCL backend: a one-rank group sending to itself while the main stream is
    busy. (More than one rank per GPU is refused by RCCL; the two-rank protocol runs under gloo in
    tests/test_strips_gloo.py.)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _SELF_EXCHANGE], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "SELF-EXCHANGE-OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("nstrips", [2, 3])
def test_distributed_grains_equal_single_domain(pkg, nstrips):
    """Strips with the GRAINS distributed (halo of 2 rows, neighbour messages only: kinematics of the margin, link-sum
    tables of the grains a cut goes through, forces of the margin) == one domain, bit for bit, while grains move
    across the cuts. Every rank overwrites the grains it does not integrate with NaN after each sub-step, so any use
    of state that should have come from a neighbour shows."""
    import torch
    from strip_backends import LoopbackComm, lockstep_render_dist
    strips = pkg.strips_module()
    lx, ly = 1024, 256
    r, x, y = samples.row_packing(lx, ly, 700, seed=11)          # fills the lower 19 mm of the 25.6 mm
    parts = pkg.strips_module().partition(lx, nstrips)
    # on every cut, one grain on each side whose disc is clipped by the lattice-interior clamp at the top wall and has
    # links into wall nodes: their link sums are completed (gathered) by whichever rank holds the far end
    for a, _ in parts[1:]:
        for off in (-0.5, 0.56):
            r = np.append(r, 0.5); x = np.append(x, 0.1 * a + off); y = np.append(y, 25.07)
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
    assert min(b - a for a, b in parts) >= margin
    rng = np.random.default_rng(5)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.05, 0.03, 20.0]
    n = 30 * cfg.npDEM + 5        # crosses Verlet rebuilds at sub-steps 100, 200, 300
    for a, _ in parts[1:]:        # the grains nearest to every cut are sent across it
        d = x1 / cfg.dx - a
        for i in (np.argmin(np.where(d >= 0, d, np.inf)), np.argmin(np.where(d < 0, -d, np.inf))):
            k[i, 3] = -np.sign(d[i]) * min(0.9, (abs(d[i]) + 1.5) * cfg.dx / (n * cfg.dt))
    runners = []
    for rank, strip in enumerate(parts):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin,
                                    poison=True)
        be.sim.kinematics = k
        runners.append(strips.DistStripRunner(be, LoopbackComm(), rank, nstrips))
    single = pkg.LbmDem(lx, ly, r, x1, x2)
    single.kinematics = k
    lockstep_render_dist(runners, n)
    single.renderScene(n)
    for R in runners:
        R.b.sim.sync()            # reports table / capacity errors
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.download_f_into(got)
    assert np.array_equal(got, single.f)
    ks, fs = single.kinematics, single.fhf
    xc = ks[:, 0] / cfg.dx
    moved = 0
    for R, (a, b) in zip(runners, parts):
        own = ((a == 0) | (xc >= a)) & ((b == lx) | (xc < b))
        assert own.sum() > 0
        assert np.array_equal(R.b.sim.kinematics[own], ks[own])
        xc0 = x1 / cfg.dx
        moved += int((own & ~(((a == 0) | (xc0 >= a)) & ((b == lx) | (xc0 < b)))).sum())
    assert moved > 0, "no grain changed owner: the test does not exercise migration"


@pytest.mark.parametrize("nstrips", [2, 3])
def test_strip_outputs_are_the_single_domain_files(pkg, tmp_path, nstrips):
    """Drop-in outputs under a strip decomposition (main.c:1767-1776, 237-478): after 4000 renderScene() calls the five
    VTK files (columns merged over the strips), DEM000000.dat, its stats.data line and DEM000000.ps -- written from the
    table sub-step rank 0 ran on a replica merged from every rank's owned grains, with the "previous contact" carries
    resolved over all ranks -- are byte for byte the files of the single-domain run, whose writers are pinned to the
    reference's files (tests/test_gpu_dem_output.py, tests/test_gpu_golden.py)."""
    import torch
    from strip_backends import LoopbackComm, lockstep_render_dist
    strips = pkg.strips_module()
    lx, ly = 1024, 256
    r, x, y = samples.row_packing(lx, ly, 700, seed=11)
    # one grain pressed into the left DEM wall above the packing (a wall carry)
    r = np.append(r, [0.6]); x = np.append(x, [0.6 - 0.002]); y = np.append(y, [22.0])
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    parts = strips.partition(lx, nstrips)
    margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
    assert min(b - a for a, b in parts) >= margin
    rng = np.random.default_rng(8)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.25, 0.2, 40.0]     # ~1 mm in 4000 sub-steps: plenty of collisions
    k[-1:, 3:6] = 0.0
    n = 4000
    single = pkg.LbmDem(lx, ly, r, x1, x2)
    single.kinematics = k
    single.renderScene(n)
    da, db = tmp_path / "single", tmp_path / "strips"
    da.mkdir(); db.mkdir()
    single.write_DEM(str(da), 0); single.write_forces(str(da), 0); single.write_vtk(str(da), 0)
    ts = single.grain_table()
    col = {c: i for i, c in enumerate("x1 x2 x3 v1 v2 v3 a1 a2 a3 r m mw It p s f1 f2 ifm fm fr ifr M11 M12 M21 M22 ice slip rw z zz".split())}
    assert (ts[:, col["z"]] > 0).sum() > 15 and (ts[:, col["slip"]] != 0).sum() >= 3, ((ts[:, col["z"]] > 0).sum(), (ts[:, col["slip"]] != 0).sum())
    runners = []
    for rank, strip in enumerate(parts):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin, poison=True)
        be.sim.kinematics = k
        runners.append(strips.DistStripRunner(be, LoopbackComm(), rank, nstrips))
    lockstep_render_dist(runners, n)
    for R in runners:
        R.b.sim.sync()
    root = runners[0].b.sim
    assert np.array_equal(root.grain_table(), ts)              # all 30 columns, fr / ice / slip / rw included
    root.write_DEM(str(db), 0); root.write_forces(str(db), 0)
    fields = np.zeros(11 * lx * ly, np.float32)
    for R in runners:
        part = np.zeros_like(fields)
        R.b.sim.vtk_place_owned(part)
        fields.view(np.uint32)[:] |= part.view(np.uint32)          # disjoint columns
    pkg.write_vtk_fields(str(db), 0, lx, ly, fields)
    names = sorted(p.name for p in da.iterdir())
    assert len(names) == 8 and names == sorted(p.name for p in db.iterdir())
    for nme in names:
        assert (da / nme).read_bytes() == (db / nme).read_bytes(), nme
    # ... and the run goes on identically after the table sub-step (rank 0 swapped its Verlet list for one sub-step)
    lockstep_render_dist(runners, 2 * cfg.npDEM + 3); single.renderScene(2 * cfg.npDEM + 3)
    ks = single.kinematics
    xc = ks[:, 0] / cfg.dx
    for R, (a, b) in zip(runners, parts):
        own = ((a == 0) | (xc >= a)) & ((b == lx) | (xc < b))
        assert np.array_equal(R.b.sim.kinematics[own], ks[own])


def test_strips_checkpoint_and_restart_mid_period(pkg, tmp_path):
    """Per-rank checkpoints of a strip decomposition with distributed grains: two strips stopped in the MIDDLE of a fluid
    period (and 5 sub-steps before a table sub-step), the carries agreed over the ranks, one file per rank; two new
    handles loaded from the files continue to the same bits as the uninterrupted pair -- lattice, grains, and the
    order-dependent diagnostics of the table sub-step that follows."""
    import torch
    from strip_backends import LoopbackComm, lockstep_render_dist
    strips = pkg.strips_module()
    lx, ly, nstrips = 1024, 256, 2
    r, x, y = samples.row_packing(lx, ly, 700, seed=11)
    r = np.append(r, [0.6]); x = np.append(x, [0.6 - 0.002]); y = np.append(y, [22.0])
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    parts = strips.partition(lx, nstrips)
    margin = strips.default_margin(cfg.npDEM, float(r.max()), cfg.phys.distVerlet, cfg.dx)
    rng = np.random.default_rng(8)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.25, 0.2, 40.0]
    k[-1:, 3:6] = 0.0

    def make(restart=None):
        rs = []
        for rank, strip in enumerate(parts):
            be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, 0, distributed=True, margin=margin,
                                        poison=True, restart_from=None if restart is None else restart[rank])
            if restart is None:
                be.sim.kinematics = k
            rs.append(strips.DistStripRunner(be, LoopbackComm(), rank, nstrips))
        return rs
    n1, n2 = 3994, 4003                   # 3994 % 12 = 10: mid-period; the table sub-step is number 3999
    assert n1 % cfg.npDEM not in (0, cfg.npDEM - 1)
    whole = make()
    lockstep_render_dist(whole, 4000)
    table_whole = whole[0].b.sim.grain_table()
    assert (table_whole[:, 26] != 0).sum() >= 3          # slip: the carry chain is exercised at that sub-step
    lockstep_render_dist(whole, n2 - 4000)
    first = make()
    lockstep_render_dist(first, n1)
    carries = strips.merge_carries([R.b.sim.dist_export_carries() for R in first])
    files = []
    for R in first:
        R.b.sim.dist_set_carries(carries)
        files.append(str(tmp_path / f"ck.rank{R.rank}"))
        R.b.sim.checkpoint_save(files[-1])
    del first
    second = make(restart=files)
    assert all(R.b.sim.nbsteps == n1 for R in second)
    lockstep_render_dist(second, 4000 - n1)
    assert np.array_equal(second[0].b.sim.grain_table(), table_whole)     # fr, ice, slip, rw included: the carries survived
    lockstep_render_dist(second, n2 - 4000)
    for A, B in zip(whole, second):
        A.b.sim.sync(); B.b.sim.sync()
        fa = np.full((lx, ly, 9), np.nan); fb = np.full((lx, ly, 9), np.nan)
        A.b.sim.download_f_into(fa); B.b.sim.download_f_into(fb)
        assert np.array_equal(fa, fb, equal_nan=True)
        assert np.array_equal(A.b.sim.kinematics, B.b.sim.kinematics, equal_nan=True)
        assert np.array_equal(A.b.sim.fhf, B.b.sim.fhf, equal_nan=True)
