"""x-strip decomposition of the coupled LBM-DEM step across the GPUs of one node, one process per GPU.

The reference has nothing distributed (one address space). The protocol itself lives in the C ABI
(include/lbmdem_hip.h: lbmdem_dist_*, lbmdem_halo_*, lbmdem_comm_*); this module holds the Python drivers over it.

**What runs by default** (bench.py --gpus N, host/lbmdem --gpus N): grains DISTRIBUTED over the strips, neighbour messages
only, no collective on the step path --

* Fluid: rank k owns lattice rows [x_k, x_{k+1}) (x is the reference's slow axis and the slow device axis: a strip is one
  contiguous slab, a lattice row is contiguous) plus a halo of 2 rows on each interior side. The fused kernel produces
  the 2 owned rows next to each cut first; they travel to the neighbour WHILE the interior rows are computed.
* Grains: every rank keeps arrays for all n grains (global index = array index, so Verlet lists, summation order and all
  formulas are those of one GPU) but integrates only the grains whose centre lies in its rows plus a margin of
  `default_margin()` rows on either side -- deep enough that an error at the margin's outer edge cannot reach an owned
  grain within the npDEM sub-steps between two fluid steps. Per fluid step and neighbour four point-to-point messages:
  KIN (kinematics of the owned grains inside the neighbour's margin; a grain that crossed the cut changes owner here --
  this IS the migration), the f halo rows, TABLES (this rank's part of the link-sum tables of grains the neighbour owns)
  and FHF (hydrodynamic forces of the KIN grains). Results are bit-identical for any number of strips.

Drivers of that protocol:

* `CCommRunner` -- what bench.py measures: the whole period in C (`lbmdem_comm_run`, the library's own RCCL transport:
  edge rows + halo exchange on a side stream next to the interior rows, halo rows sent straight out of / into the lattice,
  24 launches per period); torch.distributed only hands out the RCCL ids. No Python on the step path.
* `DistStripRunner` -- the same period written as a Python generator that yields its communication points, driven by
  `TorchComm` (RCCL / gloo through torch.distributed) or, in the single-process tests, by a lock-step driver that copies
  the buffers by hand. ~30 library calls per period from Python: the reference implementation of the protocol and the
  fallback bench.py measures when the C transport's trial run fails.
* `StripRunner` -- the round-1 scheme, kept for strips NARROWER than the margin (where a margin grain could belong to a
  rank that is not a neighbour): grains replicated on every rank, halo of 2 + the largest grain radius rows, one bit-exact
  all-reduce of the hydrodynamic forces per fluid step (owner's bits + zeros, summed as int64).

Backends: `GpuStripBackend` (the HIP library restricted to a strip + torch device tensors as message buffers); the CPU
oracle backends of the tests (tests/strip_backends.py) stand in for it in the world_size-2/3 gloo tests.
"""
from __future__ import annotations

import math


def partition(lx: int, world: int):
    """Row ranges [x_k, x_{k+1}) of `world` strips, as even as possible."""
    return [(k * lx // world, (k + 1) * lx // world) for k in range(world)]


def halo_rows(rmax: float, dx: float) -> int:
    """Rows kept beyond a cut: f halo (1) + obstacle halo (2) + bounding box of the largest grain."""
    return 2 + int(math.ceil(rmax / dx))


class StripRunner:
    """One rank's share of renderScene() (main.c:1697-1765) under x-strip decomposition."""

    def __init__(self, backend, comm, rank: int, world: int):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.has_lo, self.has_hi = rank > 0, rank < world - 1
        self.always_reduce = False  # bench.py --strips with one rank: still go through the collective

    # -- one fluid step, split so that an in-process test can interleave several ranks -------------
    def fluid_compute(self):
        self.b.obst_construction()
        self.b.collision_streaming()

    def halo_post(self):
        """-> list of (peer, send_buffer, recv_buffer)"""
        ops = []
        if self.has_lo:
            ops.append((self.rank - 1, self.b.halo_pack(0), self.b.halo_recv_buffer(0)))
        if self.has_hi:
            ops.append((self.rank + 1, self.b.halo_pack(1), self.b.halo_recv_buffer(1)))
        return ops

    def halo_finish(self):
        if self.has_lo:
            self.b.halo_unpack(0)
        if self.has_hi:
            self.b.halo_unpack(1)

    def forces_post(self):
        self.b.forces_fluid()
        return self.b.fhf_export()

    def forces_finish(self):
        self.b.fhf_import()

    # -- the same with the halo exchange overlapped with the interior rows -----------------------------
    def fluid_edges(self):
        self.b.obst_construction()
        self.b.collision_streaming_edges()

    def fluid_interior(self):
        self.b.collision_streaming_interior()

    def lbm_step(self):
        if self.world > 1:
            self.fluid_edges()
            pending = self.comm.exchange_begin(self.halo_post())   # rows next to the cuts are on their way
            self.fluid_interior()                                  # ... while the bulk is computed
            self.comm.exchange_end(pending)
            self.halo_finish()
        else:
            self.fluid_compute()
        buf = self.forces_post()
        if self.world > 1 or self.always_reduce:
            self.comm.all_reduce_bits(buf)
            self.forces_finish()

    def render_scene(self, n: int = 1):
        """n x renderScene(). The DEM sub-steps between two fluid steps need no communication (the DEM
        state is replicated), so they go down to the backend in one call."""
        b = self.b
        step = b.nbsteps
        while n > 0:
            if step % b.npDEM == 0:               # main.c:1710
                self.lbm_step()
            k = min(n, b.npDEM - step % b.npDEM)
            b.run_dem(k)                          # main.c:1721-1724, 1733-1764, k times
            step += k
            n -= k


class TorchComm:
    """Neighbour exchange + bit-exact all-reduce over torch.distributed (backend "nccl" = RCCL over
    xGMI on the GPU node, "gloo" in the CPU tests)."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        # per lane (a message class that may be in flight at the same time as another): the side stream the
        # transfers are issued on (device tensors only), the event "the packed data is enqueued", the cached P2POps
        self._lanes = {}

    def _p2p(self, ops):
        d = self.dist
        p2p = []
        for peer, send, recv in ops:
            p2p.append(d.P2POp(d.isend, send, peer, self.group))
            p2p.append(d.P2POp(d.irecv, recv, peer, self.group))
        return p2p

    def exchange(self, ops, lane="halo"):
        self.exchange_end(self.exchange_begin(ops, lane))

    def exchange_begin(self, ops, lane="halo"):
        """Start the neighbour transfers; they depend on what has been enqueued so far (the packed edge
        rows) but not on what the caller enqueues next (the interior rows)."""
        st = self._lanes.setdefault(lane, {"side": None, "ready": None, "cache": (None, None)})
        key = tuple((peer, send.data_ptr(), recv.data_ptr()) for peer, send, recv in ops)
        if st["cache"][0] != key:   # the P2POp list is rebuilt only when the buffers change (they do not)
            st["cache"] = (key, self._p2p(ops))
        p2p = st["cache"][1]
        if not p2p:
            return []
        if ops[0][1].is_cuda:
            import torch
            dev = ops[0][1].device
            if st["side"] is None:
                st["side"] = torch.cuda.Stream(device=dev)
                st["ready"] = torch.cuda.Event()
            st["ready"].record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st["side"]):
                st["side"].wait_event(st["ready"])
                return self.dist.batch_isend_irecv(p2p)
        return self.dist.batch_isend_irecv(p2p)

    def exchange_end(self, pending):
        for req in pending:
            req.wait()   # device tensors: the CURRENT stream waits for the transfer, not the host

    def all_reduce_bits(self, int64_tensor):
        self.dist.all_reduce(int64_tensor, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_host_bits(self, array):
        """bitwise merge, in place, of numpy arrays whose non-zero bits are disjoint across the ranks (integer sum)"""
        import numpy as np
        import torch
        raw = array.reshape(-1).view(np.uint8)
        pad = (-raw.size) % 8
        t = torch.from_numpy(np.concatenate([raw, np.zeros(pad, np.uint8)]).view(np.int64).copy())
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        raw[:] = t.cpu().numpy().view(np.uint8)[:raw.size]


KIN, FHF, TABLES = 0, 1, 2   # message kinds (include/lbmdem_hip.h LBMDEM_MSG_*)


def default_margin(npDEM: int, rmax: float, dist_verlet: float, dx: float) -> int:
    """Rows of grains a rank integrates beyond its cuts (lbmdem_dist_default_margin): an error travels one
    Verlet-list edge (centre distance <= 2 r_max + distVerlet) per sub-step, npDEM sub-steps per period."""
    hop = (2 * rmax + dist_verlet) / dx + 1.0
    return int(math.ceil(npDEM * hop + rmax / dx)) + 6


def merge_exports(exports):
    """Combine the ranks' LbmDem.dist_export_owned() results for the sub-step that feeds write_DEM: the states are
    disjoint (every grain has exactly one owner: bitwise OR), per carry the record with the greatest key pair
    (sub-step, kind, grain, partner = the contact the reference evaluated last) wins.
    -> (state12 [n][12], carry_vals [3], carry_has [3])"""
    import numpy as np
    state = np.zeros_like(exports[0][0]).view(np.uint64)
    owners = np.zeros(exports[0][1].shape, np.int32)
    for st, owned, _, _ in exports:
        state |= np.ascontiguousarray(st).view(np.uint64)
        owners += owned
    if not (owners == 1).all():
        raise RuntimeError(f"grains without exactly one owner: {np.flatnonzero(owners != 1)[:8]}")
    vals, has = np.zeros(3), np.zeros(3, np.int32)
    for c in range(3):
        best = (0, 0)
        for _, _, keys, v in exports:
            k = (int(keys[c, 0]), int(keys[c, 1]))
            if k[0] != 0 and k > best:
                best, vals[c], has[c] = k, v[c], 1
    return state.view(np.float64), vals, has


def merge_carries(exports):
    """exports: per rank LbmDem.dist_export_carries() -> the carries every rank installs (dist_set_carries) before the
    ranks write their checkpoints: per carry the youngest record over all ranks, else rank 0's standing value."""
    import numpy as np
    out = np.array(exports[0][2], dtype=np.float64)
    for c in range(3):
        best = (0, 0)
        for keys, vals, _ in exports:
            k = (int(keys[c, 0]), int(keys[c, 1]))
            if k[0] != 0 and k > best:
                best, out[c] = k, vals[c]
    return out


STEP_STROB = 4000     # write_DEM / write_forces cadence (main.c:142, 1773)


class DistStripRunner:
    """One rank's share of renderScene() with the GRAINS distributed over the strips: neighbour-to-neighbour
    messages only (no collective), halo of 2 rows. The period is written as a generator that yields the
    communication points -- ("begin", lane, ops) starts the transfers of `ops` = [(peer, send, recv)], ("end", lane)
    waits for them -- so that the same code is driven by TorchComm (RCCL / gloo) and, in the single-process tests,
    by a lock-step driver that copies the buffers by hand."""

    def __init__(self, backend, comm, rank: int, world: int):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.sides = [s for s, has in ((0, rank > 0), (1, rank < world - 1)) if has]

    def _peer(self, side):
        return self.rank - 1 if side == 0 else self.rank + 1

    def _ops(self, kind):
        sends = self.b.dist_pack_sides(kind, self.sides)           # one launch for both sides
        return [(self._peer(s), sends[s], self.b.dist_recv_buffer(kind, s)) for s in self.sides]

    def period(self):
        b = self.b
        b.dist_begin_period()                      # ownership + message lists from the current positions
        yield ("begin", "kin", self._ops(KIN))     # margin refresh / migration: travels under the whole fluid step
        b.obst_construction()
        b.collision_streaming_edges()
        rows = b.halo_pack_sides(self.sides)
        yield ("begin", "halo", [(self._peer(s), rows[s], b.halo_recv_buffer(s)) for s in self.sides])
        b.collision_streaming_interior()           # ... while the bulk of the rows is computed
        yield ("end", "halo")
        b.halo_unpack_sides(self.sides)
        yield ("begin", "tab", self._ops(TABLES))  # link sums of the grains the neighbours own
        yield ("end", "tab")
        b.dist_unpack_sides(TABLES, self.sides)
        b.forces_fluid()
        yield ("end", "kin")
        b.dist_unpack_sides(KIN, self.sides)
        yield ("begin", "fhf", self._ops(FHF))     # forces of the margin grains, from their owners
        yield ("end", "fhf")
        b.dist_unpack_sides(FHF, self.sides)

    def lbm_step(self):
        pending = {}
        for ev in self.period():
            if ev[0] == "begin":
                pending[ev[1]] = self.comm.exchange_begin(ev[2], lane=ev[1])
            else:
                self.comm.exchange_end(pending.pop(ev[1]))

    def table_substep(self):
        """The sub-step that brings the step counter to a multiple of 4000 (it feeds write_DEM): rank 0 runs it on a
        full replica assembled from every rank's owned grains; the exchange is a bitwise all-reduce of host arrays."""
        import numpy as np
        sim = self.b.sim
        if self.b.nbsteps % self.b.updateVerlet == 0:
            sim.initVerlet()
        st, owned, keys, vals = sim.dist_export_owned()
        allk = np.zeros((self.world, 3, 2), np.int64); allv = np.zeros((self.world, 3))
        allk[self.rank], allv[self.rank] = keys, vals
        for a in (st, owned, allk, allv):
            self.comm.all_reduce_host_bits(a)
        if not (owned == 1).all():
            raise RuntimeError(f"grains without exactly one owner: {np.flatnonzero(owned != 1)[:8]}")
        if self.rank != 0:
            sim.dem_substep()
            return
        vals_best, has = np.zeros(3), np.zeros(3, np.int32)
        for c in range(3):           # per carry, the youngest record over all ranks
            best = (0, 0)
            for r in range(self.world):
                k = (int(allk[r, c, 0]), int(allk[r, c, 1]))
                if k[0] != 0 and k > best:
                    best, vals_best[c], has[c] = k, allv[r, c], 1
        sim.dist_table_substep(st, vals_best, has)

    def render_scene(self, n: int = 1):
        b = self.b
        step = b.nbsteps
        while n > 0:
            if step % b.npDEM == 0:
                self.lbm_step()
            k = min(n, b.npDEM - step % b.npDEM)
            to_table = STEP_STROB - 1 - step % STEP_STROB      # ordinary sub-steps before the next table sub-step
            if to_table == 0:
                self.table_substep()
                k = 1
            else:
                k = min(k, to_table)
                b.run_dem(k)
            step += k
            n -= k


class GpuStripBackend:
    """The HIP library (one LbmDem handle restricted to a strip) + the torch device tensors used as
    exchange buffers. torch is plumbing here: device memory for the buffers and the process group."""

    def __init__(self, pkg, torch, lx, ly, r, x1, x2, strip, halo, device, force_mode=0, distributed=False,
                 margin=0, poison=False, restart_from=None):
        self.torch = torch
        dev = torch.device("cuda", device)
        if restart_from is not None:     # this rank's checkpoint file: the strip comes back as it was (distributed or not)
            self.sim = pkg.LbmDem.checkpoint_load(restart_from, device)
        else:
            self.sim = pkg.LbmDem(lx, ly, r, x1, x2, device=device, strip=strip, halo=halo)
            self.sim.set_force_mode(force_mode)
        self.msg = {}
        if distributed:
            if restart_from is None:
                self.sim.dist_enable(margin)
                if poison:
                    self.sim.dist_set_poison(True)
            for kind in (KIN, FHF, TABLES):
                nd = self.sim.dist_message_doubles(kind)
                self.msg[kind] = ([torch.zeros(nd, dtype=torch.float64, device=dev) for _ in range(2)],
                                  [torch.zeros(nd, dtype=torch.float64, device=dev) for _ in range(2)])
        # everything is enqueued on torch's current stream so that collectives issued through
        # torch.distributed are ordered with the kernels
        self.sim.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        hd = max(self.sim.halo_doubles(), 1)
        self.send = [torch.empty(hd, dtype=torch.float64, device=dev) for _ in range(2)]
        self.recv = [torch.empty(hd, dtype=torch.float64, device=dev) for _ in range(2)]
        self.fhf_buf = torch.empty(3 * self.sim.n, dtype=torch.float64, device=dev)
        self.npDEM = self.sim.cfg.npDEM
        self.updateVerlet = self.sim.cfg.phys.updateVerlet

    @property
    def nbsteps(self):
        return self.sim.nbsteps

    def obst_construction(self): self.sim.obst_construction()
    def collision_streaming(self): self.sim.collision_streaming()
    def collision_streaming_edges(self): self.sim.collision_streaming_edges()
    def collision_streaming_interior(self): self.sim.collision_streaming_interior()
    def forces_fluid(self): self.sim.forces_fluid()
    def initVerlet(self): self.sim.initVerlet()
    def dem_substep(self): self.sim.dem_substep()
    def run_dem(self, k): self.sim.run_dem(k)

    def halo_pack(self, side):
        self.sim.halo_pack(side, self.send[side].data_ptr())
        return self.send[side]

    def halo_recv_buffer(self, side):
        return self.recv[side]

    def halo_unpack(self, side):
        self.sim.halo_unpack(side, self.recv[side].data_ptr())

    def dist_begin_period(self): self.sim.dist_begin_period()

    def _ptrs(self, bufs, sides):
        return [bufs[s].data_ptr() if s in sides else None for s in (0, 1)]

    def dist_pack_sides(self, kind, sides):
        if sides:
            self.sim.dist_pack2(kind, *self._ptrs(self.msg[kind][0], sides))
        return {s: self.msg[kind][0][s] for s in sides}

    def dist_unpack_sides(self, kind, sides):
        if sides:
            self.sim.dist_unpack2(kind, *self._ptrs(self.msg[kind][1], sides))

    def halo_pack_sides(self, sides):
        if sides:
            self.sim.halo_pack2(*self._ptrs(self.send, sides))
        return {s: self.send[s] for s in sides}

    def halo_unpack_sides(self, sides):
        if sides:
            self.sim.halo_unpack2(*self._ptrs(self.recv, sides))

    def dist_pack(self, kind, side):
        self.sim.dist_pack(kind, side, self.msg[kind][0][side].data_ptr())
        return self.msg[kind][0][side]

    def dist_recv_buffer(self, kind, side):
        return self.msg[kind][1][side]

    def dist_unpack(self, kind, side):
        self.sim.dist_unpack(kind, side, self.msg[kind][1][side].data_ptr())

    def fhf_export(self):
        self.sim.fhf_export(self.fhf_buf.data_ptr())
        return self.fhf_buf.view(self.torch.int64)

    def fhf_import(self):
        self.sim.fhf_import(self.fhf_buf.data_ptr())


class _GpuRunner(StripRunner):
    @property
    def sim(self):
        return self.b.sim


class CCommRunner:
    """The same protocol driven entirely from C (lbmdem_comm_run: the library's RCCL transport, as the C host driver
    uses it): one library call per batch of renderScene() calls -- no Python, no torch on the step path. `dist` is
    only used to hand rank 0's RCCL ids to the other ranks."""

    def __init__(self, pkg, dist, rank, world, local_rank, lx, ly, r, x1, x2, connect=True):
        strip = partition(lx, world)[rank]
        self.pkg, self.rank, self.world, self.local_rank = pkg, rank, world, local_rank
        self.sim = pkg.LbmDem(lx, ly, r, x1, x2, device=local_rank, strip=strip, halo=2 if world > 1 else 0)
        self.sim.dist_enable(0)
        self.comm = None
        if connect:
            self.connect(dist)

    def connect(self, dist):
        """the collective part (RCCL communicators): only after every rank is known to have come this far"""
        ids = [self.pkg.comm_unique_id() if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(ids, src=0)
        self.comm = self.pkg.Comm(ids[0], self.rank, self.world, self.local_rank)

    def render_scene(self, n: int = 1):
        self.comm.run(self.sim, n)


class _GpuDistRunner(DistStripRunner):
    @property
    def sim(self):
        return self.b.sim


def make_gpu_runner(pkg, dist, rank, world, local_rank, lx, ly, r, x1, x2, force_mode=0, distributed=None):
    """Build this rank's strip on its GPU (bench.py --gpus N). distributed=None: distribute the grains when the
    strips are wide enough for the margin, else replicate them (all-reduce of the forces)."""
    import torch
    cfg = pkg.derive(lx, ly, r)
    strip = partition(lx, world)[rank]
    margin = default_margin(cfg.npDEM, float(max(r)), cfg.phys.distVerlet, cfg.dx)
    widths = [b - a for a, b in partition(lx, world)]
    if distributed is None:
        distributed = world > 1 and min(widths) >= margin and force_mode == 0
    if distributed:
        backend = GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, 2, local_rank, force_mode, distributed=True,
                                  margin=margin)
        return _GpuDistRunner(backend, TorchComm(dist), rank, world)
    halo = halo_rows(float(max(r)), cfg.dx)
    if world > 1 and strip[1] - strip[0] < halo:
        raise pkg.LbmDemError(-1, f"strips of {strip[1] - strip[0]} rows are narrower than the halo ({halo})")
    backend = GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, halo if world > 1 else 0, local_rank,
                              force_mode)
    runner = _GpuRunner(backend, TorchComm(dist), rank, world)
    runner.always_reduce = world == 1
    return runner
