"""Run under torch.distributed.run with N >= 2 ranks on N GPUs: the real multi-GPU strip path over RCCL against the
CPU oracle. MODE=distributed (default): grains owned by strips, neighbour messages only (kinematics of the margin,
link-sum tables, forces), halo of 2 rows; MODE=ccomm: the same protocol driven from C (lbmdem_comm_run, the
library's RCCL transport); MODE=replicated: every rank integrates all grains, one bit-exact
all-reduce of the forces per fluid step. Used by tests/test_gpu_multi.py when the box has more than one GPU.
SHARED_GPU=1 (MODE=ccomm only): every rank on device 0, control plane over gloo, the library's transport pointed at the
tests' RCCL stand-in (LBMDEM_RCCL_LIBRARY = tests/rccl_shim/librccl.so.1) -- the multi-rank code of lbmdem_comm.hip on a
one-GPU box (tests/test_gpu_rccl_shim.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import __graft_entry__ as ge, samples

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
shared = os.environ.get("SHARED_GPU") == "1"
if shared:
    local = 0
torch.cuda.set_device(local)
if shared:
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
ctl = "cpu" if shared else "cuda"
pkg = ge.load_package(); strips = pkg.strips_module()
mode = os.environ.get("MODE", "distributed")     # distributed | replicated | ccomm (distributed, driven from C)
distributed = mode in ("distributed", "ccomm")
lx, ly = (320 if distributed else 128) * world, 192
r, x, y = samples.row_packing(lx, ly, (230 if distributed else 90) * world, seed=21); r, x1, x2 = samples.to_metres(r, x, y)
rng = np.random.default_rng(4)
k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2; k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.04, 0.03, 15.0]
if mode == "ccomm":     # lbmdem_comm_run: the library's own RCCL transport, as the C host driver uses it
    runner = strips.CCommRunner(pkg, dist, rank, world, local, lx, ly, r, x1, x2)
else:
    runner = strips.make_gpu_runner(pkg, dist, rank, world, local, lx, ly, r, x1, x2, distributed=distributed)
    assert isinstance(runner, strips.DistStripRunner) == distributed
runner.sim.kinematics = k
n = int(os.environ.get("NSTEPS", 5 * runner.sim.cfg.npDEM + 3))
runner.render_scene(n)
runner.sim.sync(); torch.cuda.synchronize()
got = np.full((lx, ly, 9), np.nan)
runner.sim.download_f_into(got)
x0, x1_ = strips.partition(lx, world)[rank]
ok = True
if rank == 0:
    po = ge.load_oracle()
    ora = po.Oracle(lx, ly, r, x1, x2); ora.set_kinematics(k); ora.steps(n)
    ref_f, ref_kin, ref_fhf = (np.ascontiguousarray(a) for a in (ora.get_f(), ora.get_grains()[:, :9], ora.get_fhf()))
else:
    ref_f = np.empty((lx, ly, 9)); ref_kin = np.empty((len(r), 9)); ref_fhf = np.empty((len(r), 3))
for a in (ref_f, ref_kin, ref_fhf):                       # hand the oracle's result to every rank
    t = torch.from_numpy(a).to(ctl); dist.broadcast(t, 0); a[...] = t.cpu().numpy()
own = np.ones(len(r), bool)
if distributed:      # a rank answers for the grains whose centre lies in its rows
    xc = ref_kin[:, 0] / runner.sim.cfg.dx
    own = ((x0 == 0) | (xc >= x0)) & ((x1_ == lx) | (xc < x1_))
parts = (np.array_equal(got[x0:x1_], ref_f[x0:x1_]), np.array_equal(runner.sim.kinematics[own], ref_kin[own]),
         np.array_equal(runner.sim.fhf[own], ref_fhf[own]))
ok = all(parts)
if not ok:
    kin, fh = runner.sim.kinematics, runner.sim.fhf
    bad_k = np.flatnonzero(own & (kin != ref_kin).any(axis=1)); bad_f = np.flatnonzero(own & (fh != ref_fhf).any(axis=1))
    bad_rows = np.flatnonzero((got[x0:x1_] != ref_f[x0:x1_]).any(axis=(1, 2))) + x0
    print(f"[rank {rank}] rows [{x0}, {x1_}): f / kinematics / fhf equal: {parts}; grains with other kinematics {bad_k[:8]} "
          f"(xc {np.round(ref_kin[bad_k[:8], 0] / runner.sim.cfg.dx, 1)}), other fhf {bad_f[:8]}, rows with other f {bad_rows[:8]} .. "
          f"{bad_rows[-3:]} ({len(bad_rows)})", flush=True)
flag = torch.tensor([1 if ok else 0], device=ctl); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("MULTI-GPU-OK" if int(flag) == 1 else "MULTI-GPU-MISMATCH", world, mode, flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
