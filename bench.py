#!/usr/bin/env python3
"""bench.py -- headline benchmark: MLUPS of the coupled 2D LBM-DEM step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path = one fluid (LBM) step -- obstacle rebuild, fused
collide+IBB+stream kernel, hydrodynamic forces -- plus the npDEM DEM sub-steps that belong to it and
the Verlet rebuilds that fall inside (renderScene x npDEM, main.c:1697-1765). Nothing is skipped.

Workload at N=1: BASELINE.json configs[3], 4096 x 4096 lattice with 50 000 grains (a deterministic
synthetic row packing of the same shape as bin/50000.data: radii 0.5-0.9 mm, ~34 % solid nodes;
/root/reference does not exist on the GPU box). At N>1: the SAME lattice is strip-decomposed along x, one
process per GPU (strong scaling of the case BASELINE.json's metric names: "4096^2 / 50k grains; 1/2/4/8
MI355X"). `--workload configs4` runs BASELINE.json configs[4] (8192 x 4096, 50 000 grains) instead.

`python bench.py --gpus N` with N > 1 and no launcher around it starts itself under torch.distributed.run (one rank per
GPU); a node with fewer than N GPUs gets one clean message. At N > 1 the step is driven from C (lbmdem_comm_run: RCCL
send/recv inside the library, no Python on the step path) after a watchdogged trial run in child processes has shown
that this transport works on this node and reproduces the single-domain bits; otherwise the torch.distributed strip
driver is measured instead -- `config.driver` says which one ran.

Before the W warm-up steps the same run advances `--settle` (default 40) further untimed steps, reported as
`config.settle_steps`: a load that starts on an idle GPU sees the chip's power management dip a few milliseconds in (all
kernels ~10 % slower during roughly steps 4-10 of a run, profiles/r04_g_step_trace.jsonl), which a short warm-up would end
in the middle of. The timed region is exactly K steps either way; `--settle 0` switches it off.

Output: one JSON line on rank 0. `value` = lattice-node updates of the whole job per second / 1e6
with all state resident in HBM. `roofline` prices the dominant kernel (k_collide_stream) from HIP
events recorded on its own stream; `cpu_baseline` times the reference's serial C path (prebuilt
oracle/_ref library when it travelled, else this repo's CPU restatement) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before the HIP runtime initialises; see 2d-lbm-dem_amd/__init__.py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
import samples  # noqa: E402

BYTES_PER_LUP = 148.0   # 9x8 B read + 9x8 B write + 4 B obstacle id
# PMC-measured HBM bytes per launch of the fused kernel (scripts/pmc_traffic.sh: rocprofv3 cannot run inside this process).
# The file names the sources the measured library was built from (SHA-256); a number measured on other sources is NOT reported.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def library_source_sha256():
    """SHA-256 over the sources the library is built from (the .so itself is git-ignored and rebuilt by build())"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "2d-lbm-dem_amd", "csrc")
    for name in sorted(os.listdir(d)) + ["../../include/lbmdem_hip.h"]:
        path = os.path.join(d, name)
        if os.path.isfile(path):
            h.update(name.encode()); h.update(open(path, "rb").read())
    return h.hexdigest()
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def workload(which):
    if which == "configs4":
        return dict(name="8192x4096 D2Q9 MRT + 50000 grains (synthetic row packing, seed 1234)",
                    lx=8192, ly=4096, n=50000, fill_lx=8192, short="8192x4096 / 50k grains", data="synthetic")
    if which == "real50k":
        return dict(name="4096x4096 D2Q9 MRT + the reference's bin/50000.data (49987 grains, unscaled)",
                    lx=4096, ly=4096, n=49987, fill_lx=4096, real=True, short="4096^2 / bin/50000.data (49987 grains)",
                    data="the reference's bin/50000.data geometry (fixture tests/golden/real_50000_4096x4096.npz); lattice at rest")
    return dict(name="4096x4096 D2Q9 MRT + 50000 grains (synthetic row packing, seed 1234)",
                lx=4096, ly=4096, n=50000, fill_lx=4096, short="4096^2 / 50k grains", data="synthetic")


def make_sample(w):
    if w.get("real"):   # the reference's own bin/50000.data as its reader parsed it (fixture: tests/golden/make_golden.py)
        g = np.load(os.path.join(ROOT, "tests", "golden", "real_50000_4096x4096.npz"))
        r, x1, x2 = g["r"], g["x1"], g["x2"]
        return (r, x1, x2), (r * 1e3, x1 * 1e3, x2 * 1e3)
    r, x, y = samples.row_packing(w["fill_lx"], w["ly"], w["n"], seed=1234)
    return samples.to_metres(r, x, y), (r, x, y)


def cpu_baseline(w, sample_mm, npdem):
    """Serial C path on this host, 1 core, bounded sample: 2 coupled steps (2 fluid steps, 2*npDEM DEM
    sub-steps, one O(N^2) Verlet build) of the same workload; plus collision_streaming alone."""
    po = ge.load_oracle()
    lx, ly = w["lx"], w["ly"]
    r_mm, x_mm, y_mm = sample_mm
    ref_path = po.ref_lib_path(lx, ly, fast=True)
    out = {"cores": 1, "unit": "MLUPS"}
    nsteps = 2
    if os.path.exists(ref_path):
        import tempfile
        tmp = tempfile.NamedTemporaryFile("w", suffix=".data", delete=False)
        tmp.close()
        po.write_sample(tmp.name, r_mm, x_mm, y_mm)
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        os.dup2(devnull, 1)   # the reference prints per-step lines
        try:
            R = po.Reference(lx, ly, tmp.name, fast=True)
            t0 = time.perf_counter(); R.steps(nsteps * npdem); t1 = time.perf_counter()
            R.collision_streaming(); t2 = time.perf_counter()
        finally:
            import ctypes
            ctypes.CDLL(None).fflush(None)   # the reference's buffered stdout goes to /dev/null too
            os.dup2(saved, 1); os.close(devnull); os.close(saved)
            os.unlink(tmp.name)
        out["kind"] = "reference"
        flags = "-Ofast -march=x86-64-v3 (reference Release flags, CMakeLists.txt:17, portable -march)"
    else:
        r, x1, x2 = r_mm * 1e-3, x_mm * 1e-3, y_mm * 1e-3
        O = po.Oracle(lx, ly, r, x1, x2, fast=True)
        t0 = time.perf_counter(); O.steps(nsteps * npdem); t1 = time.perf_counter()
        O.collision_streaming(); t2 = time.perf_counter()
        out["kind"] = "port"
        flags = "-Ofast -march=native"
    out["value"] = round(1e-6 * lx * ly * nsteps / (t1 - t0), 3)
    out["collide_stream_mlups"] = round(1e-6 * lx * ly / (t2 - t1), 3)
    out["seconds"] = round(t2 - t0, 2)
    out["sample"] = (f"{nsteps} coupled steps ({nsteps} fluid steps + {nsteps * npdem} DEM sub-steps incl. one "
                     f"O(N^2) Verlet build) of the same {lx}x{ly}/{w['n']}-grain input, then one "
                     f"collision_streaming(); serial, {flags}")
    try:
        with open("/proc/cpuinfo") as fp:
            models = [l.split(":", 1)[1].strip() for l in fp if l.startswith("model name")]
        out["cpu"] = models[0] if models else "unknown"
        out["host_cores"] = len(models)
    except OSError:
        pass
    # informative: the same sample with OpenMP on the six loops the reference annotates
    # (main.c:967,996,1007,1077,1294,1327), all host cores, using this repo's restatement built on
    # this host (-Ofast -march=native -fopenmp). Everything else -- edge copies, IBB, swap, stream,
    # the O(N^2) Verlet build, the DEM loops -- is serial in the reference too.
    try:
        r, x1, x2 = r_mm * 1e-3, x_mm * 1e-3, y_mm * 1e-3
        O = po.Oracle(lx, ly, r, x1, x2, fast=True)
        O.set_threads(2)
        t0 = time.perf_counter(); O.steps(nsteps * npdem); t1 = time.perf_counter()
        out["openmp_all_cores"] = {"value": round(1e-6 * lx * ly * nsteps / (t1 - t0), 3), "unit": "MLUPS",
                                   "threads": int(os.environ.get("OMP_NUM_THREADS", out.get("host_cores", 0))),
                                   "kind": "port", "same_sample": True}
    except Exception as e:  # the baseline must never break the bench line
        out["openmp_all_cores"] = {"error": repr(e)[:200]}
    return out


def device_map(world):
    """LBMDEM_BENCH_DEVICES="0,0": the device of every rank, for running several ranks on ONE GPU (the tests do, with
    LBMDEM_RCCL_LIBRARY pointing at tests/rccl_shim -- real RCCL refuses two ranks on one device). Unset: rank k on GPU k."""
    e = os.environ.get("LBMDEM_BENCH_DEVICES")
    if not e:
        return None
    d = [int(t) for t in e.split(",")]
    if len(d) < world:
        raise SystemExit(f"LBMDEM_BENCH_DEVICES names {len(d)} devices for {world} ranks")
    return d


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_under_launcher(args):
    """`python bench.py --gpus N`, N > 1, started without a launcher (as the driver starts it): replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>`."""
    if device_map(args.gpus) is None:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py --gpus {args.gpus}: needs {args.gpus} GPUs, this node has {have}", file=sys.stderr)
            raise SystemExit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


TRIAL_PERIODS = 2


def trial_parent(args, dist, rank, world, ctl):
    """Watchdog around the C driver: every rank starts ONE child process (this script with --trial-child) -- the children
    form their own group, run TRIAL_PERIODS fluid steps of the real workload through lbmdem_comm_run and compare with a
    single-domain run -- and kills it after --trial-timeout seconds. A hang inside RCCL (ordering, topology, a driver
    problem) therefore costs a timeout, not the bench run. -> (every rank's child passed, note)"""
    import subprocess
    import torch
    port = torch.tensor([free_port() if rank == 0 else 0], dtype=torch.int64, device=ctl)
    dist.broadcast(port, 0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port[0])), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=os.environ.get("LOCAL_RANK", str(rank)))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--trial-child", "--gpus", str(world), "--workload", args.workload]
    t0 = time.perf_counter()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=args.trial_timeout)
        ok = child.returncode == 0 and "TRIAL-OK" in out
        why = "passed" if ok else ("exit code %d: %s" % (child.returncode, (err or out).strip().splitlines()[-1][:200] if (err or out).strip() else ""))
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        ok, why = False, f"rank {rank}: no answer within {args.trial_timeout:.0f} s (killed)"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    allok = int(flag[0]) == 1
    note = (f"{TRIAL_PERIODS} fluid steps of this workload through lbmdem_comm_run in child processes, bit-equal to a "
            f"single-domain run (owned grains' kinematics, serial lattice mass): {'passed' if allok else 'FAILED'} in "
            f"{time.perf_counter() - t0:.1f} s" + ("" if allok else f" [{why}]"))
    return allok, note


def trial_child(args, rank, world, device):
    """One rank of the C-driver trial (see trial_parent). Control plane: gloo among the children."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    pkg = ge.load_package()
    strips = pkg.strips_module()
    w = workload(args.workload)
    (r, x1, x2), _ = make_sample(w)
    lx, ly = w["lx"], w["ly"]
    runner = strips.CCommRunner(pkg, dist, rank, world, device, lx, ly, r, x1, x2)
    runner.comm.selftest()
    sim = runner.sim
    nsub = TRIAL_PERIODS * sim.cfg.npDEM
    runner.render_scene(nsub)
    sim.sync()
    # the reference's serial mass chain through the strips in x order, and the kinematics of the grains this rank owns
    s = torch.zeros(1, dtype=torch.float64)
    for k in range(world):
        if k == rank:
            s[0] = sim.final_density(float(s[0]))
        dist.broadcast(s, k)
    kin = sim.kinematics
    ref = torch.zeros((len(r), 9), dtype=torch.float64)
    ref_mass = torch.zeros(1, dtype=torch.float64)
    if rank == 0:
        single = pkg.LbmDem(lx, ly, r, x1, x2, device=device)
        single.renderScene(nsub)
        ref[...] = torch.from_numpy(single.kinematics)
        ref_mass[0] = single.final_density()
        single.close()
    dist.broadcast(ref, 0)
    dist.broadcast(ref_mass, 0)
    refk = ref.numpy()
    x0, x1_ = strips.partition(lx, world)[rank]
    xc = refk[:, 0] / sim.cfg.dx
    own = ((x0 == 0) | (xc >= x0)) & ((x1_ == lx) | (xc < x1_))
    ok = bool(np.array_equal(kin[own], refk[own])) and float(s[0]) == float(ref_mass[0])
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    print("TRIAL-OK" if int(flag[0]) == 1 else f"TRIAL-MISMATCH rank {rank}: own grains equal {bool(np.array_equal(kin[own], refk[own]))}, "
          f"mass {float(s[0])!r} vs {float(ref_mass[0])!r}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    raise SystemExit(0 if int(flag[0]) == 1 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # >= 200 timed steps after 20 warm-up
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=40,
                    help="untimed coupled steps of the same run BEFORE the --warmup steps (reported as config.settle_steps): "
                         "under a load that starts from idle the GPU's power management first dips -- every kernel, fused or "
                         "tiny, runs ~10 %% slower during roughly the 4th to 10th millisecond (profiles/r04_g_step_trace.jsonl) "
                         "-- and a 5-step warm-up ends in the middle of that; 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-mode", type=int, default=0, help="0 = parity hydro-force kernel, 1 = fast")
    ap.add_argument("--workload", choices=["metric", "configs4", "real50k"], default="metric",
                    help="metric = 4096x4096 / 50k-grain synthetic packing (BASELINE.json's metric, every N); "
                         "configs4 = 8192x4096/50k; real50k = 4096x4096 with the reference's bin/50000.data geometry")
    ap.add_argument("--driver", choices=["auto", "c", "torch"], default="auto",
                    help="multi-GPU step driver: c = lbmdem_comm_run (RCCL inside the library, no Python on the step path); "
                         "torch = strips.py over torch.distributed (~30 library calls per period from Python); auto = c after "
                         "a watchdogged trial of it (child processes, killed after --trial-timeout seconds, results compared "
                         "with a single-domain run), else torch")
    ap.add_argument("--trial-timeout", type=float, default=120.0)
    ap.add_argument("--trial-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="f64 = the reference's `real` (the headline); f32 = the float build of the library, the reference's "
                         "-DSINGLE_PRECISION mode (one GPU, its own line: dtype f32, 76 B/LUP; never the headline)")
    ap.add_argument("--dem-chain", type=int, default=None,
                    help="longest run of DEM sub-steps handed to one launch (lbmdem_set_dem_chain; default: the library's, "
                         "128); 0 = one launch per sub-step (A/B)")
    ap.add_argument("--obst-update", type=int, default=None,
                    help="1 = the obstacle map is updated in place (the library's default), 0 = cleared and repainted every step (A/B)")
    ap.add_argument("--change-mask", type=int, default=None,
                    help="1 = the fused kernel reads the previous obstacle map only in the rows the rasterisation marked as "
                         "changed (the library's default), 0 = both maps everywhere (A/B)")
    ap.add_argument("--long-steps", type=int, default=200,
                    help="after the timed region, in the same process: this many more coupled steps of the same run, reported "
                         "as ms_per_step_200 / collide_stream_kernel_ms_200 (SURVEY 8-d's window; skipped when --steps is "
                         "already that long; 0 = none)")
    ap.add_argument("--real-steps", type=int, default=60,
                    help="... and this many coupled steps of the reference's own bin/50000.data geometry on a handle of its "
                         "own (real50k_ms_per_step; metric workload at one GPU only; 0 = none)")
    ap.add_argument("--strips", action="store_true",
                    help="use the strip-decomposition driver (torch.distributed) even with one rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_launcher(args)        # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    devices = device_map(world)
    if devices is None and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"bench.py --gpus {world}: needs {world} GPUs, this node has {torch.cuda.device_count()}", file=sys.stderr)
        raise SystemExit(2)
    shared_gpu = devices is not None          # test mode: several ranks on one GPU (tests/rccl_shim)
    local_rank = devices[rank] if shared_gpu else local_rank
    torch.cuda.set_device(local_rank)
    if args.trial_child:
        return trial_child(args, rank, world, local_rank)
    pkg = ge.load_package()
    w = workload(args.workload)
    (r, x1, x2), sample_mm = make_sample(w)
    lx, ly = w["lx"], w["ly"]

    if world == 1 and not args.strips:
        if args.precision == "f32" and (args.gpus != 1 or args.strips):
            raise SystemExit("--precision f32 is a one-GPU, single-domain mode")
        sim = pkg.LbmDem(lx, ly, r, x1, x2, device=local_rank, precision=args.precision)
        sim.set_force_mode(args.force_mode)
        if args.dem_chain is not None:
            sim.set_dem_chain(args.dem_chain)
        if args.obst_update is not None:
            sim.set_obst_update(bool(args.obst_update))
        if args.change_mask is not None:
            sim.set_change_mask(args.change_mask)
        npdem = sim.cfg.npDEM

        def run_steps(k):
            sim.renderScene(k * npdem)

        def sync():
            sim.sync()
        barrier = lambda: None
        runner = None
        dist_mode = False
    else:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        if shared_gpu:      # RCCL refuses two ranks on one device: the control plane runs over gloo, the data plane is the
            dist.init_process_group("gloo")      # library's own transport (pointed at the tests' stand-in)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        ctl = "cpu" if shared_gpu else "cuda"
        strips = pkg.strips_module()
        cfg0 = pkg.derive(lx, ly, r)
        margin = strips.default_margin(cfg0.npDEM, float(max(r)), cfg0.phys.distVerlet, cfg0.dx)
        wide = min(b - a for a, b in strips.partition(lx, world)) >= margin
        runner = None
        driver_note = None
        want_c = args.driver in ("auto", "c") and args.force_mode == 0 and (wide or world == 1)
        if shared_gpu and not want_c:
            raise SystemExit("several ranks on one GPU: only the C driver (wide strips, parity forces) can run")
        if want_c and args.driver == "auto" and world > 1:
            ok, driver_note = trial_parent(args, dist, rank, world, ctl)
            if not ok:
                if shared_gpu:
                    raise SystemExit("C-driver trial failed: " + driver_note)
                if rank == 0:
                    print("C-driver trial failed (" + driver_note + "): measuring the torch.distributed strip driver", file=sys.stderr)
                want_c = False
        if want_c:
            # grains distributed, the library's own RCCL transport, one C call per batch of steps
            # every rank must take the same path. The vote comes BEFORE the RCCL communicator is made: a rank that
            # fails in its local set-up would otherwise leave the others blocked inside ncclCommInitRank
            ok = 1
            try:
                runner = strips.CCommRunner(pkg, dist, rank, world, local_rank, lx, ly, r, x1, x2, connect=False)
            except Exception as e:
                print(f"[rank {rank}] C driver unavailable: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                if shared_gpu:
                    raise SystemExit("C driver unavailable")
                if rank == 0:
                    print("falling back to the torch.distributed strip driver on all ranks", file=sys.stderr)
                runner = None
            else:
                runner.connect(dist)
                runner.comm.selftest()
        if runner is None:
            runner = strips.make_gpu_runner(pkg, dist, rank, world, local_rank, lx, ly, r, x1, x2,
                                            force_mode=args.force_mode)
        sim = runner.sim
        if args.dem_chain is not None:
            sim.set_dem_chain(args.dem_chain)
        npdem = sim.cfg.npDEM
        dist_mode = isinstance(runner, (strips.DistStripRunner, strips.CCommRunner))

        def run_steps(k):
            runner.render_scene(k * npdem)

        def sync():
            sim.sync(); torch.cuda.synchronize()
        barrier = dist.barrier
        reduce_dev = ctl

    # The driver's own protocol first -- its warm-up, then K timed steps straight away -- reported as ms_per_step_unsettled;
    # then the settle steps and the headline's K timed steps (see --settle for what the difference is).
    ms_unsettled = None
    if args.settle > 0:
        run_steps(args.warmup)
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        run_steps(args.steps)
        sync(); barrier()
        ms_unsettled = 1e3 * (time.perf_counter() - t0) / args.steps
        if world > 1 or args.strips:
            import torch.distributed as dist
            tu = torch.tensor([ms_unsettled], dtype=torch.float64, device=reduce_dev)
            dist.all_reduce(tu, op=dist.ReduceOp.MAX)
            ms_unsettled = float(tu[0])
        run_steps(args.settle)   # clocks settle; the run simply starts its headline steps later
        sync()
    run_steps(args.warmup)
    sync()
    # the box's yardstick: a plain copy of 1.2 GB (2.4 GB of traffic, the fused kernel's algorithmic bytes) on the same stream
    copy_gbs = sim.measure_copy(1200 * 1000 * 1000, 5) if args.precision == "f64" else None
    # HIP events around the fused kernel, on its stream, inside the timed region -- around every 4th launch (every 8th of a
    # long run): the two records of a timed launch hold the next dispatch back, ~10 us per coupled step when every launch
    # is timed (LBMDEM_BENCH_PROFILE=1 does that, =0 times none: the A/B)
    prof_stride = int(os.environ.get("LBMDEM_BENCH_PROFILE", "8" if args.steps >= 100 else "4"))
    sim.profile_enable(prof_stride)
    barrier(); sync()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync(); barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms, launches = sim.profile_read()
    sim.profile_enable(False)
    if world > 1 or args.strips:
        import torch.distributed as dist
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # the state the timed run left behind must be a lattice (checked here, before the informative legs below: those step the
    # DEM and the fluid SEPARATELY -- 100 periods of sub-steps without a fluid step in between -- which leaves a physically
    # inconsistent state by design, and with the reference's own geometry falling under gravity sometimes a non-finite one)
    mass = sim.final_density()
    if world > 1 or args.strips:   # every rank holds the mass of its own rows
        import torch.distributed as dist
        tm = torch.tensor([mass], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
        mass = float(tm[0])
    if not np.isfinite(mass):
        raise SystemExit("non-finite lattice mass after the timed run")
    # informative: the DEM side alone (sub-steps incl. the Verlet rebuilds that fall inside), after the timed region
    dem_only = lbm_only_ms = None
    chain_stats = sim.dem_chain_stats()      # of the timed run (the legs below run on handles of their own)
    if world == 1 and not args.strips:
        # each leg on a FRESH handle: stepping the two sides separately leaves a physically inconsistent (with the
        # reference's own geometry sometimes non-finite) state that nothing else may inherit
        def fresh():
            s2 = pkg.LbmDem(lx, ly, r, x1, x2, device=local_rank, precision=args.precision)
            s2.set_force_mode(args.force_mode)
            if args.dem_chain is not None:
                s2.set_dem_chain(args.dem_chain)
            if args.obst_update is not None:
                s2.set_obst_update(bool(args.obst_update))
            if args.change_mask is not None:
                s2.set_change_mask(args.change_mask)
            return s2
        nsub = 100 * npdem
        leg = fresh()
        leg.renderScene(npdem); leg.run_dem(npdem); leg.sync()
        t0 = time.perf_counter(); leg.run_dem(nsub); leg.sync(); t1 = time.perf_counter()
        dem_only = nsub / (t1 - t0)
        leg.close()
        # ... and the fluid side alone: obstacle map + fused kernel + hydrodynamic forces (SURVEY 8d, item ii)
        nl = 20
        leg = fresh()
        leg.renderScene(npdem); leg.lbm_step(); leg.sync()
        t0 = time.perf_counter()
        for _ in range(nl):
            leg.lbm_step()
        leg.sync(); t1 = time.perf_counter()
        lbm_only_ms = 1e3 * (t1 - t0) / nl
        leg.close()

    # In the same process, after the driver's timed region (its --steps 20 are 17 ms): SURVEY 8-d's window -- 200 coupled
    # steps of the SAME run, the fused kernel timed around every 8th launch (25 launches) ...
    long_leg = None
    if os.environ.get("LBMDEM_BENCH_NO_LEGS"):      # (the profiling scripts: only the timed region's launches in their traces)
        args.long_steps = args.real_steps = 0
    if world == 1 and not args.strips and args.long_steps > 0 and args.steps < args.long_steps:
        sim.profile_enable(8)
        sync()
        t0 = time.perf_counter(); run_steps(args.long_steps); sync(); t1 = time.perf_counter()
        lk_ms, lk_n = sim.profile_read()
        sim.profile_enable(False)
        long_leg = {"steps": args.long_steps, "ms_per_step": 1e3 * (t1 - t0) / args.long_steps, "kernel_ms": lk_ms, "launches_timed": lk_n}
    # ... and the reference's OWN geometry (bin/50000.data as its reader parsed it, 49 987 grains, not row-coherent: 5.8 list
    # entries per grain against the synthetic packing's 3.4), on a handle of its own
    real_leg = None
    real_fixture = os.path.join(ROOT, "tests", "golden", "real_50000_4096x4096.npz")
    if world == 1 and not args.strips and args.workload == "metric" and args.real_steps > 0 and os.path.exists(real_fixture):
        wr = workload("real50k")
        (rr, rx1, rx2), _ = make_sample(wr)
        s3 = pkg.LbmDem(wr["lx"], wr["ly"], rr, rx1, rx2, device=local_rank, precision=args.precision)
        s3.set_force_mode(args.force_mode)
        s3.renderScene(20 * npdem); s3.sync()
        s3.profile_enable(4)
        t0 = time.perf_counter(); s3.renderScene(args.real_steps * npdem); s3.sync(); t1 = time.perf_counter()
        rk_ms, _rk_n = s3.profile_read()
        real_leg = {"steps": args.real_steps, "ms_per_step": 1e3 * (t1 - t0) / args.real_steps, "kernel_ms": rk_ms,
                    "recoveries": s3.dem_chain_recoveries()}
        s3.close()

    # informative: the fast (shuffle-tree) force kernel on the same state: drift against the parity kernel's
    # bits for one and the same lattice, and the coupled-step rate with it (SURVEY hard part 11)
    fast = None
    if world == 1 and not args.strips and args.force_mode == 0 and args.precision == "f64":
        sim.forces_fluid(); fp = sim.fhf.copy()
        sim.set_force_mode(1); sim.forces_fluid(); ff = sim.fhf.copy()
        scale_f = np.abs(fp).max(axis=0)
        drift = float((np.abs(ff - fp).max(axis=0) / np.where(scale_f > 0, scale_f, 1.0)).max())
        if not np.isfinite(drift):   # (the separately stepped state above can be non-finite: see the mass check)
            drift = None
        nf = max(10, min(50, args.steps))
        run_steps(2); sync()
        t0 = time.perf_counter(); run_steps(nf); sync(); t1 = time.perf_counter()
        sim.set_force_mode(0)
        fast = {"ms_per_step": round(1e3 * (t1 - t0) / nf, 4), "mlups": round(1e-6 * lx * ly * nf / (t1 - t0), 1),
                "steps": nf, "max_rel_drift_vs_parity": drift,
                "note": "force_mode=1: the table's link sums reduced across lanes instead of replayed in reference order (last-bit differences); not the headline"}


    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mlups = 1e-6 * lx * ly * args.steps / elapsed
        cfgd = sim.config()
        rows = cfgd.x_end - cfgd.x_begin
        bytes_per_lup = BYTES_PER_LUP if args.precision == "f64" else 76.0   # 9x4 B read + 9x4 B write + 4 B obstacle id
        achieved = bytes_per_lup * rows * ly / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_source = None, None
        if args.gpus == 1 and args.workload == "metric" and os.path.exists(TRAFFIC_FILE) and args.precision == "f64":
            tf = json.load(open(TRAFFIC_FILE))
            if tf.get("source_sha256") == library_source_sha256():       # same workload, same kernel, same sources
                traffic = round(tf["traffic_bytes_per_launch"])
                traffic_source = "measured on a library built from these sources by scripts/pmc_traffic.sh: " + os.path.relpath(TRAFFIC_FILE, ROOT)
            else:
                traffic_source = ("none: " + os.path.relpath(TRAFFIC_FILE, ROOT) + " was measured on another build of the "
                                  "library (" + f"{tf['traffic_bytes_per_launch'] / 1e9:.3f} GB per launch, "
                                  f"{tf.get('traffic_over_algorithmic', 0):.3f} x algorithmic)")
        out = {
            "metric": f"MLUPS (D2Q9 collide+stream, coupled LBM-DEM step) on {w['short']}",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # the same K steps measured straight after the driver's W warm-up steps, earlier in this run (before --settle)
            "ms_per_step_unsettled": round(ms_unsettled, 4) if ms_unsettled is not None else round(ms_per_step, 4),
            "hbm_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
            "data": w["data"],
            "config": {"workload": w["name"], "lx": lx, "ly": ly, "grains": int(len(r)), "npDEM": int(npdem),
                       "step": "1 fluid step + npDEM DEM sub-steps (+ Verlet rebuild every 100 DEM steps)",
                       "force_kernel": "parity" if args.force_mode == 0 else "fast",
                       "dem_chain": dict(zip(("launches", "substeps", "tile_slots", "resident"), chain_stats)),
                       "obst_map": dict(zip(("updated_in_place", "cleared_and_repainted"), sim.obst_stats())),
                       "fused_launches_reading_one_map": (sim.change_mask_stats()[0] if world == 1 and not args.strips else 0),
                       "settle_steps": args.settle,   # untimed steps of this run before the warm-up (GPU clocks; see --settle)
                       "driver": None if runner is None else ("C (lbmdem_comm_run, RCCL send/recv inside the library)" if
                                                                isinstance(runner, strips.CCommRunner) else "torch.distributed (strips.py)"),
                       "driver_trial": None if runner is None else driver_note,
                       "decomposition": "none" if args.gpus == 1 else (
                           f"{args.gpus} x-strips, halo 2 rows, grains owned by strips (margin integrated redundantly); per "
                           f"fluid step and neighbour: f halo rows and grain kinematics (both overlapped with the fluid "
                           f"step), link-sum tables of the grains on the cut, forces of the margin grains -- point to "
                           f"point only" if dist_mode else
                           f"{args.gpus} x-strips; halo exchange overlapped with the interior rows, one bit-exact "
                           f"all-reduce of the hydrodynamic forces per fluid step; grains replicated (strips narrower "
                           f"than the margin)")},
            "dem_steps_per_s": round(args.steps * npdem / elapsed, 1),
            "dem_only_steps_per_s": round(dem_only, 1) if dem_only else None,
            "lbm_step_only_ms": round(lbm_only_ms, 4) if lbm_only_ms else None,
            "lbm_step_only_mlups": round(1e-3 * lx * ly / lbm_only_ms, 1) if lbm_only_ms else None,
            # the same run 200 steps on (fused kernel: mean over 25 timed launches), and the reference's own geometry
            "ms_per_step_200": round(long_leg["ms_per_step"], 4) if long_leg else None,
            "collide_stream_kernel_ms_200": round(long_leg["kernel_ms"], 4) if long_leg else None,
            "launches_timed_200": long_leg["launches_timed"] if long_leg else None,
            "real50k_ms_per_step": round(real_leg["ms_per_step"], 4) if real_leg else None,
            "real50k_mlups": round(1e-3 * lx * ly / real_leg["ms_per_step"], 1) if real_leg else None,
            "real50k_collide_stream_kernel_ms": round(real_leg["kernel_ms"], 4) if real_leg else None,
            "dem_chain_recoveries": sim.dem_chain_recoveries(),
            "collide_stream_kernel_ms": round(kernel_ms, 4),
            "collide_stream_kernel_mlups": round(1e-6 * rows * ly / (kernel_ms * 1e-3), 1) if kernel_ms > 0 else None,
            "roofline": {"bound": "hbm", "kernel": "k_cs_march (fused reinit+collide+IBB+stream)", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the whole coupled step priced the same way: `value` x 148 B / peak (north_star's 40 % is asked of this)
                         "step_frac": round(mlups * 1e6 * bytes_per_lup / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_200": (round(bytes_per_lup * rows * ly / (long_leg["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                      if long_leg and long_leg["kernel_ms"] > 0 else None),
                         "step_frac_200": (round(bytes_per_lup * rows * ly / (long_leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                           if long_leg else None),
                         "traffic": traffic, "traffic_source": traffic_source,
                         # the kernel's REAL traffic rate against what a plain copy moves on this very GPU (null without
                         # counter traffic for this build); and the algorithmic rate against the same yardstick
                         "frac_of_copy": round(traffic / (kernel_ms * 1e-3) / 1e9 / copy_gbs, 4) if (traffic and copy_gbs and kernel_ms > 0) else None,
                         "achieved_over_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
                         "bytes_per_lup": bytes_per_lup, "launches_timed": launches, "timed_every": prof_stride,
                         "note": "achieved = algorithmic 148 B/LUP x lattice nodes per launch / mean HIP-event "
                                 "duration of every `timed_every`-th launch of the timed region; traffic = HBM bytes per launch from rocprofv3 FETCH_SIZE + WRITE_SIZE "
                                 "(separate passes, calibrated on copy kernels) -- reported only when the file was measured on this very library binary; "
                                 "hbm_copy_gbs = the best of four plain-copy shapes on this GPU in this process (8 / 16 B per lane, 2 048 - 8 192 "
                                 "workgroups; the guide's own copy figure for the part is 6.29 TB/s: this yardstick compares boxes, it is "
                                 "not the memory system's ceiling); step_frac / *_200 = the coupled step, and the 200-step leg, priced like frac"},
            "fast_force_mode": fast,
            "total_mass": mass,
        }
        if args.precision == "f32":
            out["metric"] += " -- float build (the reference's -DSINGLE_PRECISION mode), NOT the headline"
            out["roofline"]["note"] = ("achieved = algorithmic 76 B/LUP x lattice nodes per launch / mean HIP-event duration of every `timed_every`-th launch of the timed region; the "
                                       "float build mirrors the reference's mixed float/double arithmetic and gathers the hydrodynamic "
                                       "forces from the lattice (no link-sum table)")
        if args.gpus == 1 and not args.no_cpu_baseline and args.precision == "f64":
            out["cpu_baseline"] = cpu_baseline(w, sample_mm, npdem)
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL prints through C stdio: whatever it buffered comes out BEFORE the line
        print(json.dumps(out), flush=True)
    if world > 1 or args.strips:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
