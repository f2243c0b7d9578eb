"""GPU: the library's own RCCL transport (2d-lbm-dem_amd/csrc/lbmdem_comm.hip) with SEVERAL RANKS ON THE ONE GPU of the
test box. Real RCCL refuses two ranks on one device, so these tests point the library (LBMDEM_RCCL_LIBRARY) at
tests/rccl_shim/librccl.so.1 -- a test-only stand-in that implements the nine RCCL entry points the library uses between
processes sharing a GPU, stream-ordered like the real thing (see the header of tests/rccl_shim/rccl_shim.hip). What runs is
the product's multi-rank code: neighbour flags true, four communicators in flight on side and main streams, grain
migration through the kinematics message, the table sub-step over the ranks, VTK frames gathered on rank 0."""
import os
import re
import subprocess
import time
import sys
import textwrap

import numpy as np
import pytest

import samples

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "2d-lbm-dem_amd", "host", "lbmdem")
SHIM = os.path.join(ROOT, "tests", "rccl_shim", "librccl.so.1")


def shim_env(**extra):
    assert os.path.exists(SHIM), "tests/rccl_shim/librccl.so.1 not built (__graft_entry__.build())"
    env = dict(os.environ, LBMDEM_RCCL_LIBRARY=SHIM, RCCL_SHIM_TIMEOUT_S="30")
    env.update(extra)
    return env


def spawn_ranks(code, world, tmp_path, env, timeout=300):
    """`code` in `world` python processes (RANK, WORLD, IDFILE in the environment); -> their outputs"""
    idfile = str(tmp_path / "rccl_id")
    procs = [subprocess.Popen([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT,
                              env=dict(env, RANK=str(k), WORLD=str(world), IDFILE=idfile),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


PRELUDE = '''
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
rank, world, idfile = int(os.environ["RANK"]), int(os.environ["WORLD"]), os.environ["IDFILE"]
if rank == 0:
    uid = pkg.comm_unique_id()
    open(idfile + ".tmp", "wb").write(uid); os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
comm = pkg.Comm(uid, rank, world, 0)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_transport_selftest_with_several_ranks_on_one_gpu(world, tmp_path):
    """lbmdem_comm_selftest with world > 1: on every lane one grouped exchange with both neighbours, all four lanes in
    flight at once, payload naming sender and lane -- once with a small message, once with one of several chunks -- and
    the host all-reduce behind check_density / the merged outputs."""
    code = PRELUDE + '''
comm.selftest(4096)
comm.selftest(300000)        # 2.4 MB per message: several rounds through the stand-in's 128 KB slots
s = comm.allreduce_sum(np.array([rank + 1.0, 10.0 * (rank + 1)]))
assert s[0] == world * (world + 1) / 2 and s[1] == 10 * s[0], s
print("SELFTEST-OK")
'''
    for rc, out, err in spawn_ranks(code, world, tmp_path, shim_env()):
        assert rc == 0 and "SELFTEST-OK" in out, (out[-300:], err[-1500:])


def test_misordered_exchange_fails_the_test_instead_of_hanging(tmp_path):
    """The stand-in is stream-ordered with one slot per direction, like RCCL's FIFOs: ranks that issue their exchanges in
    DIFFERENT orders wait for each other. That must surface as an error within the stand-in's timeout -- in the test, not
    as a hang of some later multi-GPU run -- and must leave the GPU usable. Here rank 0 sends twice on lane A and then on
    lane B on one stream, while rank 1 receives B first."""
    code = '''
import ctypes as C, os, sys, time
import torch
shim = C.CDLL(os.environ["LBMDEM_RCCL_LIBRARY"])
shim.ncclGetErrorString.restype = C.c_char_p
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
rank, idfile = int(os.environ["RANK"]), os.environ["IDFILE"]
ids = (Uid * 2)()
if rank == 0:
    for k in range(2):
        assert shim.ncclGetUniqueId(C.byref(ids[k])) == 0
    open(idfile + ".tmp", "wb").write(bytes(ids)); os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    C.memmove(ids, open(idfile, "rb").read(), 256)
torch.cuda.set_device(0)
buf = torch.zeros(1024, dtype=torch.float64, device="cuda")
comms = [C.c_void_p(), C.c_void_p()]
for k in range(2):
    assert shim.ncclCommInitRank(C.byref(comms[k]), 2, ids[k], rank) == 0, shim.ncclGetErrorString(1)
A, B = comms
args = lambda c: (C.c_void_p(buf.data_ptr()), C.c_size_t(1024), C.c_int(8), C.c_int(1 - rank), c, C.c_void_p(0))
if rank == 0:
    rcs = [shim.ncclSend(*args(A)), shim.ncclSend(*args(A)), shim.ncclSend(*args(B))]
else:
    rcs = [shim.ncclRecv(*args(B)), shim.ncclRecv(*args(A)), shim.ncclRecv(*args(A))]
assert rcs == [0, 0, 0], rcs
t0 = time.time()
torch.cuda.synchronize()                       # returns: the waiting kernels give up after RCCL_SHIM_TIMEOUT_S
waited = time.time() - t0
# (whichever wait gave up first raised the error word of ITS communicator -- and thereby let the other one through)
rcs = [shim.ncclSend(*args(c)) if rank == 0 else shim.ncclRecv(*args(c)) for c in (A, B)]
bad = [rc for rc in rcs if rc != 0]
assert bad and b"wrong order" in shim.ncclGetErrorString(bad[0]), (rcs, waited)
assert float((buf + 1).sum()) == 1024.0        # the GPU still works
print("MISORDER-DETECTED %.1f" % waited)
'''
    outs = spawn_ranks(code, 2, tmp_path, shim_env(RCCL_SHIM_TIMEOUT_S="4"), timeout=120)
    for rc, out, err in outs:
        assert rc == 0 and "MISORDER-DETECTED" in out, (out[-300:], err[-1500:])
    assert max(float(re.search(r"MISORDER-DETECTED ([0-9.]+)", o).group(1)) for _, o, _ in outs) > 2.0


def cut_sample(lx, ly, n, seed):
    """a packing at rest + two overlapping grains in the free space above it, the right one 3 um left of the cut between
    two strips: their contact pushes it across the cut within a few sub-steps (a migration through the kinematics message)"""
    r, x, y = samples.row_packing(lx, ly, n, seed=seed)
    assert y.max() + 0.9 < 0.1 * ly - 3.2, "no free space above the packing"
    cut_mm = 1e3 * (lx // 2) * (1e-4 * lx / (lx - 1))
    ya = 0.1 * ly - 2.0
    r = np.concatenate([r, [0.7, 0.7]]); y = np.concatenate([y, [ya, ya]])
    x = np.concatenate([x, [cut_mm - 0.003 - 1.35, cut_mm - 0.003]])
    return r, x, y, cut_mm


def test_c_driver_two_ranks_on_one_gpu_writes_the_single_gpu_files(po, tmp_path):
    """`lbmdem sample --gpus 2 --devices 0,0` (forked ranks, lbmdem_comm_run, the step-4000 table sub-step merged over the
    ranks, lbmdem_comm_write_vtk gathering on rank 0, the serial density chain through the ranks) leaves the same bytes on
    disk as `lbmdem sample`: five VTK frames at step 8000, DEM%06d.dat / .ps at 4000 and 8000, stats.data, and the
    final_density string. The run contains 81 Verlet rebuilds, two table sub-steps and a grain that changes owner."""
    lx, ly = 640, 160
    r, x, y, cut_mm = cut_sample(lx, ly, 230, seed=33)
    outs = {}
    for mode, extra in (("single", []), ("two", ["--gpus", "2", "--devices", "0,0"])):
        d = tmp_path / mode
        d.mkdir()
        sample = d / "packing.data"
        po.write_sample(str(sample), r, x, y)
        cmd = [EXE, str(sample), "--lx", str(lx), "--ly", str(ly), "--steps", "8001"] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=d, env=shim_env(), timeout=900)
        assert out.returncode == 0, (out.stdout[-400:], out.stderr[-1200:])
        outs[mode] = out
    a, b = tmp_path / "single", tmp_path / "two"
    names = sorted(p.name for p in a.iterdir())
    assert names == sorted(p.name for p in b.iterdir())
    assert sum(n.endswith(".vtk") for n in names) == 5 and "DEM000001.dat" in names and "DEM000000.ps" in names
    for n in names:
        assert (a / n).read_bytes() == (b / n).read_bytes(), n
    fd = lambda o: re.search(r"final_density: ([0-9.]+)", o.stderr).group(1)
    assert fd(outs["single"]) == fd(outs["two"])
    assert "(2 GPUs)" in outs["two"].stderr
    # the pushed grain started left of the cut and is right of it in the step-4000 table: it changed owner on the way
    table = np.loadtxt(b / "DEM000000.dat")
    assert x[-1] < cut_mm and 1e3 * table[-1, 2] > cut_mm + 0.01, (x[-1], 1e3 * table[-1, 2], cut_mm)


def test_c_driver_restart_of_two_ranks(po, tmp_path):
    """per-rank checkpoints of a two-rank run (carries agreed over the ranks: lbmdem_comm_sync_carries) and the restart"""
    lx, ly = 640, 160
    r, x, y, _ = cut_sample(lx, ly, 230, seed=34)
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), r, x, y)
    base = [EXE, str(sample), "--lx", str(lx), "--ly", str(ly), "--gpus", "2", "--devices", "0,0"]
    run = lambda extra: subprocess.run(base + extra, capture_output=True, text=True, cwd=tmp_path, env=shim_env(), timeout=600)
    full = run(["--steps", "230"])
    first = run(["--steps", "127", "--checkpoint", "half.ckpt"])
    second = run(["--steps", "230", "--restart", "half.ckpt"])
    for o in (full, first, second):
        assert o.returncode == 0, (o.stdout[-300:], o.stderr[-1200:])
    assert os.path.exists(tmp_path / "half.ckpt.rank0") and os.path.exists(tmp_path / "half.ckpt.rank1")
    fd = lambda o: re.search(r"final_density: ([0-9.]+)", o.stderr).group(1)
    assert fd(full) == fd(second)


def test_c_driver_stops_all_ranks_when_the_transport_does_not_come_up(po, tmp_path):
    """the start-up watchdog of `lbmdem --gpus N`: ranks whose communicators + first neighbour exchange are not up within
    --comm-timeout seconds are killed with one message instead of hanging the job (here: a limit no start-up can meet)"""
    lx, ly = 640, 160
    r, x, y, _ = cut_sample(lx, ly, 230, seed=35)
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), r, x, y)
    cmd = [EXE, str(sample), "--lx", str(lx), "--ly", str(ly), "--gpus", "2", "--devices", "0,0", "--steps", "40"]
    t0 = time.time()
    out = subprocess.run(cmd + ["--comm-timeout", "0.01"], capture_output=True, text=True, cwd=tmp_path, env=shim_env(), timeout=300)
    assert out.returncode != 0 and "did not come up within" in out.stderr, (out.returncode, out.stderr[-600:])
    assert time.time() - t0 < 60
    # ... and with the default limit the same command runs
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, env=shim_env(), timeout=600)
    assert out.returncode == 0 and "final_density" in out.stderr, (out.stdout[-300:], out.stderr[-800:])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_python_ccomm_runner_on_one_gpu_equals_the_oracle(world):
    """strips.CCommRunner (what bench.py --gpus N measures) with `world` ranks on one GPU against the CPU oracle, across a
    Verlet rebuild; control plane over gloo. 8 = the rank count of the node the scaling curve is taken on."""
    env = shim_env(SHARED_GPU="1", MODE="ccomm", NSTEPS="113" if world < 8 else "41", RCCL_SHIM_TIMEOUT_S="90")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29670 + world), os.path.join(ROOT, "tests", "multi_gpu_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "MULTI-GPU-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_self_launch_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver starts it (no launcher): it re-executes itself under torch.distributed.run,
    runs the watchdogged C-driver trial in child processes, and prints ONE line with n_gpus 2 from the C driver. (Both
    ranks share the GPU here, so the number says nothing; the path is what is tested.)"""
    import json
    env = shim_env(LBMDEM_BENCH_DEVICES="0,0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-600:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-600:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 100
    assert d["config"]["driver"].startswith("C (lbmdem_comm_run") and "passed" in d["config"]["driver_trial"]


def test_bench_two_gpus_on_a_one_gpu_box_fails_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("LBMDEM_BENCH_DEVICES",)}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 2 and "needs 2 GPUs, this node has 1" in out.stderr, (out.returncode, out.stderr[-800:])
    assert "Traceback" not in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
