"""GPU: the kept C host side (2d-lbm-dem_amd/host/lbmdem) as a drop-in for the reference binary:
`lbmdem <sample.data>`, the reference's console lines, `final_density:` on stderr (what the
reference's JUBE benchmark parses, benchmark.xml:99-102), VTK frames at the stepFilm cadence."""
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "2d-lbm-dem_amd", "host", "lbmdem")


def test_usage_error_like_the_reference():
    out = subprocess.run([EXE], capture_output=True, text=True)
    assert out.returncode != 0 and "usage" in out.stdout


def test_run_matches_oracle_and_writes_a_frame(po, tmp_path):
    c = gu.CASES["G4_coupled_256x200"]
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"], comment="#Length =25 Height =20")
    nsteps = 8001      # one VTK frame at nbsteps == 8000 (main.c:1767)
    out = subprocess.run([EXE, str(sample), "--lx", "256", "--ly", "200", "--steps", str(nsteps)],
                         capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert out.returncode == 0, out.stderr[-500:]
    assert "Nb grains %d" % len(c["r_mm"]) in out.stdout
    assert re.search(r"dtLB=.*npDEM=\d+", out.stdout)
    assert "Iteration Number 0, Total density in the system" in out.stdout
    fd_text = re.search(r"final_density: ([0-9.]+)", out.stderr).group(1)
    r, x1, x2 = po.read_sample(str(sample))
    ora = po.Oracle(256, 200, r, x1, x2)
    ora.steps(nsteps)
    assert fd_text == "%f" % ora.total_density()     # the string the reference's benchmark parses (benchmark.xml:99-102)
    frames = sorted(p.name for p in tmp_path.glob("*.vtk"))
    assert frames == ["fluid_pressure_000000.vtk", "fluid_velocity_000000.vtk", "grain_acceleration_000000.vtk",
                      "grain_pressure_000000.vtk", "grain_velocity_000000.vtk"]
    assert os.path.getsize(tmp_path / "fluid_velocity_000000.vtk") > 256 * 200 * 12
    # write_DEM at steps 4000 and 8000 (main.c:1773): nFile is 0 at 4000 and 1 at 8000 (incremented by the
    # VTK frame written just before)
    assert (tmp_path / "DEM000000.dat").exists() and (tmp_path / "DEM000001.dat").exists()
    assert (tmp_path / "DEM000000.ps").exists() and (tmp_path / "DEM000001.ps").exists()   # write_forces
    stats = open(tmp_path / "stats.data").read().splitlines()
    assert stats[0].startswith("#1_t 2_xfront") and len(stats) == 3 and len(stats[1].split()) == 22
    assert re.search(r"steps 8000 steps .* KE [0-9.e+-]+ PE [0-9.e+-]+ SE", out.stdout)


@pytest.mark.parametrize("comm", [[], ["--comm"]])
def test_checkpoint_and_restart_from_the_command_line(po, tmp_path, comm):
    """--checkpoint / --restart; with --comm (the code path of --gpus N) every rank keeps its own file, FILE.rank<k>"""
    c = gu.CASES["G4_coupled_256x200"]
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"])
    base = [EXE, str(sample), "--lx", "256", "--ly", "200"] + comm
    full = subprocess.run(base + ["--steps", "150"], capture_output=True, text=True, cwd=tmp_path, timeout=900)
    a = subprocess.run(base + ["--steps", "67", "--checkpoint", "half.ckpt"], capture_output=True, text=True,
                       cwd=tmp_path, timeout=900)
    b = subprocess.run(base + ["--steps", "150", "--restart", "half.ckpt"], capture_output=True, text=True,
                       cwd=tmp_path, timeout=900)
    assert full.returncode == 0 and a.returncode == 0 and b.returncode == 0, (a.stderr[-300:], b.stderr[-300:])
    assert ("Restarted from half.ckpt.rank0 at step 67" if comm else "Restarted from half.ckpt at step 67") in b.stdout
    assert os.path.exists(tmp_path / ("half.ckpt.rank0" if comm else "half.ckpt"))
    fd = lambda out: re.search(r"final_density: ([0-9.]+)", out.stderr).group(1)
    assert fd(full) == fd(b)


def test_duration_stops_after_the_same_step_as_the_reference(po, tmp_path):
    """`do renderScene() while (nbsteps * dt <= duration)` (main.c:1880-1890): the stop condition is tested after
    every DEM step, so a run ends at the first step count with nbsteps * dt > duration -- here a count that is not
    a multiple of the 100-step console cadence -- and final_density is the oracle's after exactly that many steps."""
    c = gu.CASES["G4_coupled_256x200"]
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"])
    r, x1, x2 = po.read_sample(str(sample))
    ora = po.Oracle(256, 200, r, x1, x2)
    dt = ora.scalars()["dt"]
    want_steps = 237
    duration = (want_steps - 0.5) * dt          # 236 * dt <= duration < 237 * dt
    out = subprocess.run([EXE, str(sample), "--lx", "256", "--ly", "200", "--duration", repr(float(duration))],
                         capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert out.returncode == 0, out.stderr[-500:]
    assert int(re.search(r"dem_steps: (\d+)", out.stderr).group(1)) == want_steps
    assert re.search(r"steps 200 steps", out.stdout) and not re.search(r"steps 300 steps", out.stdout)
    fd_text = re.search(r"final_density: ([0-9.]+)", out.stderr).group(1)
    ora.steps(want_steps)
    assert fd_text == "%f" % ora.total_density()     # the string the reference's benchmark parses (benchmark.xml:99-102)


def test_rccl_path_of_the_c_driver_with_one_rank(po, tmp_path):
    """`lbmdem --comm`: the multi-GPU code path of the C driver (grains distributed, lbmdem_comm_* over RCCL: communicator
    from a unique id, a grouped self send/recv on a side stream, lbmdem_comm_run, the density all-reduce) with a single
    rank -- all a one-GPU box can run; `--gpus N` forks one such process per GPU."""
    c = gu.CASES["G4_coupled_256x200"]
    sample = tmp_path / "packing.data"
    po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"])
    nsteps = 130
    out = subprocess.run([EXE, str(sample), "--lx", "256", "--ly", "200", "--steps", str(nsteps), "--comm"],
                         capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert out.returncode == 0, (out.stdout[-300:], out.stderr[-600:])
    assert int(re.search(r"dem_steps: (\d+)", out.stderr).group(1)) == nsteps
    fd_text = re.search(r"final_density: ([0-9.]+)", out.stderr).group(1)
    r, x1, x2 = po.read_sample(str(sample))
    ora = po.Oracle(256, 200, r, x1, x2)
    ora.steps(nsteps)
    assert fd_text == "%f" % ora.total_density()     # the string the reference's benchmark parses (benchmark.xml:99-102)


def test_transport_selftest_and_exchange_probe_with_one_rank():
    """The library's RCCL transport from Python (one-rank communicator, in a fresh process: RCCL initialises once per
    process): the self-test, and -- with the experiment build, which alone carries the helper -- the probe that prices
    one critical-path exchange: through a side stream with event hand-overs, and in line on the main stream, which is
    how lbmdem_comm_lbm_step sends the link-sum tables and the forces."""
    import subprocess, sys
    code = r'''
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "scripts")]
import __graft_entry__ as ge
pkg = ge.load_package()
c = pkg.Comm(pkg.comm_unique_id(), 0, 1, 0)
c.selftest()
assert not hasattr(pkg.load_library(), "lbmdem_comm_exchange_probe") or "_ab" in pkg.LIB_PATH
if "_ab" in pkg.LIB_PATH:
    from exchange_probe import exchange_probe
    side, without, inline = exchange_probe(pkg, c, 20000, 100)
    assert 0 < without < inline < side, (side, without, inline)
    print("PROBE %.1f %.1f %.1f" % (side, without, inline))
print("TRANSPORT-OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, "2d-lbm-dem_amd", "liblbmdem_hip_ab.so")
    for lib in [None] + ([ab] if os.path.exists(ab) else []):
        env = dict(os.environ)
        env.pop("LBMDEM_HIP_LIBRARY", None)
        if lib:
            env["LBMDEM_HIP_LIBRARY"] = lib
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "TRANSPORT-OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
        assert (lib is None) or "PROBE" in out.stdout


def test_comm_driver_writes_the_same_files_as_the_single_gpu_driver(po, tmp_path):
    """`lbmdem --comm` (the code path of `--gpus N`: grains distributed, lbmdem_comm_run, outputs merged over the ranks:
    lbmdem_comm_write_vtk, the table sub-step on rank 0's full replica) leaves the same frames and tables on disk as the
    single-GPU driver: five VTK files at DEM step 8000, DEM000000.dat/.ps at 4000, DEM000001.dat/.ps at 8000, stats.data."""
    c = gu.CASES["G4_coupled_256x200"]
    outs = {}
    for mode in ("single", "comm"):
        d = tmp_path / mode
        d.mkdir()
        sample = d / "packing.data"
        po.write_sample(str(sample), c["r_mm"], c["x_mm"], c["y_mm"])
        cmd = [EXE, str(sample), "--lx", "256", "--ly", "200", "--steps", "8001"] + (["--comm"] if mode == "comm" else [])
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=d, timeout=600)
        assert out.returncode == 0, (out.stdout[-300:], out.stderr[-600:])
        outs[mode] = out
    a, b = tmp_path / "single", tmp_path / "comm"
    names = sorted(p.name for p in a.iterdir())
    assert names == sorted(p.name for p in b.iterdir())
    assert sum(n.endswith(".vtk") for n in names) == 5 and "DEM000001.dat" in names and "DEM000000.ps" in names
    for n in names:
        assert (a / n).read_bytes() == (b / n).read_bytes(), n
    fd = lambda o: re.search(r"final_density: ([0-9.]+)", o.stderr).group(1)
    assert fd(outs["single"]) == fd(outs["comm"])
