#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container, where /root/reference exists: oracle/ref_harness.c compiles the
reference's src/main.c (one library per lattice size, -O2 -ffp-contract=off) and this script drives
it. What is committed is data only -- the inputs we authored (grain radii/positions in the
reference's .data units, initial perturbation seeds) and the reference's outputs -- never the
reference's source.

    python tests/golden/make_golden.py            # regenerate every fixture

Each case runs in its own process (the reference keeps its state in file-scope globals).
Large arrays are stored as a strided sample plus the SHA-256 of the full little-endian float64
buffer, which pins every bit while keeping the repository small.
"""
import hashlib
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def perturbation(lx, ly, seed, amp):
    """Smooth + random multiplicative perturbation of f (applied to whatever f currently is)."""
    rng = np.random.default_rng(seed)
    x = np.arange(lx)[:, None, None] / lx
    y = np.arange(ly)[None, :, None] / ly
    smooth = 0.02 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.linspace(-1, 1, 9)[None, None, :]
    return 1.0 + smooth + amp * rng.standard_normal((lx, ly, 9))


# ---- case definitions (shared with the tests through cases()) ---------------------------------------

def cases():
    import samples
    c = {}
    # G1: (almost) fluid-only -- the reference cannot run with 0 grains (main.c:220), so one small
    # grain sits in a corner; f starts from a smooth + noisy perturbation. Fluid phases only.
    for name, lx, ly in (("G1_fluid_64x48", 64, 48), ("G1_fluid_128x128", 128, 128)):
        c[name] = dict(kind="lbm", lx=lx, ly=ly, r_mm=[0.5], x_mm=[1.2], y_mm=[1.1], kin=None,
                       pert=(101, 1e-3), dumps=(1, 10, 50))
    # G2: one translating + spinning grain -> reinit, both delta branches, moving-wall term, fhf
    c["G2_moving_grain_96x96"] = dict(kind="lbm", lx=96, ly=96, r_mm=[0.85], x_mm=[4.63], y_mm=[4.21],
                                      kin=[[0.03, -0.02, 40.0]], pert=(202, 1e-3), dumps=(1, 5, 20),
                                      drift=[[2.3e-5, -1.7e-5]])
    # G3: two grains one fluid node apart -> the in-place IBB order hazard (SURVEY.md hard part 2)
    dx = 0.1 * 96 / 95
    d = 0.85 * (0.7 + 0.6) + 1.6 * dx
    ang = 0.25 * np.pi
    c["G3_hazard_96x96"] = dict(kind="lbm", lx=96, ly=96, r_mm=[0.7, 0.6],
                                x_mm=[4.0, 4.0 + d * np.cos(ang)], y_mm=[4.2, 4.2 + d * np.sin(ang)],
                                kin=[[0.01, -0.015, 15.0], [-0.02, 0.01, -25.0]], pert=(303, 1e-2),
                                dumps=(1, 2, 3))
    # G4: ~600-grain packing, fully coupled, 20 fluid steps
    r, x, y = samples.row_packing(256, 200, 600, seed=77)
    c["G4_coupled_256x200"] = dict(kind="coupled", lx=256, ly=200, r_mm=r, x_mm=x, y_mm=y, fluid_steps=20)
    # G6: output files. G4's packing plus four grains pressed into the DEM walls, 4000 renderScene calls:
    # the reference then writes DEM000000.dat and appends a line to stats.data (main.c:1773-1776)
    r, x, y = samples.row_packing(256, 200, 600, seed=77)
    r = np.concatenate([r, [0.6, 0.7, 0.8, 0.9]])
    x = np.concatenate([x, [0.6 - 0.002, 60.0, 256 - 0.8 + 0.002, 80.0]])
    y = np.concatenate([y, [90.0, 0.7 - 0.003, 40.0, 200 - 0.9 + 0.003]])
    c["G6_output_256x200"] = dict(kind="output", lx=256, ly=200, r_mm=r, x_mm=x, y_mm=y, steps=4000)
    # G5: DEM-focused: 240 renderScene calls on a tiny lattice with random initial grain velocities:
    # film law at step 0, Verlet rebuilds at 0/100/200, grain-grain contacts, and four extra grains
    # pressed 2-3 um into the left/bottom/right/top DEM walls (the right/top walls sit at 1e-3*lx,
    # 1e-3*ly metres after the first VerletWall, main.c:1559-1560 -- outside the lattice).
    r, x, y = samples.row_packing(64, 48, 14, seed=9)
    r = np.concatenate([r, [0.6, 0.7, 0.8, 0.9]])
    x = np.concatenate([x, [0.6 - 0.002, 20.0, 64 - 0.8 + 0.002, 30.0]])
    y = np.concatenate([y, [30.0, 0.7 - 0.003, 10.0, 48 - 0.9 + 0.003]])
    c["G5_dem_64x48"] = dict(kind="dem", lx=64, ly=48, r_mm=r, x_mm=x, y_mm=y, seed=505, dumps=(1, 120, 240))
    return c


def kin_table(case):
    r_mm = np.asarray(case["r_mm"], float)
    n = len(r_mm)
    k = np.zeros((n, 9))
    k[:, 0] = np.asarray(case["x_mm"], float) * 1e-3
    k[:, 1] = np.asarray(case["y_mm"], float) * 1e-3
    if case.get("kin") is not None:
        k[:, 3:6] = np.asarray(case["kin"], float)
    return k


def dem_initial_kinematics(case):
    k = kin_table(case)
    rng = np.random.default_rng(case["seed"])
    k[:, 3:6] = rng.normal(0, 1, (len(k), 3)) * [0.05, 0.05, 30.0]
    return k


# ---- drivers (work on anything with the Reference/Oracle interface) ---------------------------------

def run_lbm_case(sim, case):
    """-> dict of dumps. `sim` has set_f/get_f/set_kinematics/lbm_steps/get_obst/get_fhf."""
    lx, ly = case["lx"], case["ly"]
    k = kin_table(case)
    sim.set_kinematics(k)
    f0 = sim.get_f() * perturbation(lx, ly, *case["pert"])
    sim.set_f(f0)
    out = {}
    done = 0
    for target in case["dumps"]:
        while done < target:
            if case.get("drift") is not None:
                k[:, 0:2] += np.asarray(case["drift"], float)
                sim.set_kinematics(k)
            sim.lbm_steps(1)
            done += 1
        out[f"f_{target}"] = sim.get_f()
        out[f"obst_{target}"] = sim.get_obst()
        out[f"fhf_{target}"] = sim.get_fhf()
    return out


def run_coupled_case(sim, case):
    npdem = sim.scalars()["npDEM"]
    sim.steps(case["fluid_steps"] * npdem)
    return {"f": sim.get_f(), "obst": sim.get_obst(), "fhf": sim.get_fhf(), "grains": sim.get_grains()}


def run_dem_case(sim, case):
    sim.set_kinematics(dem_initial_kinematics(case))
    out = {}
    done = 0
    for target in case["dumps"]:
        sim.steps(target - done)
        done = target
        out[f"grains_{target}"] = sim.get_grains()
    out["f_final"] = sim.get_f()
    return out


def _generate(name, case, q, sp=False):
    import pyoracle as po
    lx, ly = case["lx"], case["ly"]
    tmp = tempfile.NamedTemporaryFile("w", suffix=".data", delete=False)
    tmp.close()
    po.write_sample(tmp.name, case["r_mm"], case["x_mm"], case["y_mm"], comment=f"#golden {name}")
    R = po.Reference(lx, ly, tmp.name, sp=sp)      # sp: the reference compiled -DSINGLE_PRECISION (main.c:34-40)
    os.unlink(tmp.name)
    if case["kind"] == "lbm":
        res = run_lbm_case(R, case)
    elif case["kind"] == "coupled":
        res = run_coupled_case(R, case)
    else:
        res = run_dem_case(R, case)
    res["scalars"] = np.array([R.scalars()[k] for k in po.SCALARS], float)
    q.put(res)


def pack(name, case, res):
    """Reduce a result dict to what is stored."""
    out = {"r_mm": np.asarray(case["r_mm"], float), "x_mm": np.asarray(case["x_mm"], float),
           "y_mm": np.asarray(case["y_mm"], float), "scalars": res["scalars"]}
    for k, v in res.items():
        if k == "scalars":
            continue
        if isinstance(v, str):
            out[k] = np.array(v)
            continue
        v = np.asarray(v)
        out[k + "_sha"] = np.array(sha(v))
        if v.ndim == 3 and v.shape[0] * v.shape[1] > 64 * 48:      # big f dump: strided sample
            out[k + "_sample"] = v[::4, ::4, :].copy()
        elif v.ndim == 2 and v.dtype.kind == "i" and v.size > 64 * 48:
            out[k + "_sample"] = v[::4, ::4].copy()
        else:
            out[k] = v
    return out


def make_vtk_fixture():
    """tests/golden/vtk_G5_25steps/*.vtk: the five files the reference's write_vtk (main.c:237-338)
    produces for case G5 after 25 renderScene calls with nFile = 3 (data written by the reference)."""
    import pyoracle as po
    c = cases()["G5_dem_64x48"]
    tmp = tempfile.NamedTemporaryFile("w", suffix=".data", delete=False)
    tmp.close()
    po.write_sample(tmp.name, c["r_mm"], c["x_mm"], c["y_mm"])
    R = po.Reference(64, 48, tmp.name)
    os.unlink(tmp.name)
    R.set_kinematics(dem_initial_kinematics(c))
    R.steps(25)
    out = os.path.join(HERE, "vtk_G5_25steps")
    os.makedirs(out, exist_ok=True)
    assert R.L.ref_write_vtk(os.fsencode(out), 3) == 0


def make_dry_output_fixture():
    """tests/golden/dem_dry_G6_4000steps/: the same files from the reference compiled WITHOUT `_FLUIDE_` (main.c:16 taken out:
    oracle/Makefile ref_dry) -- its DEM-only mode: no fluid step, no VTK frames, hydrodynamic forces 0 (main.c:1709-1719,
    1768-1772)."""
    make_dem_output_fixture(dry=True)


def make_dem_output_fixture(dry=False):
    """tests/golden/dem_G6_4000steps/{DEM000000.dat, stats.data}: what the reference's write_DEM
    (main.c:340-438) writes for case G6 at step 4000, plus the grain table at that moment."""
    import ctypes
    import pyoracle as po
    c = cases()["G6_output_256x200"]
    tmp = tempfile.NamedTemporaryFile("w", suffix=".data", delete=False)
    tmp.close()
    po.write_sample(tmp.name, c["r_mm"], c["x_mm"], c["y_mm"])
    R = po.Reference(c["lx"], c["ly"], tmp.name, dry=dry)
    os.unlink(tmp.name)
    out = os.path.join(HERE, "dem_dry_G6_4000steps" if dry else "dem_G6_4000steps")
    os.makedirs(out, exist_ok=True)
    for f in os.listdir(out):
        os.unlink(os.path.join(out, f))
    assert R.L.ref_steps_in_dir(ctypes.c_long(c["steps"]), os.fsencode(out)) == 0
    # write_forces' PostScript picture (main.c:440-478). Three header lines come from undefined printf
    # conversions ("%%%BoundingBox" ...) and the (nbgrains+1)-th grain line from g[nbgrains], one element
    # past the array: dropped here; every other line is kept as the fixture.
    ps = os.path.join(out, "DEM000000.ps")
    lines = open(ps, "rb").read().split(b"\n")
    n = len(c["r_mm"])
    assert lines[0].startswith(b"%!PS-Adobe") and lines[4].startswith(b"0.1 setlinewidth")
    assert all(l.startswith(b"newpath ") for l in lines[5:5 + n + 1])
    kept = [lines[0], lines[4]] + lines[5:5 + n] + lines[5 + n + 1:]
    open(ps, "wb").write(b"\n".join(kept))
    np.savez_compressed(os.path.join(out, "inputs_and_table.npz"), r_mm=np.asarray(c["r_mm"], float),
                        x_mm=np.asarray(c["x_mm"], float), y_mm=np.asarray(c["y_mm"], float),
                        grains=R.get_grains())


# ---- the reference's own sample inputs at BASELINE.json's sizes ----------------------------------------

REAL_CASES = {
    # name: (sample file under /root/reference/bin, lx, ly, fluid steps at which the state is hashed)
    "real_a08d83_600x500": ("a08d83.data", 600, 500, (1, 10, 20)),             # SURVEY.md section 8c known answers
    "real_7000_2048x2048": ("a08_a4b4r18_7000.data", 2048, 2048, (1, 2, 50)),  # BASELINE.json configs[2]
    "real_50000_4096x4096": ("50000.data", 4096, 4096, (1, 2, 10, 20, 100)),       # BASELINE.json configs[3]
    "real_50000test_3072x3072": ("50000-test.data", 3072, 3072, (1, 2, 30)),   # BASELINE.json configs[0] (47 980 grains)
    "real_50000_8192x4096": ("50000.data", 8192, 4096, (1, 2, 10, 20)),        # BASELINE.json configs[4] (one domain = 8 strips)
    # the JUBE cases of the reference's own benchmark.xml (lines 7, 13-27) at scale 1, and the remaining shipped samples
    "real_a08d83_2000x1000": ("a08d83.data", 2000, 1000, (1, 2, 10)),
    "real_a6d83_2000x1000": ("a6d83.data", 2000, 1000, (1, 2, 10)),            # grains up to y = 1493 nodes: outside the lattice
    "real_slope_12500x2500": ("slope.data", 12500, 2500, (1, 2, 10)),          # r up to 1.5 mm (12.8 nodes reduced)
    "real_spl04_1600x2400": ("spl04.data", 1600, 2400, (1, 2, 10)),            # = 10000_before_packing.data, 9 984 grains
}
# fixtures that do not repeat the parsed grains of another fixture of the same sample file (r, x1, x2 are the same
# doubles: read_sample does not depend on the lattice)
GRAINS_FROM = {"real_50000_8192x4096": "real_50000_4096x4096"}


F32_REAL_CASES = {"real_a08d83_600x500_f32": ("a08d83.data", 600, 500, (1, 10, 20))}


def _real_case(name, q, sp=False):
    """Runs in its own process: the reference on one of its shipped samples; returns the parsed grains
    (exactly the doubles the reference works with) and SHA-256 digests of its state."""
    import pyoracle as po
    fname, lx, ly, dumps = (F32_REAL_CASES if sp else REAL_CASES)[name]
    po.build_ref(lx, ly, sp=sp)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    R = po.Reference(lx, ly, os.path.join(po.REF_ROOT, "bin", fname), sp=sp)
    g0 = R.get_grains()
    s = R.scalars()
    out = dict(r=g0[:, po.COL["r"]].copy(), x1=g0[:, 0].copy(), x2=g0[:, 1].copy(),
               npDEM=np.int64(s["npDEM"]), dx=np.float64(s["dx"]), c=np.float64(s["c"]), dumps=np.array(dumps))
    if name in GRAINS_FROM:
        other = np.load(os.path.join(HERE, GRAINS_FROM[name] + ".npz"))
        assert all(np.array_equal(out[k], other[k]) for k in ("r", "x1", "x2"))
        for k in ("r", "x1", "x2"):
            del out[k]
        out["grains_from"] = np.array(GRAINS_FROM[name])
    done = 0
    for k in dumps:
        R.steps((k - done) * int(s["npDEM"]))
        done = k
        g = R.get_grains()
        out[f"sha_f_{k}"] = sha(R.get_f())
        out[f"sha_obst_{k}"] = sha(R.get_obst().astype(np.int32))
        out[f"sha_fhf_{k}"] = sha(R.get_fhf())
        out[f"sha_kin_{k}"] = sha(g[:, :9])
        out[f"mass_{k}"] = np.float64(R.total_density())
        out[f"grain0_{k}"] = g[0, :9].copy()
    q.put(out)


def make_real_fixtures(only=None):
    for name in (only or REAL_CASES):
        q = mp.Queue()
        p = mp.Process(target=_real_case, args=(name, q))
        p.start()
        res = q.get()
        p.join()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
        print("wrote", name, {k: str(v)[:20] for k, v in res.items() if k.startswith(("sha_f", "mass"))})


def make_f32_fixtures():
    """<case>_f32.npz: the same cases on the reference compiled -DSINGLE_PRECISION (typedef float real, main.c:34-40):
    the oracle of the float build of the library (liblbmdem_hip_sp.so). Stored as float64 arrays holding float values."""
    for name, case in cases().items():
        if case["kind"] == "output":
            continue
        q = mp.Queue()
        p = mp.Process(target=_generate, args=(name, case, q, True))
        p.start()
        res = q.get()
        p.join()
        np.savez_compressed(os.path.join(HERE, name + "_f32.npz"), **pack(name, case, res))
        print("wrote", name + "_f32")
    for name in F32_REAL_CASES:
        q = mp.Queue()
        p = mp.Process(target=_real_case, args=(name, q, True))
        p.start()
        res = q.get()
        p.join()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
        print("wrote", name, {k: str(v)[:20] for k, v in res.items() if k.startswith(("sha_f", "mass"))})


def main():
    import pyoracle as po
    if not po.reference_available():
        raise SystemExit("the reference is not present here; golden vectors can only be made in the build container")
    if len(sys.argv) > 1 and sys.argv[1] == "--f32":
        make_f32_fixtures()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--dry":
        make_dry_output_fixture()
        return
    if len(sys.argv) > 1:      # python make_golden.py real_50000_8192x4096 ...: only these real-sample fixtures
        make_real_fixtures(sys.argv[1:])
        return
    for name, case in cases().items():
        if case["kind"] == "output":
            continue
        q = mp.Queue()
        p = mp.Process(target=_generate, args=(name, case, q))
        p.start()
        res = q.get()
        p.join()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(name, case, res))
        print("wrote", name, {k: getattr(v, "shape", None) for k, v in pack(name, case, res).items()})
    make_real_fixtures()
    for target in (make_vtk_fixture, make_dem_output_fixture, make_dry_output_fixture):
        p = mp.Process(target=target)
        p.start()
        p.join()


if __name__ == "__main__":
    main()
