"""Shared driver for the golden-vector tests: run a case definition (tests/golden/make_golden.py)
on any backend that offers the Reference/Oracle method names and compare with the stored dumps."""
import hashlib
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)

ALL_CASES = mg.cases()
CASES = {k: v for k, v in ALL_CASES.items() if v["kind"] != "output"}   # "output" cases have file fixtures


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    g = dict(np.load(os.path.join(HERE, "golden", name + ".npz")))
    if "grains_from" in g:   # same sample file as another fixture: the parsed grains are stored once
        other = np.load(os.path.join(HERE, "golden", str(g["grains_from"]) + ".npz"))
        for k in ("r", "x1", "x2"):
            g[k] = other[k]
    return g


def inputs_m(name):
    g = load(name)
    return g["r_mm"] * 1e-3, g["x_mm"] * 1e-3, g["y_mm"] * 1e-3


def run_case(sim, name):
    case = CASES[name]
    if case["kind"] == "lbm":
        return mg.run_lbm_case(sim, case)
    if case["kind"] == "coupled":
        return mg.run_coupled_case(sim, case)
    return mg.run_dem_case(sim, case)


def compare(name, res, grain_cols=None):
    """Exact comparison with the fixture. grain_cols: restrict grain tables to these columns (the HIP
    path carries the 9 kinematic columns only)."""
    g = load(name)
    checked = 0
    for key, val in res.items():
        val = np.asarray(val)
        if key.startswith("grains") and grain_cols is not None:
            ref = g[key][:, grain_cols]
            assert np.array_equal(val, ref), f"{name}/{key}: max abs {np.abs(val - ref).max():.3e}"
            checked += 1
            continue
        if key in g:
            assert np.array_equal(val, g[key], equal_nan=True), f"{name}/{key} differs from the reference dump"
        else:
            samp = val[::4, ::4] if val.ndim == 2 else val[::4, ::4, :]
            assert np.array_equal(samp, g[key + "_sample"]), f"{name}/{key} (strided sample) differs"
        ref_sha = str(g[key + "_sha"])
        if val.dtype == np.float64 or val.dtype == np.int32:
            assert sha(val) == ref_sha, f"{name}/{key}: SHA-256 of the full buffer differs"
        checked += 1
    assert checked > 0


class GpuAdapter:
    """Gives the HIP path (LbmDem) the method names the case drivers use."""

    def __init__(self, pkg, name):
        c = CASES[name]
        r, x1, x2 = inputs_m(name)
        self.sim = pkg.LbmDem(c["lx"], c["ly"], r, x1, x2)

    def set_kinematics(self, k): self.sim.kinematics = k
    def set_f(self, f): self.sim.f = f
    def get_f(self): return self.sim.f
    def get_obst(self): return self.sim.obst
    def get_fhf(self): return self.sim.fhf
    def get_grains(self): return self.sim.kinematics
    def scalars(self): return {"npDEM": self.sim.cfg.npDEM}
    def steps(self, n): self.sim.renderScene(n)

    def lbm_steps(self, n):
        for _ in range(n):
            self.sim.lbm_step()


# ---- the reference's own shipped samples (tests/golden/real_*.npz: parsed grains + SHA-256 of its state) ----
REAL_CASES = mg.REAL_CASES


def check_real_case(name, make_sim, state_of, dumps=None):
    """Run `name` on a backend and compare the SHA-256 of f, obst, fhf and the kinematics with what the
    unmodified reference produced from its own sample file. make_sim(lx, ly, r, x1, x2) -> sim with
    .steps(n); state_of(sim) -> (f[lx][ly][9], obst int32, fhf[n][3], kin[n][9], total mass)."""
    g = load(name)
    _, lx, ly, all_dumps = REAL_CASES[name]
    sim = make_sim(lx, ly, g["r"], g["x1"], g["x2"])
    npdem = int(g["npDEM"])
    done = 0
    for k in (dumps or all_dumps):
        sim.steps((k - done) * npdem)
        done = k
        f, obst, fhf, kin, mass = state_of(sim)
        assert np.array_equal(kin[0], g[f"grain0_{k}"]), (name, k, "grain 0")
        assert sha(kin) == str(g[f"sha_kin_{k}"]), (name, k, "kinematics")
        assert sha(fhf) == str(g[f"sha_fhf_{k}"]), (name, k, "hydrodynamic forces")
        assert sha(obst.astype(np.int32)) == str(g[f"sha_obst_{k}"]), (name, k, "obstacle map")
        assert sha(f) == str(g[f"sha_f_{k}"]), (name, k, "populations")
        if mass is not None:
            assert mass == float(g[f"mass_{k}"]), (name, k, "total density")
    return sim
