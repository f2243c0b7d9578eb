"""ctypes bindings for the CHECKERS: the in-repo CPU oracle (oracle/lbmdem_oracle.c) and, where it
has been built in-container, the unmodified reference TU (oracle/ref_harness.c -> oracle/_ref/).

TEST INFRASTRUCTURE ONLY. Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (2d-lbm-dem_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
GRAIN_COLS = 30
# column indices in the grain table (reference struct order, main.c:182-197)
COL = {n: i for i, n in enumerate(
    "x1 x2 x3 v1 v2 v3 a1 a2 a3 r m mw It p s f1 f2 ifm fm fr ifr M11 M12 M21 M22 ice slip rw z zz".split())}
SCALARS = "dx dtLB dt dt2 c npDEM Mgx Mdx Mby Mhy xG yG".split()


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle(fast: bool = False) -> str:
    """Compile the CPU restatement (pinned flags, or Release-style flags for timing)."""
    target = "oracle_fast" if fast else "oracle"
    name = "liblbmdem_oracle_fast.so" if fast else "liblbmdem_oracle.so"
    path = os.path.join(HERE, "_build", name)
    src = os.path.join(HERE, "lbmdem_oracle.c")
    if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", HERE, target], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return path


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "src", "main.c"))


def ref_lib_path(lx: int, ly: int, fast: bool = False, sp: bool = False, dry: bool = False) -> str:
    return os.path.join(HERE, "_ref", f"libref_{'fast_' if fast else ''}{'sp_' if sp else ''}{'dry_' if dry else ''}{lx}x{ly}.so")


def build_ref(lx: int, ly: int, fast: bool = False, sp: bool = False, dry: bool = False) -> str | None:
    """Compile the reference TU for one lattice size (only where /root/reference exists). sp: -DSINGLE_PRECISION;
    dry: with its `#define _FLUIDE_` (main.c:16) taken out -- the reference's DEM-only mode."""
    path = ref_lib_path(lx, ly, fast, sp, dry)
    if os.path.exists(path):
        return path
    if not reference_available():
        return None
    subprocess.run(["make", "-s", "-C", HERE, "ref_dry" if dry else ("ref_sp" if sp else ("ref_fast" if fast else "ref")), f"LX={lx}", f"LY={ly}"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return path if os.path.exists(path) else None


class Oracle:
    """The in-repo CPU restatement."""

    def __init__(self, lx, ly, r, x1, x2, scale=1.0, fast=False):
        self.L = C.CDLL(build_oracle(fast))
        L = self.L
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for fn in ("ora_f", "ora_delta", "ora_obst", "ora_act"):
            getattr(L, fn).restype = C.c_void_p
            getattr(L, fn).argtypes = [C.c_void_p]
        L.ora_total_density.restype = C.c_double
        L.ora_total_density.argtypes = [C.c_void_p]
        L.ora_nbsteps.restype = C.c_long
        L.ora_link_sums.restype = C.c_int
        L.ora_link_sums.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ora_force_from_link_sums.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ora_count_act_anomalies.restype = C.c_long
        r = np.ascontiguousarray(r, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        self.lx, self.ly, self.n = int(lx), int(ly), len(r)
        self.h = C.c_void_p(L.ora_create(lx, ly, float(scale), self.n, _vp(r), _vp(x1), _vp(x2)))
        if not self.h:
            raise RuntimeError("ora_create failed")

    @classmethod
    def from_file(cls, path, lx, ly, scale=1.0, fast=False):
        r, x1, x2 = read_sample(path)
        return cls(lx, ly, r, x1, x2, scale, fast)

    def close(self):
        if self.h:
            self.L.ora_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        fn = getattr(self.L, name)
        return fn(self.h, *args)

    # stepping
    def steps(self, n): self._call("ora_steps", C.c_long(n))
    def steps_dry(self, n): self._call("ora_steps_dry", C.c_long(n))   # the reference without _FLUIDE_ (main.c:16)
    def lbm_steps(self, n): self._call("ora_lbm_steps", C.c_int(n))
    def reinit(self): self._call("ora_reinit_obst_density")
    def obst_construction(self): self._call("ora_obst_construction")
    def collision_streaming(self): self._call("ora_collision_streaming")
    def collide(self): self._call("ora_collide")
    def edges(self): self._call("ora_edges")
    def grain_ibb(self): self._call("ora_grain_ibb")
    def swap_stream(self): self._call("ora_swap_stream")
    def forces_fluid(self): self._call("ora_forces_fluid")
    def verlet_rebuild(self): self._call("ora_verlet_rebuild")
    def dem_substep(self): self._call("ora_dem_substep")
    def set_threads(self, n): self._call("ora_set_threads", C.c_int(n))
    def set_reduction(self, v): self._call("ora_set_reduction", C.c_double(v))
    def set_lid(self, uw_h): self._call("ora_set_lid", C.c_double(uw_h))

    def set_physics(self, p29, updateVerlet, stepFilm):
        p = np.ascontiguousarray(p29, dtype=np.float64); assert p.shape == (29,)
        self._call("ora_set_physics", _vp(p), C.c_int(updateVerlet), C.c_int(stepFilm))
    def set_nbsteps(self, n): self._call("ora_set_nbsteps", C.c_long(n))
    @property
    def nbsteps(self): return int(self._call("ora_nbsteps"))

    # state (views into the oracle's memory, host layout [lx][ly][9])
    def f_view(self):
        p = self.L.ora_f(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.lx, self.ly, 9))

    def delta_view(self):
        p = self.L.ora_delta(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.lx, self.ly, 9))

    def obst_view(self):
        p = self.L.ora_obst(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int)), shape=(self.lx, self.ly))

    def act_view(self):
        p = self.L.ora_act(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int)), shape=(self.lx, self.ly))

    def get_f(self): return self.f_view().copy()
    def set_f(self, f): self.f_view()[...] = f
    def get_obst(self): return self.obst_view().copy()
    def get_act(self): return self.act_view().copy()
    def get_delta(self): return self.delta_view().copy()

    def get_fhf(self):
        out = np.zeros((self.n, 3)); self._call("ora_get_fhf", _vp(out)); return out

    def set_fhf(self, fhf):
        fhf = np.ascontiguousarray(fhf, dtype=np.float64); assert fhf.shape == (self.n, 3)
        self._call("ora_set_fhf", _vp(fhf))

    def get_grains(self):
        out = np.zeros((self.n, GRAIN_COLS)); self._call("ora_get_grains", _vp(out)); return out

    def set_kinematics(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64); assert k.shape == (self.n, 9)
        self._call("ora_set_kinematics", _vp(k))

    def scalars(self):
        out = np.zeros(12); self._call("ora_get_scalars", _vp(out))
        d = dict(zip(SCALARS, out)); d["npDEM"] = int(d["npDEM"]); return d

    def rlb(self):
        out = np.zeros(self.n); self._call("ora_get_rlb", _vp(out)); return out

    def verlet(self):
        cap = int(self._call("ora_verlet_capacity"))
        cumul = np.zeros(self.n, np.int32); neigh = np.zeros(cap, np.int32); cnt = np.zeros(4, np.int32)
        w = [np.zeros(self.n, np.int32) for _ in range(4)]
        self._call("ora_get_verlet", _vp(cumul), _vp(neigh), _vp(cnt), *[_vp(a) for a in w])
        return cumul, neigh, cnt, [w[k][:cnt[k]].copy() for k in range(4)]

    def pairs(self):
        """Verlet pair set as an (npairs, 2) array of (i, j), i < j, reference list order."""
        cumul, neigh, _, _ = self.verlet()
        out = []
        for i in range(self.n):
            start = 0 if i == 0 else int(cumul[i - 1])
            for k in range(start, int(cumul[i])):
                out.append((i, int(neigh[k])))
        return np.array(out, dtype=np.int64).reshape(-1, 2)

    def total_density(self): return float(self.L.ora_total_density(self.h))

    # test-only helpers of the strip-decomposition protocol (tests/strip_backends.py)
    def link_sums(self, i, nlo, nhi):
        out = np.empty((512, 4))
        n = int(self.L.ora_link_sums(self.h, C.c_int(i), C.c_int(nlo), C.c_int(nhi), _vp(out), C.c_int(len(out))))
        if n > len(out):
            out = np.empty((n, 4))
            self.L.ora_link_sums(self.h, C.c_int(i), C.c_int(nlo), C.c_int(nhi), _vp(out), C.c_int(n))
        return out[:n].copy()

    def force_from_link_sums(self, i, terms):
        t = np.ascontiguousarray(terms, dtype=np.float64)
        self.L.ora_force_from_link_sums(self.h, C.c_int(i), _vp(t), C.c_int(len(t)))
    def act_anomalies(self): return int(self.L.ora_count_act_anomalies(self.h))


class Reference:
    """The unmodified reference TU, compiled for one (lx, ly). One instance per process:
    the reference keeps its state in globals."""

    def __init__(self, lx, ly, sample_path, fast=False, sp=False, dry=False):
        path = build_ref(lx, ly, fast, sp, dry)
        if path is None:
            raise FileNotFoundError("reference build not available")
        self.L = C.CDLL(path)
        L = self.L
        L.ref_total_density.restype = C.c_double
        L.ref_nbsteps.restype = C.c_long
        self.lx, self.ly = lx, ly
        rc = L.ref_init(os.fsencode(sample_path))
        if rc != 0:
            raise RuntimeError(f"ref_init -> {rc}")
        self.n = L.ref_nbgrains()

    def steps(self, n): self.L.ref_steps(C.c_long(n))
    def lbm_steps(self, n): self.L.ref_lbm_steps(C.c_int(n))
    def reinit(self): self.L.ref_reinit_obst_density()
    def obst_construction(self): self.L.ref_obst_construction()
    def collision_streaming(self): self.L.ref_collision_streaming()
    def forces_fluid(self): self.L.ref_forces_fluid()
    def verlet_rebuild(self): self.L.ref_init_verlet()
    def set_nbsteps(self, n): self.L.ref_set_nbsteps(C.c_long(n))
    @property
    def nbsteps(self): return int(self.L.ref_nbsteps())

    def get_f(self):
        out = np.zeros((self.lx, self.ly, 9)); self.L.ref_get_f(_vp(out)); return out

    def set_f(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64); self.L.ref_set_f(_vp(f))

    def get_obst(self):
        out = np.zeros((self.lx, self.ly), np.int32); self.L.ref_get_obst(_vp(out)); return out

    def get_act(self):
        out = np.zeros((self.lx, self.ly), np.int32); self.L.ref_get_act(_vp(out)); return out

    def get_delta(self):
        out = np.zeros((self.lx, self.ly, 9)); self.L.ref_get_delta(_vp(out)); return out

    def get_fhf(self):
        out = np.zeros((self.n, 3)); self.L.ref_get_fhf(_vp(out)); return out

    def get_grains(self):
        out = np.zeros((self.n, GRAIN_COLS)); self.L.ref_get_grains(_vp(out)); return out

    def set_kinematics(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64); assert k.shape == (self.n, 9)
        self.L.ref_set_kinematics(_vp(k))

    def scalars(self):
        out = np.zeros(12); self.L.ref_get_scalars(_vp(out))
        d = dict(zip(SCALARS, out)); d["npDEM"] = int(d["npDEM"]); return d

    def rlb(self):
        out = np.zeros(self.n); self.L.ref_get_rlb(_vp(out)); return out

    def verlet(self):
        cumul = np.zeros(self.n, np.int32); neigh = np.zeros(6 * self.n, np.int32); cnt = np.zeros(4, np.int32)
        w = [np.zeros(self.n, np.int32) for _ in range(4)]
        self.L.ref_get_verlet(_vp(cumul), _vp(neigh), _vp(cnt), *[_vp(a) for a in w])
        return cumul, neigh, cnt, [w[k][:cnt[k]].copy() for k in range(4)]

    def total_density(self): return float(self.L.ref_total_density())


def read_sample(path):
    """ora_read_sample wrapper -> (r, x1, x2) in metres."""
    L = C.CDLL(build_oracle())
    n = C.c_int(0)
    pr, p1, p2 = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    rc = L.ora_read_sample(os.fsencode(path), C.byref(n), C.byref(pr), C.byref(p1), C.byref(p2))
    if rc != 0:
        raise RuntimeError(f"ora_read_sample({path}) -> {rc}")
    out = [np.ctypeslib.as_array(p, shape=(n.value,)).copy() for p in (pr, p1, p2)]
    L.ora_free.argtypes = [C.c_void_p]
    for p in (pr, p1, p2):
        L.ora_free(C.cast(p, C.c_void_p))
    return tuple(out)


def write_sample(path, r_mm, x_mm, y_mm, comment="#synthetic sample"):
    """Write a sample in the reference's .data format (main.c:612-622): units of 1 mm."""
    with open(path, "w") as fp:
        fp.write(comment.rstrip("\n") + "\n")
        fp.write(f"{len(r_mm)}\n")
        for a, b, c in zip(r_mm, x_mm, y_mm):
            fp.write(f"{a:.17e}\t{b:.17e}\t{c:.17e}\n")
