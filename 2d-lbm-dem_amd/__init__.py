"""2d-lbm-dem_amd -- MI355X-native hot path of cb-geo/2d-lbm-dem behind the reference's own seam.

The reference exposes no library API: its seam is the set of ``void fn(void)`` routines that
``renderScene()`` calls on file-scope globals (src/main.c:1697-1777). :class:`LbmDem` mirrors that
seam one-to-one -- same routine names, same call order, same host data layout
(``f[x][y][q]``, x slow) -- on top of the C ABI in ``include/lbmdem_hip.h``
(``liblbmdem_hip.so``, hand-written HIP for gfx950).

There is deliberately NO CPU fallback here: if the HIP library is missing or no GPU is present
every entry point raises. The CPU oracle lives under ``oracle/`` and is test infrastructure only.

The directory name is not a Python identifier; load it with ``importlib`` (see
``__graft_entry__.load_package()``) or put the repo root on ``sys.path`` and use
``importlib.import_module("2d-lbm-dem_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBMDEM_HIP_LIBRARY: load another build of the same ABI (e.g. the `make AB=1` experiment build)
LIB_PATH = os.environ.get("LBMDEM_HIP_LIBRARY") or os.path.join(_HERE, "liblbmdem_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lbmdem_hip.h")


class LbmDemError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class Physics(C.Structure):
    """lbmdem_physics -- the initialised globals of main.c:74-118,143,163-165."""
    _fields_ = [(n, C.c_double) for n in (
        "rho_moy tau s2 s3 s5 s7 s8 s9 nu reductionR G angleG km kg kt ktm nug num nugt "
        "mu mum mumb murf distVerlet dtt iterDEM freq amp t").split()] + [
        ("updateVerlet", C.c_int), ("stepFilm", C.c_int)]


class Config(C.Structure):
    """lbmdem_config"""
    _fields_ = [("lx", C.c_int), ("ly", C.c_int), ("x_begin", C.c_int), ("x_end", C.c_int),
                ("halo", C.c_int), ("device", C.c_int), ("nbgrains", C.c_int), ("scale", C.c_double),
                ("dx", C.c_double), ("dtLB", C.c_double), ("c", C.c_double), ("dt", C.c_double),
                ("dt2", C.c_double), ("npDEM", C.c_int),
                ("Mgx", C.c_double), ("Mdx", C.c_double), ("Mby", C.c_double), ("Mhy", C.c_double),
                ("xG", C.c_double), ("yG", C.c_double), ("phys", Physics)]


_lib = None
_lib_sp = None
SP_LIB_PATH = os.path.join(_HERE, "liblbmdem_hip_sp.so")   # `real` = float: the reference's -DSINGLE_PRECISION mode


def load_library(precision="f64"):
    """dlopen liblbmdem_hip.so (precision "f32": liblbmdem_hip_sp.so, same ABI); fails loudly when it has not been built."""
    global _lib, _lib_sp
    if precision == "f32":
        if _lib_sp is None:
            load_library()       # runtime set-up (torch first, kernel arguments) happens with the default library
            _lib_sp = _open_library(SP_LIB_PATH)
        return _lib_sp
    if _lib is not None:
        return _lib
    _lib = _open_library(LIB_PATH)
    return _lib


def _open_library(LIB_PATH):
    if not os.path.exists(LIB_PATH):
        raise LbmDemError(-2, f"{LIB_PATH} not built -- run __graft_entry__.build() "
                              "(make -C 2d-lbm-dem_amd/csrc); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (requested by the
    # unversioned name), liblbmdem_hip.so links the system one (libamdhip64.so.7). If ours were
    # loaded first and torch later (the strip driver uses torch.distributed), the process would end
    # up with two runtimes and torch would see no GPU. Importing torch first makes both share one.
    # Launch latency: the HIP runtime places kernel arguments in host memory by default on this stack; a step
    # is ~17 dependent launches, 12 of them ~8 us DEM sub-steps, and device-resident arguments
    # (HIP_FORCE_DEV_KERNARG, read when the runtime initialises) take ~1.2 us off each (measured: 1.182 ->
    # 1.163 ms per coupled step). Only a default: an explicit setting of the caller wins.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.lbmdem_last_error.restype = C.c_char_p
    L.lbmdem_version.restype = C.c_char_p
    L.lbmdem_nbsteps.restype = C.c_long
    L.lbmdem_nbsteps.argtypes = [C.c_void_p]
    L.lbmdem_halo_doubles.restype = C.c_long
    L.lbmdem_halo_doubles.argtypes = [C.c_void_p]
    L.lbmdem_free_host.argtypes = [C.c_void_p]
    L.lbmdem_free_host.restype = None
    L.lbmdem_run.argtypes = [C.c_void_p, C.c_long]
    L.lbmdem_run_dem.argtypes = [C.c_void_p, C.c_long]
    L.lbmdem_set_nbsteps.argtypes = [C.c_void_p, C.c_long]
    L.lbmdem_derive.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    L.lbmdem_create.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    for name in ("destroy", "lbm_step", "obst_construction", "collide_stream", "forces_fluid",
                 "verlet_rebuild", "dem_substep", "sync", "use_own_stream"):
        getattr(L, "lbmdem_" + name).argtypes = [C.c_void_p]
    for name in ("upload_f", "download_f", "download_obst", "total_density", "upload_kinematics",
                 "download_kinematics", "download_fhf", "set_stream"):
        getattr(L, "lbmdem_" + name).argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_path_info.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_fused_work_order.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_dist_export_carries.argtypes = [C.c_void_p] * 4
    L.lbmdem_dist_set_carries.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_dist_export_owned.argtypes = [C.c_void_p] * 5
    L.lbmdem_dist_table_substep.argtypes = [C.c_void_p] * 4
    L.lbmdem_vtk_place_owned.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_write_vtk_fields.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lbmdem_total_density_serial.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.lbmdem_download_macro.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lbmdem_download_verlet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.lbmdem_download_grain_pressure.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_download_vtk_fields.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.lbmdem_write_vtk.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.lbmdem_set_diagnostics.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_download_grain_table.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_write_dem.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
    L.lbmdem_write_forces.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.lbmdem_collide_stream_part.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_checkpoint_save.argtypes = [C.c_void_p, C.c_char_p]
    L.lbmdem_checkpoint_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.lbmdem_set_force_mode.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_set_dem_chain.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_dem_chain_paints.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    if hasattr(L, "lbmdem_set_dem_tiles"):
        L.lbmdem_set_dem_tiles.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "lbmdem_dem_chain_recoveries"):   # (older builds kept for A/B runs do not have it)
        L.lbmdem_dem_chain_recoveries.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    if hasattr(L, "lbmdem_debug_chain_giveup"):   # experiment build only
        L.lbmdem_debug_chain_giveup.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_set_obst_update.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_obst_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.lbmdem_set_change_mask.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_change_mask_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.lbmdem_measure_copy.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.lbmdem_dem_chain_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lbmdem_set_lid.argtypes = [C.c_void_p, C.c_double]
    L.lbmdem_force_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lbmdem_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lbmdem_halo_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.lbmdem_halo_unpack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.lbmdem_dist_default_margin.argtypes = [C.c_void_p]
    L.lbmdem_dist_enable.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_dist_message_doubles.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_dist_message_doubles.restype = C.c_long
    L.lbmdem_dist_begin_period.argtypes = [C.c_void_p]
    L.lbmdem_dist_pack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.lbmdem_dist_unpack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.lbmdem_dist_set_poison.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_dist_pack2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.lbmdem_dist_unpack2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.lbmdem_halo_pack2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lbmdem_halo_unpack2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lbmdem_comm_unique_id.argtypes = [C.c_void_p]
    L.lbmdem_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.lbmdem_comm_destroy.argtypes = [C.c_void_p]
    L.lbmdem_comm_lbm_step.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_comm_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.lbmdem_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.lbmdem_comm_selftest.argtypes = [C.c_void_p, C.c_int]
    L.lbmdem_fhf_export.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_fhf_import.argtypes = [C.c_void_p, C.c_void_p]
    L.lbmdem_fhf_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.lbmdem_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.lbmdem_read_sample.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)),
                                     C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double))]
    return L


def strips_module():
    """The x-strip decomposition driver (one process per GPU), imported lazily."""
    from . import strips
    return strips


def write_vtk_fields(directory, nFile, lx, ly, fields11):
    """the five VTK files of write_vtk (main.c:237-338) from merged lattice-sized fields (LbmDem.vtk_place_owned)"""
    f = np.ascontiguousarray(fields11, dtype=np.float32)
    _chk(load_library().lbmdem_write_vtk_fields(os.fsencode(directory), int(nFile), int(lx), int(ly), _vp(f)))


def exported_symbols():
    """Names the public header declares (used by the CPU test that checks the ABI surface)."""
    import re
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(lbmdem_[a-z_0-9]+)\s*\(", txt)))


def _chk(rc):
    if rc != 0:
        msg = load_library().lbmdem_last_error().decode(errors="replace")
        if _lib_sp is not None:      # the float build keeps its own error text
            sp = _lib_sp.lbmdem_last_error().decode(errors="replace")
            msg = sp if not msg else (msg if not sp else f"{msg} | f32 library: {sp}")
        raise LbmDemError(rc, msg)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def read_sample(path, precision="f64"):
    """read_sample (main.c:609-639) -> (r, x1, x2) in metres. precision "f32": parsed as the reference's
    -DSINGLE_PRECISION build parses it (text -> float, times the float 1e-3)."""
    L = load_library(precision)
    n = C.c_int(0)
    pr, p1, p2 = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    _chk(L.lbmdem_read_sample(os.fsencode(path), C.byref(n), C.byref(pr), C.byref(p1), C.byref(p2)))
    out = tuple(np.ctypeslib.as_array(p, shape=(n.value,)).copy() for p in (pr, p1, p2))
    for p in (pr, p1, p2):
        L.lbmdem_free_host(C.cast(p, C.c_void_p))
    return out


def derive(lx, ly, r, scale=1.0, physics: Physics | None = None, precision="f64") -> Config:
    """Time-step derivation of main.c:1836-1860 (host arithmetic, bit-identical; "f32": in the float build's types)."""
    L = load_library(precision)
    cfg = Config()
    if physics is None:
        _chk(L.lbmdem_physics_defaults(C.byref(cfg.phys)))
    else:
        cfg.phys = physics
    r = np.ascontiguousarray(r, dtype=np.float64)
    _chk(L.lbmdem_derive(C.byref(cfg), int(lx), int(ly), float(scale), len(r), _vp(r)))
    cfg.x_begin, cfg.x_end, cfg.halo, cfg.device = 0, int(lx), 0, 0
    return cfg


COMM_ID_BYTES = 512


def comm_unique_id() -> bytes:
    """RCCL ids for one communicator group (rank 0 makes them, every rank gets the same bytes)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(load_library().lbmdem_comm_unique_id(buf))
    return buf.raw


class Comm:
    """The library's own RCCL transport (lbmdem_comm_*): neighbour send/recv of the distributed-grain protocol, driven
    from C -- one call per renderScene batch, no Python on the step path."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        self._L = load_library()
        self._c = C.c_void_p()
        assert len(unique_id) == COMM_ID_BYTES
        _chk(self._L.lbmdem_comm_create(C.c_char_p(unique_id), int(rank), int(world), int(device), C.byref(self._c)))

    def close(self):
        if self._c:
            self._L.lbmdem_comm_destroy(self._c)
            self._c = C.c_void_p()

    def run(self, sim, n_dem_steps):
        _chk(self._L.lbmdem_comm_run(sim._h, self._c, int(n_dem_steps)))

    def allreduce_sum(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64)
        _chk(self._L.lbmdem_comm_allreduce_sum(self._c, _vp(a), len(a)))
        return a

    def selftest(self, doubles=4096):
        _chk(self._L.lbmdem_comm_selftest(self._c, int(doubles)))


class LbmDem:
    """One simulation on one GPU (or one x-strip of it). Method names follow the reference.

    >>> sim = LbmDem.from_sample("a08d83.data", lx=600, ly=500)
    >>> for _ in range(240): sim.renderScene()
    >>> f = sim.f            # [lx][ly][9], the reference's host layout
    """

    def __init__(self, lx, ly, r, x1, x2, scale=1.0, device=0, strip=None, halo=0, physics=None, precision="f64"):
        """precision "f32": the float build of the library (the reference's -DSINGLE_PRECISION mode, main.c:34-40): same
        methods, host arrays stay float64 (holding float values); step path and state transfers only."""
        L = load_library(precision)
        self._L = L
        self.precision = precision
        self._h = C.c_void_p()
        r = np.ascontiguousarray(r, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        if not (len(r) == len(x1) == len(x2)) or len(r) < 1:
            raise LbmDemError(-1, "r, x1, x2 must be equally long, at least one grain "
                                  "(the reference cannot run with 0 grains either: main.c:220)")
        cfg = derive(lx, ly, r, scale, physics, precision)
        cfg.device = int(device)
        if strip is not None:
            cfg.x_begin, cfg.x_end, cfg.halo = int(strip[0]), int(strip[1]), int(halo)
        self.cfg = cfg
        self.lx, self.ly, self.n = int(lx), int(ly), len(r)
        _chk(L.lbmdem_create(C.byref(cfg), _vp(r), _vp(x1), _vp(x2), C.byref(self._h)))

    def checkpoint_save(self, path):
        _chk(self._L.lbmdem_checkpoint_save(self._h, os.fsencode(path)))

    @classmethod
    def checkpoint_load(cls, path, device=0):
        """Restart: a new simulation continuing bit-identically from a checkpoint."""
        L = load_library()
        self = cls.__new__(cls)
        self._L, self._h = L, C.c_void_p()
        _chk(L.lbmdem_checkpoint_load(os.fsencode(path), int(device), C.byref(self._h)))
        self.cfg = self.config()
        self.lx, self.ly, self.n = self.cfg.lx, self.cfg.ly, self.cfg.nbgrains
        return self

    @classmethod
    def from_sample(cls, path, lx, ly, **kw):
        r, x1, x2 = read_sample(path)
        return cls(lx, ly, r, x1, x2, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._L.lbmdem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- the reference's routines -------------------------------------------------------------
    def renderScene(self, n=1):
        """n x renderScene() (main.c:1697-1765)."""
        _chk(self._L.lbmdem_run(self._h, int(n)))

    def run_dem(self, n=1):
        """n x (Verlet rebuild when due; DEM sub-step) -- renderScene without its fluid step."""
        _chk(self._L.lbmdem_run_dem(self._h, int(n)))

    def renderScene_dry(self, n=1):
        """n x renderScene() of the reference compiled without `#define _FLUIDE_` (main.c:16, 1709-1719): DEM only -- no
        fluid step, the hydrodynamic forces keep their values (0 from the start)."""
        self.run_dem(n)

    def lbm_step(self):
        """reinit_obst_density + obst_construction + collision_streaming + forces_fluid (main.c:1711-1717)."""
        _chk(self._L.lbmdem_lbm_step(self._h))

    def obst_construction(self):
        _chk(self._L.lbmdem_obst_construction(self._h))

    def collision_streaming(self):
        """reinit_obst_density (with the previous obstacle map) + collision_streaming."""
        _chk(self._L.lbmdem_collide_stream(self._h))

    def collision_streaming_edges(self):
        """First part of a split collision_streaming: the owned rows a neighbouring strip needs."""
        _chk(self._L.lbmdem_collide_stream_part(self._h, 1))

    def collision_streaming_interior(self):
        """Second part: all other owned rows (runs while the halo rows travel)."""
        _chk(self._L.lbmdem_collide_stream_part(self._h, 2))

    def forces_fluid(self):
        _chk(self._L.lbmdem_forces_fluid(self._h))

    def initVerlet(self):
        """initVerlet + VerletWall (main.c:1519-1594)."""
        _chk(self._L.lbmdem_verlet_rebuild(self._h))

    VerletWall = initVerlet

    def dem_substep(self):
        """the integrator + acceleration_grains block of renderScene (main.c:1733-1764)."""
        _chk(self._L.lbmdem_dem_substep(self._h))

    def final_density(self, sum_in=0.0):
        """check_density / final_density (main.c:1249-1273) with the reference's bits: the serial chain over f[x][y][q],
        continued from `sum_in` over the owned rows. `self.density_rows_replayed` = rows added element by element."""
        s, n = C.c_double(0), C.c_int(0)
        _chk(self._L.lbmdem_total_density_serial(self._h, C.c_double(sum_in), C.byref(s), C.byref(n)))
        self.density_rows_replayed = n.value
        return s.value

    check_density = final_density

    def total_density_tree(self):
        """the same sum in tree order (one pass, last bits differ from the reference's)"""
        s = C.c_double(0)
        _chk(self._L.lbmdem_total_density(self._h, C.byref(s)))
        return s.value

    # ---- state ----------------------------------------------------------------------------------
    @property
    def nbsteps(self):
        return int(self._L.lbmdem_nbsteps(self._h))

    @nbsteps.setter
    def nbsteps(self, v):
        _chk(self._L.lbmdem_set_nbsteps(self._h, int(v)))

    @property
    def f(self):
        out = np.zeros((self.lx, self.ly, 9))
        _chk(self._L.lbmdem_download_f(self._h, _vp(out)))
        return out

    @f.setter
    def f(self, value):
        a = np.ascontiguousarray(value, dtype=np.float64)
        if a.shape != (self.lx, self.ly, 9):
            raise LbmDemError(-1, f"f must have shape {(self.lx, self.ly, 9)}")
        _chk(self._L.lbmdem_upload_f(self._h, _vp(a)))

    def download_f_into(self, out):
        _chk(self._L.lbmdem_download_f(self._h, _vp(out)))

    @property
    def obst(self):
        out = np.zeros((self.lx, self.ly), dtype=np.int32)
        _chk(self._L.lbmdem_download_obst(self._h, _vp(out)))
        return out

    def macro(self):
        rho, ux, uy = (np.zeros((self.lx, self.ly)) for _ in range(3))
        _chk(self._L.lbmdem_download_macro(self._h, _vp(rho), _vp(ux), _vp(uy)))
        return rho, ux, uy

    @property
    def kinematics(self):
        """(n, 9): x1 x2 x3 v1 v2 v3 a1 a2 a3"""
        out = np.zeros((self.n, 9))
        _chk(self._L.lbmdem_download_kinematics(self._h, _vp(out)))
        return out

    @kinematics.setter
    def kinematics(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64)
        if k.shape != (self.n, 9):
            raise LbmDemError(-1, f"kinematics must have shape {(self.n, 9)}")
        _chk(self._L.lbmdem_upload_kinematics(self._h, _vp(k)))

    @property
    def fhf(self):
        out = np.zeros((self.n, 3))
        _chk(self._L.lbmdem_download_fhf(self._h, _vp(out)))
        return out

    @property
    def grain_pressure(self):
        out = np.zeros(self.n)
        _chk(self._L.lbmdem_download_grain_pressure(self._h, _vp(out)))
        return out

    def set_diagnostics(self, always=True):
        _chk(self._L.lbmdem_set_diagnostics(self._h, int(bool(always))))

    def grain_table(self):
        """(n, 30) in the reference's struct order (main.c:182-197)."""
        out = np.zeros((self.n, 30))
        _chk(self._L.lbmdem_download_grain_table(self._h, _vp(out)))
        return out

    def write_DEM(self, directory=".", nFile=0):
        """write_DEM (main.c:340-438): DEM%06d.dat + a line of stats.data.
        -> (KE, PE, SE, IFR, WF, INCE, TSLIP, TRW)"""
        e = np.zeros(8)
        _chk(self._L.lbmdem_write_dem(self._h, os.fsencode(directory), int(nFile), _vp(e)))
        return tuple(e)

    def write_forces(self, directory=".", nFile=0):
        """write_forces (main.c:440-478): DEM%06d.ps, grains + one line per overlapping pair."""
        _chk(self._L.lbmdem_write_forces(self._h, os.fsencode(directory), int(nFile)))

    def write_vtk(self, directory=".", nFile=0):
        """write_vtk (main.c:237-338): five binary legacy-VTK files, byte-identical to the reference's."""
        _chk(self._L.lbmdem_write_vtk(self._h, os.fsencode(directory), int(nFile)))

    def vtk_fields(self):
        nx = self.cfg.x_end - self.cfg.x_begin
        gp = np.zeros((self.ly, nx), np.float32); fp = np.zeros((self.ly, nx), np.float32)
        gv = np.zeros((self.ly, nx, 3), np.float32); ga = np.zeros((self.ly, nx, 3), np.float32)
        fv = np.zeros((self.ly, nx, 3), np.float32)
        _chk(self._L.lbmdem_download_vtk_fields(self._h, _vp(gp), _vp(gv), _vp(ga), _vp(fp), _vp(fv)))
        return gp, gv, ga, fp, fv

    def verlet(self):
        """-> (cumul[n], neighbours[npairs], wallflags[n]) in the reference's form."""
        npairs = C.c_int(0)
        cumul = np.zeros(self.n, np.int32)
        wf = np.zeros(self.n, np.int32)
        _chk(self._L.lbmdem_download_verlet(self._h, None, None, 0, C.byref(npairs), None))
        neigh = np.zeros(max(npairs.value, 1), np.int32)
        _chk(self._L.lbmdem_download_verlet(self._h, _vp(cumul), _vp(neigh), len(neigh), C.byref(npairs), _vp(wf)))
        return cumul, neigh[:npairs.value], wf

    def config(self) -> Config:
        out = Config()
        _chk(self._L.lbmdem_get_config(self._h, C.byref(out)))
        return out

    # ---- plumbing -------------------------------------------------------------------------------
    def set_lid(self, uw_h):
        """EXTENSION: the top plate's lid terms the reference has commented out (main.c:1129-1130); lattice units."""
        _chk(self._L.lbmdem_set_lid(self._h, float(uw_h)))

    def set_force_mode(self, mode):
        _chk(self._L.lbmdem_set_force_mode(self._h, int(mode)))

    def set_dem_chain(self, max_substeps):
        """Longest run of ordinary sub-steps renderScene hands to ONE launch (< 2: one launch per sub-step)."""
        _chk(self._L.lbmdem_set_dem_chain(self._h, int(max_substeps)))

    def set_obst_update(self, on=True):
        """obst_construction writes only the nodes whose owner changes (default) / clears and repaints the map (False)"""
        _chk(self._L.lbmdem_set_obst_update(self._h, 1 if on else 0))

    def obst_stats(self):
        """(rasterisations that updated the map in place, rasterisations that cleared and repainted it)"""
        a, b = C.c_long(0), C.c_long(0)
        _chk(self._L.lbmdem_obst_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_change_mask(self, mode=1):
        """0: the fused kernel reads both obstacle maps everywhere; 1 (default): the previous map only in the rows the
        rasterisation marked as changed; 2: as 1, every use verified against the two maps (change_mask_stats()[1] == 0)"""
        _chk(self._L.lbmdem_set_change_mask(self._h, int(mode)))

    def change_mask_stats(self):
        """(fused launches that used the change bits, (row, window) pairs mode 2 found clear over differing maps)"""
        a, b = C.c_long(0), C.c_long(0)
        _chk(self._L.lbmdem_change_mask_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def measure_copy(self, nbytes=1200 * 1000 * 1000, reps=5):
        """GB/s (read + written) of a plain copy of `nbytes` on this handle's device: the box's yardstick"""
        g = C.c_double(0.0)
        _chk(self._L.lbmdem_measure_copy(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def dem_chain_stats(self):
        """(launches, sub-steps covered, tile slots a launch needs resident, slots the census found; -1 = not taken)"""
        a, b, c, d = C.c_long(0), C.c_long(0), C.c_int(0), C.c_int(0)
        _chk(self._L.lbmdem_dem_chain_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def dem_chain_paints(self):
        """rasterisations done by the tail of a run of sub-steps instead of a launch of their own"""
        a = C.c_long(0)
        _chk(self._L.lbmdem_dem_chain_paints(self._h, C.byref(a)))
        return a.value

    def set_dem_tiles(self, mode):
        """1: the tiles of the multi-sub-step DEM kernel are patches of the packing (default); 0: 64 consecutive indices"""
        _chk(self._L.lbmdem_set_dem_tiles(self._h, int(mode)))

    def dem_chain_recoveries(self):
        """launches of the multi-sub-step kernel that gave up and were undone (the run went on one launch per sub-step)"""
        if not hasattr(self._L, "lbmdem_dem_chain_recoveries"):
            return 0
        a = C.c_long(0)
        _chk(self._L.lbmdem_dem_chain_recoveries(self._h, C.byref(a)))
        return a.value

    def debug_chain_giveup(self, launch):
        """experiment build: the launch of the multi-sub-step kernel with this number (from 0) gives up half way"""
        _chk(self._L.lbmdem_debug_chain_giveup(self._h, int(launch)))

    def force_stats(self):
        """(grains summed from the fused kernel's link table, grains gathered from the lattice) of the last
        forces_fluid call."""
        a, b = C.c_int(0), C.c_int(0)
        _chk(self._L.lbmdem_force_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_stream(self, hip_stream_ptr):
        """Enqueue on a caller-owned hipStream_t; 0/None = the HIP default stream."""
        _chk(self._L.lbmdem_set_stream(self._h, C.c_void_p(hip_stream_ptr or None)))

    def use_own_stream(self):
        _chk(self._L.lbmdem_use_own_stream(self._h))

    def sync(self):
        _chk(self._L.lbmdem_sync(self._h))

    def profile_enable(self, on=True):
        """HIP-event timing of the fused kernel; on = N > 1: every N-th launch only (two event records cost a step ~10 us)"""
        _chk(self._L.lbmdem_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, cnt = C.c_double(0), C.c_long(0)
        _chk(self._L.lbmdem_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def halo_doubles(self):
        return int(self._L.lbmdem_halo_doubles(self._h))

    def halo_pack(self, side, dev_ptr):
        _chk(self._L.lbmdem_halo_pack(self._h, int(side), C.c_void_p(dev_ptr)))

    def halo_unpack(self, side, dev_ptr):
        _chk(self._L.lbmdem_halo_unpack(self._h, int(side), C.c_void_p(dev_ptr)))

    # ---- strips with distributed grains (include/lbmdem_hip.h: lbmdem_dist_*) --------------------------------------
    MSG_KIN, MSG_FHF, MSG_TABLES = 0, 1, 2

    def dist_default_margin(self):
        return int(self._L.lbmdem_dist_default_margin(self._h))

    def dist_enable(self, margin_rows=0):
        _chk(self._L.lbmdem_dist_enable(self._h, int(margin_rows)))

    def dist_message_doubles(self, kind):
        return int(self._L.lbmdem_dist_message_doubles(self._h, int(kind)))

    def dist_begin_period(self):
        _chk(self._L.lbmdem_dist_begin_period(self._h))

    def dist_pack(self, kind, side, dev_ptr):
        _chk(self._L.lbmdem_dist_pack(self._h, int(kind), int(side), C.c_void_p(dev_ptr)))

    def dist_unpack(self, kind, side, dev_ptr):
        _chk(self._L.lbmdem_dist_unpack(self._h, int(kind), int(side), C.c_void_p(dev_ptr)))

    def dist_pack2(self, kind, ptr_lo, ptr_hi):
        """both sides in one launch; None skips a side"""
        _chk(self._L.lbmdem_dist_pack2(self._h, int(kind), C.c_void_p(ptr_lo), C.c_void_p(ptr_hi)))

    def dist_unpack2(self, kind, ptr_lo, ptr_hi):
        _chk(self._L.lbmdem_dist_unpack2(self._h, int(kind), C.c_void_p(ptr_lo), C.c_void_p(ptr_hi)))

    def halo_pack2(self, ptr_lo, ptr_hi):
        _chk(self._L.lbmdem_halo_pack2(self._h, C.c_void_p(ptr_lo), C.c_void_p(ptr_hi)))

    def halo_unpack2(self, ptr_lo, ptr_hi):
        _chk(self._L.lbmdem_halo_unpack2(self._h, C.c_void_p(ptr_lo), C.c_void_p(ptr_hi)))

    # ---- drop-in outputs of a strip decomposition (include/lbmdem_hip.h) ----------------------------------
    def dist_export_carries(self):
        keys = np.zeros((3, 2), np.int64); vals = np.zeros(3); standing = np.zeros(3)
        _chk(self._L.lbmdem_dist_export_carries(self._h, _vp(keys), _vp(vals), _vp(standing)))
        return keys, vals, standing

    def dist_set_carries(self, carry3):
        c = np.ascontiguousarray(carry3, dtype=np.float64)
        _chk(self._L.lbmdem_dist_set_carries(self._h, _vp(c)))

    def path_info(self):
        """which size-dependent fast paths are active: dict(table, slots_per_direction, mincov, marching)"""
        v = (C.c_int * 4)()
        _chk(self._L.lbmdem_path_info(self._h, v))
        return dict(table=bool(v[0]), slots_per_direction=int(v[1]), mincov=bool(v[2]), marching=bool(v[3]))

    def fused_work_order(self):
        """how the fused kernel's launch is cut into work items: dict(levels, band_rows, chunk_rows, segment_rows [per level],
        level_rows [band rows cut at each level], items); levels == 0: uniform segments of segment_rows[0] rows"""
        v = (C.c_int * 12)()
        _chk(self._L.lbmdem_fused_work_order(self._h, v))
        nl = max(int(v[0]), 1)
        return dict(levels=int(v[0]), band_rows=int(v[1]), chunk_rows=int(v[2]), segment_rows=[int(v[3 + k]) for k in range(nl)],
                    level_rows=[int(v[7 + k]) for k in range(nl)], items=int(v[11]))

    def dist_export_owned(self):
        """-> (state12 [n][12], owned [n] uint8, carry_keys [3][2] int64, carry_vals [3]): what this rank contributes
        to the sub-step that feeds write_DEM (strips.merge_exports combines the ranks' exports)."""
        st = np.zeros((self.n, 12)); owned = np.zeros(self.n, np.uint8)
        keys = np.zeros((3, 2), np.int64); vals = np.zeros(3)
        _chk(self._L.lbmdem_dist_export_owned(self._h, _vp(st), _vp(owned), _vp(keys), _vp(vals)))
        return st, owned, keys, vals

    def dist_table_substep(self, state12_full, carry_vals, carry_has):
        """the root's sub-step on the full replica; afterwards grain_table / write_DEM / write_forces work here"""
        st = np.ascontiguousarray(state12_full, dtype=np.float64)
        cv = np.ascontiguousarray(carry_vals, dtype=np.float64); ch = np.ascontiguousarray(carry_has, dtype=np.int32)
        assert st.shape == (self.n, 12)
        _chk(self._L.lbmdem_dist_table_substep(self._h, _vp(st), _vp(cv), _vp(ch)))

    def vtk_place_owned(self, fields11):
        """this strip's columns of the five write_vtk fields into a zero-initialised float32 array of 11 * lx * ly"""
        assert fields11.dtype == np.float32 and fields11.size == 11 * self.lx * self.ly and fields11.flags.c_contiguous
        _chk(self._L.lbmdem_vtk_place_owned(self._h, _vp(fields11)))

    def dist_set_poison(self, on=True):
        _chk(self._L.lbmdem_dist_set_poison(self._h, 1 if on else 0))

    def fhf_export(self, dev_ptr):
        _chk(self._L.lbmdem_fhf_export(self._h, C.c_void_p(dev_ptr)))

    def fhf_import(self, dev_ptr):
        _chk(self._L.lbmdem_fhf_import(self._h, C.c_void_p(dev_ptr)))

    def fhf_device(self):
        a, b = C.c_void_p(), C.c_void_p()
        _chk(self._L.lbmdem_fhf_device(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
