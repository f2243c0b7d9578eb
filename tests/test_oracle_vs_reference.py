"""CPU: the in-repo oracle against the unmodified reference TU compiled in this container
(oracle/ref_harness.c -> oracle/_ref/libref_<lx>x<ly>.so), driven live on the reference's own sample
files. Skipped where neither /root/reference nor a prebuilt reference library exists. One process
per case: the reference keeps its state in globals."""
import multiprocessing as mp
import os

import numpy as np
import pytest

REF_BIN = "/root/reference/bin"


def _worker(lx, ly, path, nsteps, q):
    import pyoracle as po
    try:
        R = po.Reference(lx, ly, path)
        O = po.Oracle.from_file(path, lx, ly)
        bad = []
        if R.scalars() != O.scalars(): bad.append("scalars")
        if not np.array_equal(R.rlb(), O.rlb()): bad.append("rLB")
        if not np.array_equal(R.get_obst(), O.get_obst()): bad.append("init_obst")
        for chunk in (1, nsteps - 1):
            R.steps(chunk); O.steps(chunk)
            for k in ("f", "obst", "act", "delta", "fhf"):
                if not np.array_equal(getattr(R, "get_" + k)(), getattr(O, "get_" + k)()): bad.append(k)
            gr, go = R.get_grains(), O.get_grains()
            bad += [n for n, c in po.COL.items() if not np.array_equal(gr[:, c], go[:, c], equal_nan=True)]
            rv, ov = R.verlet(), O.verlet()
            npairs = int(rv[0].max())
            if not (np.array_equal(rv[0], ov[0]) and np.array_equal(rv[1][:npairs], ov[1][:npairs])
                    and np.array_equal(rv[2], ov[2]) and all(np.array_equal(a, b) for a, b in zip(rv[3], ov[3]))):
                bad.append("verlet")
            if R.total_density() != O.total_density(): bad.append("total_density")
        q.put((bad, O.act_anomalies(), npairs))
    except Exception as e:  # pragma: no cover
        q.put(([repr(e)], -1, -1))


@pytest.mark.parametrize("lx,ly,sample,nsteps", [
    (600, 500, "a08d83.data", 230),              # 726 grains, npDEM 10: 23 fluid steps, 3 Verlet rebuilds
    (1600, 900, "a08_a4b4r18_7000.data", 49),    # 6355 grains: the packing with true IBB order hazards
])
def test_oracle_bit_equal_to_reference(po, lx, ly, sample, nsteps):
    path = os.path.join(REF_BIN, sample)
    if not os.path.exists(path) or po.build_ref(lx, ly) is None:
        pytest.skip("reference sources/samples not present (GPU box): oracle is pinned by tests/golden there")
    q = mp.get_context("spawn").Queue()
    p = mp.get_context("spawn").Process(target=_worker, args=(lx, ly, path, nsteps, q))
    p.start()
    bad, anomalies, npairs = q.get(timeout=600)
    p.join()
    assert bad == [], f"fields differing from the reference: {bad}"
    assert anomalies == 0 and npairs > 1000


def test_harness_init_equals_real_main(po, tmp_path):
    """oracle/ref_harness.c's own init must equal the reference's real main() (stopped by the time()
    hook after N renderScene calls)."""
    path = os.path.join(REF_BIN, "a08d83.data")
    if not os.path.exists(path) or po.build_ref(600, 500) is None:
        pytest.skip("reference not present")
    code = f"""
import sys, ctypes, numpy as np
sys.path.insert(0, {os.path.dirname(po.__file__)!r})
import pyoracle as po
L = ctypes.CDLL(po.ref_lib_path(600, 500))
mode = sys.argv[1]
if mode == "own":
    assert L.ref_init({path.encode()!r}) == 0; L.ref_steps(ctypes.c_long(25))
else:
    assert L.ref_run_real_main({path.encode()!r}, ctypes.c_long(25), {str(tmp_path).encode()!r}) == 0
f = np.zeros(600 * 500 * 9); L.ref_get_f(f.ctypes.data_as(ctypes.c_void_p))
g = np.zeros((L.ref_nbgrains(), 30)); L.ref_get_grains(g.ctypes.data_as(ctypes.c_void_p))
np.savez({str(tmp_path)!r} + "/" + mode + ".npz", f=f, g=g)
"""
    import subprocess, sys
    for mode in ("own", "real"):
        subprocess.run([sys.executable, "-c", code, mode], check=True, stdout=subprocess.DEVNULL)
    a, b = np.load(tmp_path / "own.npz"), np.load(tmp_path / "real.npz")
    assert np.array_equal(a["f"], b["f"]) and np.array_equal(a["g"], b["g"])


def test_f32_goldens_are_what_the_single_precision_reference_produces(po):
    """tests/golden/*_f32.npz pin the float build of the library (liblbmdem_hip_sp.so). They are dumps of the reference
    compiled -DSINGLE_PRECISION (typedef float real, main.c:34-40) through the same harness: regenerate one fluid case
    and the DEM case here and compare with what is stored."""
    import multiprocessing as mp
    import golden_util as gu
    if not po.reference_available():
        pytest.skip("the reference is not present on this machine")
    for name in ("G2_moving_grain_96x96", "G5_dem_64x48"):
        case = gu.ALL_CASES[name]
        q = mp.Queue()
        p = mp.Process(target=gu.mg._generate, args=(name, case, q, True))
        p.start()
        res = q.get()
        p.join()
        stored = gu.load(name + "_f32")
        fresh = gu.mg.pack(name, case, res)
        assert set(fresh) == set(stored)
        for k in fresh:
            a, b = np.asarray(fresh[k]), np.asarray(stored[k])
            assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), (name, k)
            if np.asarray(fresh[k]).dtype == np.float64 and k not in ("r_mm", "x_mm", "y_mm", "scalars"):
                v = np.asarray(fresh[k])
                assert np.array_equal(v, v.astype(np.float32).astype(np.float64), equal_nan=True), (name, k, "not floats")
