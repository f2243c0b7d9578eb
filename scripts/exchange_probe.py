"""What one exchange on the step's critical path costs: lbmdem_comm_exchange_probe of the experiment build
(`make -C 2d-lbm-dem_amd/csrc AB=1`; the helper is not part of the product ABI). Run with
LBMDEM_HIP_LIBRARY=2d-lbm-dem_amd/liblbmdem_hip_ab.so."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())


def exchange_probe(pkg, comm, doubles, iters=200):
    """(us with the exchange on a side stream, us without it, us with it in line on the main stream): one critical-path
    exchange of `doubles` values from this rank to itself."""
    L = pkg.load_library()
    fn = getattr(L, "lbmdem_comm_exchange_probe", None)
    if fn is None:
        raise RuntimeError("lbmdem_comm_exchange_probe is only in the AB build: set LBMDEM_HIP_LIBRARY to liblbmdem_hip_ab.so")
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    fn.restype = C.c_int
    out = (C.c_double * 3)()
    rc = fn(comm._c, int(doubles), int(iters), out)
    if rc != 0:
        raise RuntimeError(L.lbmdem_last_error().decode())
    return float(out[0]), float(out[1]), float(out[2])


if __name__ == "__main__":
    import __graft_entry__ as ge
    pkg = ge.load_package()
    c = pkg.Comm(pkg.comm_unique_id(), 0, 1, 0)
    for nd in (18960, 99226, 589824):
        print(nd, [round(v, 1) for v in exchange_probe(pkg, c, nd, 300)])
