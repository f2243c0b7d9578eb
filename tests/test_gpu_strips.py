"""GPU: the strip path of the HIP library (owned-row kernels, halo pack/unpack, ownership masks,
bit-exact force combine) with 2 and 3 strips living on ONE GPU, stepped in lock-step by the same
StripRunner phases that run one-per-GPU under torch.distributed. Must equal the single-domain HIP
run and the CPU oracle bit for bit -- results do not depend on the number of strips."""
import numpy as np
import pytest

import samples
from strip_backends import LoopbackComm, lockstep_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_strips_on_one_gpu_bit_exact(pkg, po, world):
    import torch
    strips = pkg.strips_module()
    lx, ly = 256, 128
    r, x, y = samples.row_packing(lx, ly, 150, seed=13)
    r, x1, x2 = samples.to_metres(r, x, y)
    cfg = pkg.derive(lx, ly, r)
    halo = strips.halo_rows(float(r.max()), cfg.dx)
    rng = np.random.default_rng(2)
    k = np.zeros((len(r), 9)); k[:, 0], k[:, 1] = x1, x2
    k[:, 3:6] = rng.normal(0, 1, (len(r), 3)) * [0.03, 0.03, 20.0]
    runners = []
    for rank, strip in enumerate(strips.partition(lx, world)):
        be = strips.GpuStripBackend(pkg, torch, lx, ly, r, x1, x2, strip, halo, 0)
        be.sim.kinematics = k
        runners.append(strips.StripRunner(be, LoopbackComm(), rank, world))
    nsteps = 4 * cfg.npDEM + 3
    lockstep_render(runners, nsteps)
    got = np.full((lx, ly, 9), np.nan)
    for R in runners:
        R.b.sim.download_f_into(got)
    ora = po.Oracle(lx, ly, r, x1, x2); ora.set_kinematics(k); ora.steps(nsteps)
    assert np.array_equal(got, ora.get_f())
    for R in runners:
        assert np.array_equal(R.b.sim.kinematics, ora.get_grains()[:, :9])
        assert np.array_equal(R.b.sim.fhf, ora.get_fhf())
    # mass over the owned rows adds up to the whole
    tot = sum(R.b.sim.final_density() for R in runners)
    # different summation trees (the reference's is one serial chain over 9*lx*ly values)
    assert abs(tot - ora.total_density()) <= 1e-10 * ora.total_density()


def test_strip_needs_enough_halo(pkg):
    with pytest.raises(pkg.LbmDemError):
        pkg.LbmDem(128, 64, [0.8e-3], [3e-3], [3e-3], strip=(0, 64), halo=3)
