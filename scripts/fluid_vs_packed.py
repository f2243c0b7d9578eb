"""Fused-kernel time per node update where the lattice is pure fluid and where it is packed with grains: the same
4096^2 lattice with 1 grain, with the bench packing (50 000 grains, lower 57 % of the lattice), and with the packing
stretched over the whole lattice height (same grain count: the packed fraction of waves goes from 57 % to 100 %)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import __graft_entry__ as ge, samples
pkg = ge.load_package()
lx = ly = 4096
out = {}
def run(name, r, x1, x2):
    sim = pkg.LbmDem(lx, ly, r, x1, x2)
    for _ in range(5): sim.lbm_step()
    sim.sync(); sim.profile_enable(True)
    for _ in range(40): sim.lbm_step()
    sim.sync()
    ms, n = sim.profile_read()          # mean duration of the fused kernel over n launches (HIP events)
    out[name] = {"grains": len(r), "launches": n, "fused_kernel_ms": round(ms, 4), "ps_per_node": round(1e9 * ms / (lx * ly), 2),
                 "roofline_frac": round(148 * lx * ly / (ms * 1e-3) / 8e12, 3)}
    del sim
run("one grain (pure fluid)", np.array([0.8e-3]), np.array([0.2]), np.array([0.2]))
r, x, y = samples.row_packing(lx, ly, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
run("bench packing (lower part of the lattice)", r, x1, x2)
frac = float((x2.max() + r.max()) / (ly * 1e-4))
out["packed_fraction_of_height"] = round(frac, 3)
print(json.dumps(out))
