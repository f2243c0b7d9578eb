"""DEM-only sub-step rate (run_dem, list rebuilds included) with the library LBMDEM_HIP_LIBRARY points at.
    python scripts/dem_rate.py [bench|real] [tiles 0|1] [chain max]
bench = the 50 000-grain row packing, real = the reference's own bin/50000.data geometry (tests/golden fixture);
tiles: how the multi-sub-step kernel's tiles are composed (lbmdem_set_dem_tiles: 1 = patches of the packing, 0 = by index);
also prints the halo sizes (distinct partners outside a tile) the composition gives with the list as built."""
import sys, time, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import __graft_entry__ as ge, samples
pkg = ge.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "bench"
if which == "real":
    g = np.load(os.path.join(ROOT, "tests", "golden", "real_50000_4096x4096.npz")); r, x1, x2 = g["r"], g["x1"], g["x2"]
else:
    r, x, y = samples.row_packing(4096, 4096, 50000, seed=1234); r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(4096, 4096, r, x1, x2)
if len(sys.argv) > 2 and hasattr(sim._L, "lbmdem_set_dem_tiles"): sim.set_dem_tiles(int(sys.argv[2]))
if len(sys.argv) > 3: sim.set_dem_chain(int(sys.argv[3]))
sim.run_dem(200); sim.sync(); t0 = time.perf_counter(); sim.run_dem(2400); sim.sync()
rate = 2400 / (time.perf_counter() - t0)
print(which, "tiles", sys.argv[2] if len(sys.argv) > 2 else "default", round(rate), "sub-steps/s", round(1e6 / rate, 2), "us per sub-step", sim.dem_chain_stats(),
      "recoveries", sim.dem_chain_recoveries())
# halo sizes of the two compositions with the list as it stands (host arithmetic: the same curve as lbmdem_capi.hip)
def hilbert(x, y):
    x = x.copy(); y = y.copy(); d = np.zeros(len(x), np.uint64)
    s = 32768
    while s > 0:
        rx = ((x & s) > 0).astype(np.uint64); ry = ((y & s) > 0).astype(np.uint64)
        d += np.uint64(s) * np.uint64(s) * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x[flip] = 65535 - x[flip]; y[flip] = 65535 - y[flip]
        sw = ry == 0
        x[sw], y[sw] = y[sw], x[sw].copy()
        s >>= 1
    return d
cumul, nbrs, _ = sim.verlet()
n = len(r)
first = np.concatenate([[0], cumul[:-1]]); counts = np.maximum(cumul - first, 0); counts[-1] = 0
own = np.repeat(np.arange(n), counts); npairs = int(counts.sum())
pairs = np.concatenate([np.stack([own, nbrs[:npairs]], 1), np.stack([nbrs[:npairs], own], 1)])
k = sim.kinematics
span = max(np.ptp(k[:, 0]), np.ptp(k[:, 1]))
hx = ((k[:, 0] - k[:, 0].min()) * 65535.0 / span).astype(np.uint64); hy = ((k[:, 1] - k[:, 1].min()) * 65535.0 / span).astype(np.uint64)
order = np.argsort(hilbert(hx, hy), kind="stable")
for name, tile_of in (("by index", np.arange(n) // 64), ("hilbert", np.argsort(order) // 64)):
    off = pairs[tile_of[pairs[:, 0]] != tile_of[pairs[:, 1]]]
    key = tile_of[off[:, 0]].astype(np.int64) * n + off[:, 1]
    halo = np.bincount(np.unique(key) // n, minlength=(n + 63) // 64)
    ent = np.bincount(tile_of[pairs[:, 0]], minlength=(n + 63) // 64)
    print(f"  tiles {name}: halo grains per tile mean {halo.mean():.1f} max {halo.max()}, list entries per tile mean {ent.mean():.1f} max {ent.max()}")
