"""Where a tile of k_dem_chain spends its time (experiment build with the clock marks compiled in:
make -C 2d-lbm-dem_amd/csrc AB=1 ABTAG=_ct ABFLAGS=-DLBMDEM_CHAIN_TIMING, run with LBMDEM_HIP_LIBRARY=2d-lbm-dem_amd/liblbmdem_hip_ab_ct.so;
the marks cost ~0.5 us per sub-step): per tile clocks waiting for the halo grains' state, clocks in the
contact phases, poll rounds, placement. DEM only (run_dem), the bench packing or a smaller one (argv[1] = grains)."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge, samples
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 96
if len(sys.argv) > 3 and sys.argv[3] == "real":   # the reference's own bin/50000.data geometry (fixture)
    gfix = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_50000_4096x4096.npz"))
    r, x1, x2 = gfix["r"], gfix["x1"], gfix["x2"]
else:
    r, x, y = samples.row_packing(4096, 4096, n, seed=1234)
    r, x1, x2 = samples.to_metres(r, x, y)
sim = pkg.LbmDem(4096, 4096, r, x1, x2)
L = sim._L
L.lbmdem_debug_chain_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
tiles = (len(r) + 63) // 64
buf = np.zeros((tiles, 16), dtype=np.int64)
L.lbmdem_debug_chain_times(sim._h, buf.ctypes.data, tiles)      # first call: allocates
sim.run_dem(100)                 # rebuild at 0, census, warm
sim.sync()
t0 = time.perf_counter(); sim.run_dem(nsub); sim.sync(); t1 = time.perf_counter()
got = L.lbmdem_debug_chain_times(sim._h, buf.ctypes.data, tiles)
print(f"{len(r)} grains, {tiles} tiles, last launch of {sim.dem_chain_stats()} ; wall {1e6*(t1-t0)/nsub:.2f} us per sub-step")
wait, work, spins = buf[:, 0] * 0.01, buf[:, 1] * 0.01, buf[:, 2]
placed, far = buf[:, 3] & 1, (buf[:, 3] >> 8) & 0xFFFFFF
xcc = (buf[:, 3] >> 32) & 0xF
hwid = (buf[:, 3] >> 36) & 0xFFFFFFFF
print('raw XCC_ID of tiles 0..15:', [hex(int(v)) for v in xcc[:16]], 'tiles per xcd', (tiles + 7) // 8)
span = (buf[:, 7].max() - buf[:, 6].min()) * 0.01
k = nsub if nsub <= 100 else None
print(f"launch span {span:.1f} us; per tile (us over the launch): wait mean {wait.mean():.1f} max {wait.max():.1f}; work mean {work.mean():.1f} "
      f"max {work.max():.1f}; poll rounds mean {spins.mean():.0f} max {spins.max()}; placed {placed.sum()}/{tiles}; "
      f"far halo grains mean {far.mean():.1f} max {far.max()}; halo mean {buf[:,4].mean():.0f} max {buf[:,4].max()}; entries mean {buf[:,5].mean():.0f} max {buf[:,5].max()}")
start = (buf[:, 6] - buf[:, 6].min()) * 0.01
print(f"tile start skew: mean {start.mean():.1f} max {start.max():.1f} us")

# HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13 (the XCD comes from XCC_ID)
cu = (xcc.astype(np.int64) << 16) | ((hwid >> 8) & 0xFF)
simd = (hwid >> 4) & 3
uniq, inv, counts = np.unique(cu, return_inverse=True, return_counts=True)
per_cu = counts[inv]
print("tiles per CU histogram:", dict(zip(*np.unique(counts, return_counts=True))), "; first wavefront's SIMD histogram:", dict(zip(*np.unique(simd, return_counts=True))))
for c in sorted(set(per_cu)):
    m = per_cu == c
    print(f"  tiles sharing a CU with {c - 1} others: {m.sum():4d} tiles, work mean {work[m].mean() / nsub:.2f} max {work[m].max() / nsub:.2f} us per sub-step, wait mean {wait[m].mean() / nsub:.2f}")
for lo, hi in ((0, 1), (1, 20), (20, 1000)):
    m = (far >= lo) & (far < hi)
    if m.any():
        print(f"  tiles with {lo}..{hi - 1} far halo grains: {m.sum():4d}, work mean {work[m].mean() / nsub:.2f}, wait mean {wait[m].mean() / nsub:.2f}, poll rounds per sub-step {spins[m].mean() / nsub:.2f}")
# same SIMD for the first wavefronts of the tiles of one CU?
key = cu * 4 + simd
_, kc = np.unique(key, return_counts=True)
print("first wavefronts sharing a SIMD (count of (CU, SIMD) groups by size):", dict(zip(*np.unique(kc, return_counts=True))))

names = ["drift+publish", "barrier A", "halo fetch", "barrier B", "phase 1", "barrier", "phase 2+barrier", "walls/acc/kick"]
print("first wavefront's lane 0, us per sub-step (mean over tiles / max): " + "; ".join(f"{nm} {buf[:, 8 + k].mean() * 0.01 / nsub:.2f}/{buf[:, 8 + k].max() * 0.01 / nsub:.2f}" for k, nm in enumerate(names)))
own = (buf[:, 8] + buf[:, 11] + buf[:, 12] + buf[:, 13] + buf[:, 14] + buf[:, 15]) * 0.01 / nsub   # all but the waits (barrier A = hint wait, halo fetch)
waits = (buf[:, 9] + buf[:, 10]) * 0.01 / nsub
print(f"per tile, us per sub-step: own time mean {own.mean():.2f} p90 {np.percentile(own, 90):.2f} p99 {np.percentile(own, 99):.2f} max {own.max():.2f}; waits mean {waits.mean():.2f} min {waits.min():.2f}; pace {(own + waits).mean():.2f}")
top = np.argsort(-own)[:12]
for t in top:
    print(f"   tile {t:4d}: own {own[t]:.2f} waits {waits[t]:.2f}; tiles on its CU {per_cu[t]}; far {far[t]}; entries {buf[t,5]}; halo {buf[t,4]}; phases " +
          " ".join(f"{buf[t, 8 + k] * 0.01 / nsub:.2f}" for k in range(8)))
