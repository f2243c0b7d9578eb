"""Deterministic synthetic grain packings shared by the tests and bench.py (inputs only).

Units: millimetres, as in the reference's .data files (main.c:612-628: every value x 1e-3 -> m).
The lattice spacing is ~0.1 mm (dx = 1e-4 * lx / (lx - 1) m), so a lattice of lx x ly nodes spans
0.1*lx x 0.1*ly mm.
"""
import numpy as np


def splitmix64(seed):
    """splitmix64 stream -> floats in [0, 1)."""
    state = np.uint64(seed)
    mask = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        while True:
            state = (state + np.uint64(0x9E3779B97F4A7C15)) & mask
            z = state
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & mask
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & mask
            z = z ^ (z >> np.uint64(31))
            yield float(z >> np.uint64(11)) / float(1 << 53)


def row_packing(lx, ly, n_target, seed=1234, rmin=0.5, rmax=0.9, touch_prob=0.5, max_overlap=4e-3,
                margin=0.3):
    """Rows of discs, radii ~U[rmin, rmax] mm, laid left to right; neighbours in a row either touch
    with a tiny overlap (<= max_overlap mm, typical of a packing under its own weight) or leave a
    gap <= 0.1 mm. Rows are 2*rmax + 0.02 apart so different rows never overlap. Fills from the
    bottom until n_target grains are placed. Returns (r, x, y) in mm."""
    g = splitmix64(seed)
    W, H = 0.1 * lx, 0.1 * ly
    rs, xs, ys = [], [], []
    row_h = 2 * rmax + 0.02
    y = margin + rmax
    while len(rs) < n_target and y + rmax + margin < H:
        r_prev = None
        x = margin
        while len(rs) < n_target:
            r = rmin + (rmax - rmin) * next(g)
            if r_prev is None:
                xc = x + r
            else:
                gap = -max_overlap * next(g) if next(g) < touch_prob else 0.1 * next(g)
                xc = x + r_prev + r + gap
            if xc + r + margin > W:
                break
            rs.append(r); xs.append(xc); ys.append(y + 0.01 * (next(g) - 0.5))
            x, r_prev = xc, r
        y += row_h
    return np.array(rs), np.array(xs), np.array(ys)


def to_metres(r, x, y):
    return r * 1e-3, x * 1e-3, y * 1e-3
