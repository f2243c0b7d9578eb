#!/usr/bin/env python3
"""plot.py -- the reference's plot.py (plot.py:1-24) without python-vtk and without matplotlib.

The reference script reads a legacy-VTK RECTILINEAR_GRID file with python-vtk and draws
`pcolormesh(where(grain_pressure >= 0, grain_pressure, fluid_pressure))` with a colour bar into plot.png. Here the
binary files write_vtk produces (main.c:237-338 -> visit_writer.c: big-endian float32, one variable per file) are
parsed with numpy, and the picture is encoded as a PNG directly (zlib + struct): same data, same selection rule, a
viridis-like colour map, y upwards, a colour bar with the value range on the right.

    python plot.py grain_pressure_000000.vtk [fluid_pressure_000000.vtk] [-o plot.png]

With one argument the sibling file (grain_pressure_* <-> fluid_pressure_*) is looked up next to it.
"""
import os
import struct
import sys
import zlib

import numpy as np


def read_vtk(path):
    """-> (name, array[ny][nx] or [ny][nx][3]) of a binary RECTILINEAR_GRID file with one point-data variable."""
    blob = open(path, "rb").read()
    pos = 0

    def line():
        nonlocal pos
        end = blob.index(b"\n", pos)
        s = blob[pos:end].decode("ascii", "replace")
        pos = end + 1
        return s

    if not line().startswith("# vtk DataFile"):
        raise ValueError(f"{path}: not a legacy VTK file")
    line()                                    # title
    if line().strip() != "BINARY":
        raise ValueError(f"{path}: only the binary flavour the code writes is supported")
    if line().split()[1] != "RECTILINEAR_GRID":
        raise ValueError(f"{path}: not a RECTILINEAR_GRID")
    dims = [int(v) for v in line().split()[1:4]]
    for _ in range(3):                        # X/Y/Z_COORDINATES n float + n big-endian floats (no separator)
        n = int(line().split()[1])
        pos += 4 * n
    nx, ny = dims[0], dims[1]
    name, comps = None, 1
    while pos < len(blob):
        words = line().split()
        if not words:
            continue
        if words[0] == "SCALARS":
            name, comps = words[1], 1
            line()                            # LOOKUP_TABLE default
            break
        if words[0] == "VECTORS":
            name, comps = words[1], 3
            break
    if name is None:
        raise ValueError(f"{path}: no point data")
    data = np.frombuffer(blob, dtype=">f4", count=nx * ny * comps, offset=pos).astype(np.float32)
    return name, data.reshape((ny, nx) if comps == 1 else (ny, nx, 3))


def _colormap(t):
    """viridis-like: piecewise-linear through five anchor colours; t in [0, 1] -> uint8 RGB."""
    anchors = np.array([[68, 1, 84], [59, 82, 139], [33, 145, 140], [94, 201, 98], [253, 231, 37]], float)
    x = np.clip(t, 0.0, 1.0) * (len(anchors) - 1)
    i = np.minimum(x.astype(int), len(anchors) - 2)
    w = (x - i)[..., None]
    return (anchors[i] * (1 - w) + anchors[i + 1] * w + 0.5).astype(np.uint8)


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[r].tobytes() for r in range(h))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as fp:
        fp.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def picture(grain_pressure, fluid_pressure):
    """The reference's selection (plot.py:22) and a colour bar; returns (rgb image, vmin, vmax)."""
    field = np.where(grain_pressure >= 0, grain_pressure, fluid_pressure)
    vmin, vmax = float(np.nanmin(field)), float(np.nanmax(field))
    span = vmax - vmin if vmax > vmin else 1.0
    img = _colormap((field[::-1] - vmin) / span)          # y upwards, as pcolormesh draws it
    h = img.shape[0]
    bar = _colormap(np.linspace(1, 0, h)[:, None].repeat(max(8, img.shape[1] // 40), axis=1))
    gap = np.full((h, max(4, img.shape[1] // 80), 3), 255, np.uint8)
    return np.concatenate([img, gap, bar], axis=1), vmin, vmax


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("-")]
    out = "plot.png"
    if "-o" in argv:
        out = argv[argv.index("-o") + 1]
        args = [a for a in args if a != out]
    if not args:
        raise SystemExit(__doc__)
    files = {}
    for p in args:
        name, arr = read_vtk(p)
        files[name] = arr
    if len(args) == 1:                                    # look for the sibling next to it
        base = os.path.basename(args[0])
        for mine, other in (("grain_pressure", "fluid_pressure"), ("fluid_pressure", "grain_pressure")):
            if base.startswith(mine):
                sib = os.path.join(os.path.dirname(args[0]), other + base[len(mine):])
                if os.path.exists(sib):
                    name, arr = read_vtk(sib)
                    files[name] = arr
    if "grain_pressure" not in files or "fluid_pressure" not in files:
        raise SystemExit("need grain_pressure_*.vtk and fluid_pressure_*.vtk (write_vtk writes one variable per file)")
    rgb, vmin, vmax = picture(files["grain_pressure"], files["fluid_pressure"])
    write_png(out, rgb)
    print(f"{out}: {rgb.shape[1]} x {rgb.shape[0]} px, colour range {vmin:g} .. {vmax:g}")


if __name__ == "__main__":
    main(sys.argv)
