#!/usr/bin/env python3
"""Compile the binding INTEGRATION.md documents -- build container only.

Cuts the five `[binding:*]` code blocks out of INTEGRATION.md, splices them into a TEMPORARY copy of the reference's
src/main.c (under a mkdtemp directory that is deleted before returning: reference source is never written into the
repository), compiles it with gcc -DUSE_LBMDEM_HIP -Iinclude and links it against 2d-lbm-dem_amd/liblbmdem_hip.so:

    oracle/_ref/ref_hip_<lx>x<ly>      (git-ignored binary; the reference's own main() driving the HIP library)

Test infrastructure: only tests/ run it. usage: python oracle/make_integration_check.py [lx ly]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("LBMDEM_REFERENCE", "/root/reference")


def binding_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = {}
    for m in re.finditer(r"```c\n/\* \[binding:(\w+)\] \*/\n(.*?)```", text, re.S):
        blocks[m.group(1)] = m.group(2)
    missing = {"helpers", "step", "refresh", "attach", "final"} - set(blocks)
    if missing:
        raise SystemExit(f"INTEGRATION.md lacks the binding blocks {sorted(missing)}")
    return blocks


def patched_main_c(src, b):
    """The edit, anchored on short unique lines of the reference (checked: each anchor must occur exactly once)."""
    lines = src.split("\n")

    def find(pred, start=0, what=""):
        hits = [i for i in range(start, len(lines)) if pred(lines[i])]
        if not hits:
            raise SystemExit(f"anchor not found: {what}")
        return hits[0]

    def only(text):
        hits = [i for i, l in enumerate(lines) if l.strip() == text]
        if len(hits) != 1:
            raise SystemExit(f"anchor {text!r} occurs {len(hits)} times")
        return hits[0]

    hip = lambda code, orig: ["#ifdef USE_LBMDEM_HIP"] + code.rstrip("\n").split("\n") + ["#else"] + orig + ["#endif"]
    # work bottom-up so that earlier indices stay valid
    i_final = only("final_density();")
    lines[i_final:i_final + 1] = hip(b["final"], [lines[i_final]])
    i_while = only("} while (nbsteps * dt <= duration);")
    lines[i_while:i_while + 1] = ["#ifdef LBMDEM_MAX_STEPS", "  } while (nbsteps * dt <= duration && nbsteps < LBMDEM_MAX_STEPS);",
                                  "#else", lines[i_while], "#endif"]
    i_attach = only("init_obst();")
    lines[i_attach + 1:i_attach + 1] = ["#ifdef USE_LBMDEM_HIP"] + b["attach"].rstrip("\n").split("\n") + ["#endif"]
    i_rs = only("void renderScene(void) {")
    i_dem = find(lambda l: l.strip() == "write_DEM();", i_rs, "write_DEM();")
    lines[i_dem:i_dem] = ["#ifdef USE_LBMDEM_HIP"] + b["refresh"].rstrip("\n").split("\n") + ["#endif"]
    i_vtk = find(lambda l: l.strip() == "write_vtk(lx, ly, f, nbgrains, g);", i_rs, "write_vtk call")
    lines[i_vtk:i_vtk] = ["#ifdef USE_LBMDEM_HIP"] + b["refresh"].rstrip("\n").split("\n") + ["#endif"]
    i_step0 = find(lambda l: l.strip() == "#ifdef _FLUIDE_", i_rs, "#ifdef _FLUIDE_ in renderScene")
    i_step1 = find(lambda l: l.strip() == "nbsteps++;", i_step0, "nbsteps++")
    lines[i_step0:i_step1] = hip(b["step"], lines[i_step0:i_step1])
    lines[i_rs:i_rs] = ["#ifdef USE_LBMDEM_HIP"] + b["helpers"].rstrip("\n").split("\n") + ["#endif", ""]
    return "\n".join(lines)


def build(lx, ly, max_steps=None, out=None):
    if not os.path.isfile(os.path.join(REF, "src", "main.c")):
        raise SystemExit(f"the reference is not present at {REF}")
    lib_dir = os.path.join(ROOT, "2d-lbm-dem_amd")
    if not os.path.isfile(os.path.join(lib_dir, "liblbmdem_hip.so")):
        raise SystemExit("build 2d-lbm-dem_amd/liblbmdem_hip.so first (make -C 2d-lbm-dem_amd/csrc)")
    out = out or os.path.join(HERE, "_ref", f"ref_hip_{lx}x{ly}")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="lbmdem_integration_")
    try:
        src = open(os.path.join(REF, "src", "main.c")).read()
        with open(os.path.join(tmp, "main_hip.c"), "w") as fh:
            fh.write(patched_main_c(src, binding_blocks()))
        cmd = ["gcc", "-std=gnu99", "-O2", "-ffp-contract=off", "-w", f"-Dlx={lx}", f"-Dly={ly}", "-DUSE_LBMDEM_HIP",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "src"),
               os.path.join(tmp, "main_hip.c"), os.path.join(REF, "src", "visit_writer.c"),
               "-L" + lib_dir, "-llbmdem_hip", "-Wl,-rpath,$ORIGIN/../../2d-lbm-dem_amd", "-lm", "-o", out]
        if max_steps is not None:
            cmd.insert(1, f"-DLBMDEM_MAX_STEPS={int(max_steps)}")
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    lx, ly = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (256, 200)
    ms = int(sys.argv[3]) if len(sys.argv) >= 4 else None
    print(build(lx, ly, ms))
