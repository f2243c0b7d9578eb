#!/usr/bin/env python3
"""isa_census.py -- per-phase census of the fused fluid kernel's main loop, from the shipped ISA.

    python scripts/isa_census.py [--kernel SUBSTR] [--freq profiles/r06_fused_block_freq.json] [-o OUT]

Compiles 2d-lbm-dem_amd/csrc/lbm_fused.hip for gfx950 with the product's flags plus -gline-tables-only (line tables do not
change the generated instructions: the script checks that the instruction count equals the build without them), takes the
main loop of k_cs_march<0,2,60,true> (the two unrolled rows), and attributes every instruction to
  * a PHASE, from the source line the compiler ascribes it to (the innermost inlined callee: mrt_collide, node_active, ...),
  * a CLASS, from its opcode (fp64 arithmetic, division/sqrt sequences, compares/selects, integer/address VALU, moves,
    lane spills, DPP, scalar ALU, s_nop, waits, LDS, vector memory, branches),
  * a REGION: a stretch of the loop between two control-flow instructions, with the condition that guards it.
With --freq (a JSON {region_kind: executions per wave-row} produced by scripts/fused_region_freq.py from the bench packing's
obstacle map) the static counts are weighted into dynamic instructions per wave-row and compared with SQ_INSTS_VALU.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "2d-lbm-dem_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "--cuda-device-only", "-S"]


def compile_asm(src, out, extra):
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + [src, "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)


INSTR = re.compile(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$")


def is_instr(line):
    m = INSTR.match(line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        return op, m.group(2)
    return None


def classify(op, args):
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "lane_spill"
    if op == "v_readfirstlane_b32":
        return "readfirstlane"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vmem_store"
    if "dpp" in args or "_dpp" in op:
        return "dpp"
    if op.startswith(("v_div_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_ldexp_", "v_frexp_", "v_trig")):
        return "div_sqrt"
    if op.startswith("v_fma_f64"):
        return "div_sqrt"     # -ffp-contract=off: every fp64 fma belongs to a division / square-root sequence
    if op.startswith(("v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64", "v_fma_f32", "v_add_f32", "v_mul_f32")):
        return "fp_arith"
    if op.startswith(("v_cmp", "v_cndmask", "v_cmpx")):
        return "cmp_select"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "v_mov"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_"):
        return "int_valu"
    return "other"


VALU = {"lane_spill", "readfirstlane", "dpp", "div_sqrt", "fp_arith", "cmp_select", "v_mov", "cvt", "int_valu"}

# phases from (file, line): line ranges of the CURRENT sources, found by looking for the marker functions
def phase_table():
    tab = []   # (file, first, last, phase)

    def span(fname, start_pat, phase, end_pat=r"^}"):
        path = os.path.join(CSRC, fname)
        lines = open(path).read().split("\n")
        for i, l in enumerate(lines):
            if re.search(start_pat, l):
                j = i
                while j < len(lines) and not re.match(end_pat, lines[j]):
                    j += 1
                tab.append((fname, i + 1, j + 1, phase))
                return
        raise SystemExit(f"phase marker {start_pat!r} not found in {fname}")

    span("lbm_device.h", r"void mrt_collide\(", "collide (mrt_collide)")
    span("lbm_device.h", r"void grain_equilibrium_u\(", "reinit (grain_equilibrium)")
    span("lbm_device.h", r"void grain_equilibrium\(", "reinit (grain_equilibrium)")
    span("lbm_device.h", r"real wall_ux\(", "wall velocity (reinit / bounce-back)")
    span("lbm_device.h", r"real wall_uy\(", "wall velocity (reinit / bounce-back)")
    span("lbm_device.h", r"real link_delta_rt\(", "bounce-back evaluation (ibb_eval_rt)")
    span("lbm_device.h", r"real ibb_far_rt\(", "bounce-back evaluation (ibb_eval_rt)")
    span("lbm_device.h", r"real ibb_eval_rt\(", "bounce-back evaluation (ibb_eval_rt)")
    span("lbm_device.h", r"int slot_line\(", "link-sum store (slot_line + table address)")
    span("lbm_device.h", r"GP load_gp\(", "grain-record gather (load_gp)")
    span("lbm_device.h", r"long fbase_xy\(", "population address (fbase_xy)")
    span("lbm_device.h", r"unsigned mbcnt\(", "bounce-back compaction (mbcnt, LDS slots)")
    span("lbm_device.h", r"bool pull_classify\(", "classification, edge rows (pull_classify)")
    span("lbm_march.h", r"real buf_load_real\(", "row loads (load_raw)")
    span("lbm_march.h", r"void buf_store_real\(", "the row's nine stores (results of the links merged in)")
    span("lbm_march.h", r"int fcol_bytes\(", "population address (fbase_xy)")
    span("lbm_device.h", r"GPv load_gpv\(", "grain-record gather (load_gp)")
    span("lbm_device.h", r"real link_delta\(", "bounce-back evaluation (ibb_eval_rt)")
    span("lbm_march.h", r"Ids3 load_ids\(", "obstacle ids (load_ids)")
    span("lbm_march.h", r"bool node_active\(", "act (node_active)")
    span("lbm_march.h", r"struct RecRing", "record ring (LDS put/get)", r"^};")
    span("lbm_march.h", r"void classify_store_row\(", "classification, edge rows (pull_classify)")
    span("lbm_march.h", r"bool lane_of\(", "lane masks (lbm_march.h helpers)", r"^#define MARCH_ROW")
    span("lbm_march.h", r"lmask m_enclosed\(", "act (node_active)")
    span("lbm_march.h", r"int dpp_up1\(int", "DPP shifts", r"^__device__ __forceinline__ int shfl_dn1")
    return tab


def fused_phase(line_no, marks):
    """phase of a line of lbm_fused.hip inside k_cs_march (by the comment markers of iterate())"""
    ph = "prologue / other"
    for first, name in marks:
        if line_no >= first:
            ph = name
    return ph


def fused_marks():
    lines = open(os.path.join(CSRC, "lbm_fused.hip")).read().split("\n")
    marks = []

    def at(pat, name):
        for i, l in enumerate(lines):
            if re.search(pat, l):
                marks.append((i + 1, name))
                return
        raise SystemExit(f"marker {pat!r} not found in lbm_fused.hip")

    at(r"void k_cs_march\(", "prologue / work-item decode")
    at(r"auto load_old = ", "previous-map id (load_old, change bits)")
    at(r"auto load_raw = ", "row loads (load_raw)")
    at(r"auto interior = ", "interior test / make_fstar glue")
    at(r"auto grain_rec = ", "grain-record gather (load_gp)")
    at(r"real Fm\[9\]", "prologue / other")
    at(r"auto iterate = ", "loop glue: row rotation, buffers")
    at(r"---- \(1\) small gathers", "gathers issue (ids, previous id, records)")
    at(r"---- \(2\) the big loads", "row loads (load_raw)")
    at(r"int y_act = y;", "act (node_active)")
    at(r"real Fo\[9\], In\[9\];", "pull context: DPP shifts")
    at(r"const int gx = L.gx0 \+ x;", "classification + stores, deep rows (lane masks)")
    at(r"// rows and windows next to a lattice edge: the general", "classification, edge rows (pull_classify)")
    at(r"auto store_row = ", "the row's nine stores (results of the links merged in)")
    at(r"for \(int base = 0; base < T; base \+= LINK_SLOTS\)", "bounce-back compaction (mbcnt, LDS slots)")
    at(r"if \(base \+ lane < T\)", "bounce-back evaluation (ibb_eval_rt)")
    at(r"if \(S.tab != nullptr\)", "link-sum store (slot_line + table address)")
    at(r"if \(merged\) \{", "the row's nine stores (results of the links merged in)")
    at(r"ring.put\(x \+ 3, lane, rec_next", "record ring (LDS put/get)")
    at(r"// rotate", "loop glue: row rotation, buffers")
    at(r"for \(int x = xs; x < xe; x \+= 2\)", "loop glue: row rotation, buffers")
    return marks


def parse(asm_path, kernel_substr):
    """returns the list of (label_or_None, op, args, file, line, comment) of the kernel"""
    files = {}
    out = []
    inside = False
    cur = (None, 0)
    for raw in open(asm_path):
        line = raw.rstrip("\n")
        m = re.match(r"^\s+\.file\s+(\d+)\s+(?:\"([^\"]*)\"\s+)?\"([^\"]*)\"", line)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3))
            continue
        if re.match(r"^_Z\w+:", line):
            inside = kernel_substr in line
            continue
        if not inside:
            continue
        if line.strip().startswith(".loc"):
            p = line.split()
            cur = (files.get(int(p[1]), "?"), int(p[2]))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        if m:
            out.append(("label", m.group(1), m.group(2), None, None))
            continue
        if line.strip().startswith("s_endpgm"):
            out.append(("instr", "s_endpgm", "", cur[0], cur[1]))
            continue
        if line.strip().startswith(".Lfunc_end"):
            inside = False
            continue
        ins = is_instr(line)
        if ins:
            out.append(("instr", ins[0], ins[1], cur[0], cur[1]))
    return out


def main_loop(items, which):
    """[first, last] indices of one of the kernel's two row loops (depth-1 loops of more than 1000 instructions): the
    kernel holds two instantiations of the work item -- `interior` (no lattice-edge logic: 94 % of a 4096^2 lattice's items)
    is the shorter one, `edge` the general one"""
    loops = []
    hdr = None
    for i, it in enumerate(items):
        if it[0] == "label":
            c = it[2]
            if "Loop Header: Depth=1" in c:
                hdr = it[1]
                loops.append([hdr, i, i])
            elif hdr and ("Header=" + hdr[2:] + " ") in c + " ":
                loops[-1][2] = i
    out = []
    for name, a, b in loops:
        j = b + 1
        while j < len(items) and items[j][0] != "label":
            j += 1
        n = sum(1 for it in items[a:j] if it[0] == "instr")
        if n > 1000:
            out.append((n, a, j - 1))
    out.sort()
    if not out:
        raise SystemExit("no row loop found")
    n, a, b = out[0] if which == "interior" else out[-1]
    return a, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_cs_marchILi0ELi2ELi60ELb1E")
    ap.add_argument("--freq", default=None)
    ap.add_argument("--asm", default=None, help="use this assembly (with .loc) instead of compiling")
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--extra", default="", help="extra compiler flags")
    ap.add_argument("--loop", choices=["interior", "edge"], default="interior")
    a = ap.parse_args()
    extra = a.extra.split()
    if a.asm:
        asm_g = a.asm
    else:
        asm_g, asm_p = "/tmp/isa_census_g.s", "/tmp/isa_census_p.s"
        compile_asm("lbm_fused.hip", asm_g, ["-gline-tables-only"] + extra)
        compile_asm("lbm_fused.hip", asm_p, extra)
        n_g = sum(1 for it in parse(asm_g, a.kernel) if it[0] == "instr")
        n_p = sum(1 for it in parse(asm_p, a.kernel) if it[0] == "instr")
        if n_g != n_p:
            raise SystemExit(f"line tables changed the code: {n_g} vs {n_p} instructions")
    items = parse(asm_g, a.kernel)
    lo, hi = main_loop(items, a.loop)
    tab = phase_table()
    marks = fused_marks()

    def phase_of(f, l):
        if f == "lbm_fused.hip":
            return fused_phase(l, marks)
        for (fn, a0, a1, ph) in tab:
            if f == fn and a0 <= l <= a1:
                return ph
        if f in ("amd_device_functions.h", "amd_warp_functions.h", "__clang_hip_math.h", "amd_hip_runtime.h"):
            return f"runtime header ({f})"
        return f"other ({f}:{l})"

    # regions: maximal stretches without a label or branch; depth of exec nesting is not reconstructed, the guard is named by
    # the first source line of the stretch
    rows = []
    total = collections.Counter()
    by_phase = collections.defaultdict(collections.Counter)
    region = 0
    regions = []
    cur = collections.Counter()
    cur_first = None
    for it in items[lo:hi + 1]:
        if it[0] == "label":
            if sum(cur.values()):
                regions.append((region, cur_first, cur))
            region += 1
            cur = collections.Counter(); cur_first = None
            continue
        _, op, args, f, l = it
        c = classify(op, args)
        ph = phase_of(f, l)
        total[c] += 1
        by_phase[ph][c] += 1
        cur[c] += 1
        if cur_first is None:
            cur_first = (f, l, ph)
        if c == "branch":
            regions.append((region, cur_first, cur))
            region += 1
            cur = collections.Counter(); cur_first = None
    if sum(cur.values()):
        regions.append((region, cur_first, cur))

    classes = ["fp_arith", "div_sqrt", "cmp_select", "int_valu", "cvt", "v_mov", "dpp", "lane_spill", "readfirstlane",
               "salu", "s_nop", "s_waitcnt", "branch", "lds", "vmem_load", "vmem_store", "other"]
    lines = []
    n_all = sum(total.values())
    lines.append(f"kernel {a.kernel}, {a.loop} instantiation of the work item: row loop = {n_all} instructions for TWO rows (static, every conditional path counted once)")
    lines.append("")
    lines.append("class totals: " + ", ".join(f"{c} {total[c]} ({100.0 * total[c] / n_all:.1f} %)" for c in classes if total[c]))
    valu = sum(total[c] for c in VALU)
    lines.append(f"VALU issue slots (incl. lane spills, DPP, moves): {valu} ({100.0 * valu / n_all:.1f} %); "
                 f"fp64 arithmetic + division/sqrt sequences: {total['fp_arith'] + total['div_sqrt']} "
                 f"({100.0 * (total['fp_arith'] + total['div_sqrt']) / n_all:.1f} %)")
    lines.append("")
    hdr = f"{'phase':58s} {'all':>5s} {'VALU':>5s} " + " ".join(f"{c[:9]:>9s}" for c in classes)
    lines.append(hdr)
    for ph, cnt in sorted(by_phase.items(), key=lambda kv: -sum(kv[1].values())):
        v = sum(cnt[c] for c in VALU)
        lines.append(f"{ph[:58]:58s} {sum(cnt.values()):5d} {v:5d} " + " ".join(f"{cnt[c]:9d}" for c in classes))
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
