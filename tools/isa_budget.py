#!/usr/bin/env python3
"""Instruction budget of ring_kernel instances, from the gfx950 assembly (no GPU needed).

    python tools/isa_budget.py --radii 38,39,46,50 [--dilate] [--np 0] ["--defs=-DSMRF_RING_JCAP(T,R)=2"] [--md out.md]

Compiles ring_kernel<float, R, DIL, 256, NP> alone with -DSMRF_ISA_MARK (csrc/morph_ring.h: comment lines that name the
phase of the batch loop the following instructions belong to), takes the batch loop (the backward branch with the longest
span), drops the blocks marked `rare` (reflected / clamped rows, NaN rule, segment ends) and the epilogue variants the
16384^2 benchmark does not take (it runs streaming stores; the erosion has no flag step), and counts what is left per
phase and per kind of instruction, divided by the 2 NP rows of a batch: instructions per 64-cell ROW, next to the
decomposition's own count - (R - 1) ring updates + K - 1 window steps (+ the extra reads' folds) + 1..2 for the two
completed rows.  NP = 0: the instance ring_launch takes on long segments (the in-place one where a radius is dual).
"""
import argparse
import math
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neilpy_amd.build import CSRC, FLAGS, hipcc  # noqa: E402

SRC = """#include "morph_ring.h"
namespace smrf {
constexpr int kNP = NPP > 0 ? NPP : (ring_tuned_inplace_dual<float>(RR) && SMRF_RING_INPLACE(float, RR)) ? SMRF_RING_INPLACE_NP(float, RR) : SMRF_RING_NP(float, RR);
void* get() { return (void*)ring_kernel<float, RR, DILL, 256, kNP>; }
}
"""

KINDS = ["minmax", "mov", "valu_other", "ds_read", "ds_write", "vmem", "salu", "wait", "barrier", "branch"]
# round 5: the scalar side by category
SCALAR = ["s_nop", "s_waitcnt", "s_setprio", "salu_addr", "salu_cond", "branch", "barrier"]


def scalar_kind(op):
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_setprio":
        return "s_setprio"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_cmp", "s_cselect", "s_and", "s_or", "s_andn2", "s_xor", "s_not", "s_mov_b64")) or "saveexec" in op:
        return "salu_cond"          # uniform tests, exec masks
    if op.startswith("s_"):
        return "salu_addr"          # row offsets, 64-bit address arithmetic, loop counters
    return None


def kind_of(op):
    if re.match(r"v_(min|max)3?_f(32|64)$", op):
        return "minmax"
    if op in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_pk_mov_b32", "v_mov_b64"):
        return "mov"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_read"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop"):
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "valu_other"


def disk(R):
    hw = [math.isqrt(R * R - d * d) for d in range(R + 1)]
    wk = sorted(set(hw))
    return hw, wk


def compile_one(R, dil, np_, defs, tmp):
    src = os.path.join(tmp, "one.hip")
    open(src, "w").write(SRC)
    out = os.path.join(tmp, "one_%d.s" % R)
    cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["-DSMRF_ISA_MARK", "-DRR=%d" % R, "-DNPP=%d" % np_,
                                                           "-DDILL=%s" % ("true" if dil else "false")] + defs + \
          ["--offload-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return open(out).read()


def analyse(text, dil):
    m = re.search(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    name, desc = m.group(1), m.group(2)
    info = {"vgpr": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)),
            "scratch": int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1)),
            "np": int(re.search(r"ELi256ELi(\d+)EEEv", name).group(1))}
    lines = text.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(body):
        mm = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = i - labels[mm.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[mm.group(1)], i)
    _, lo, hi = best
    want_epi = "epilogue:nt1:flag%d" % (1 if dil else 0)
    counts = {}
    phase = "stage"
    code_bytes = 0
    for l in body[lo:hi + 1]:
        st = l.strip()
        mm = re.match(r";\s*SMRF_MARK (\S+)", st)
        if mm:
            phase = mm.group(1)
            continue
        if not st or st[0] in ";." or st.endswith(":") or st.startswith(("#", "//")):
            continue
        op = st.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        if phase == "rare" or (phase.startswith("epilogue:") and phase != want_epi):
            continue
        ph = "epilogue" if phase.startswith("epilogue") else phase
        counts.setdefault(ph, {}).setdefault(kind_of(op), 0)
        counts[ph][kind_of(op)] += 1
        sk = scalar_kind(op)
        if sk:
            counts[ph]["sc:" + sk] = counts[ph].get("sc:" + sk, 0) + 1
    info["loop_lines"] = hi - lo
    return info, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--radii", default="38,39,46,50")
    ap.add_argument("--dilate", action="store_true")
    ap.add_argument("--np", type=int, default=0)
    ap.add_argument("--defs", default="")
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    defs = [d for d in a.defs.split() if d]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for R in [int(v) for v in a.radii.split(",")]:
            text = compile_one(R, a.dilate, a.np, defs, tmp)
            info, counts = analyse(text, a.dilate)
            rows = 2 * info["np"]
            hw, wk = disk(R)
            K = len(wk)
            tot = {k: sum(c.get(k, 0) for c in counts.values()) for k in KINDS}
            valu = tot["minmax"] + tot["mov"] + tot["valu_other"]
            out.append("### R = %d %s, NP = %d: %d VGPRs, scratch %d B, batch loop %d asm lines" %
                       (R, "dilation + flag" if a.dilate else "erosion", info["np"], info["vgpr"], info["scratch"], info["loop_lines"]))
            out.append("")
            out.append("| phase | " + " | ".join(KINDS) + " |")
            out.append("|---|" + "---|" * len(KINDS))
            for ph in sorted(counts):
                out.append("| %s | " % ph + " | ".join("%.1f" % (counts[ph].get(k, 0) / rows) for k in KINDS) + " |")
            out.append("| **all, per 64-cell row** | " + " | ".join("**%.1f**" % (tot[k] / rows) for k in KINDS) + " |")
            out.append("")
            out.append("scalar side per 64-cell row | " + " | ".join(SCALAR) + " | sum")
            out.append("---|" + "---|" * (len(SCALAR) + 1))
            for ph in sorted(counts):
                v = [counts[ph].get("sc:" + k, 0) / rows for k in SCALAR]
                if sum(v):
                    out.append("%s | " % ph + " | ".join("%.1f" % x for x in v) + " | %.1f" % sum(v))
            sv = [sum(c.get("sc:" + k, 0) for c in counts.values()) / rows for k in SCALAR]
            out.append("**all** | " + " | ".join("**%.1f**" % x for x in sv) + " | **%.1f**" % sum(sv))
            out.append("")
            out.append("VALU per row %.1f (min / max %.1f, v_mov %.1f, other %.1f) against R + K = %d (R - 1 = %d ring updates, K - 1 = %d "
                       "window steps, 2 for the completed rows): %.1f over; LDS reads per row %.1f" %
                       (valu / rows, tot["minmax"] / rows, tot["mov"] / rows, tot["valu_other"] / rows, R + K, R - 1, K - 1,
                        valu / rows - (R + K), tot["ds_read"] / rows))
            out.append("")
    txt = "\n".join(out)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
