#!/usr/bin/env python3
"""Survey the fused window steps / delayed ring slots (morph_ring.h: red_chain, RingCfg::SLOT_DELAY) over every fp32 ring
instance the library holds, from the compiled code alone (no GPU), and write csrc/ring_fuse.inc.

    python tools/ring_fuse_inc.py survey [-j 8] [--radii 15-64]      -> profiles/tuning/r05_ring_fuse_survey.json
    python tools/ring_fuse_inc.py inc [--deny 34,41] [--cap 46:1]          -> neilpy_amd/csrc/ring_fuse.inc

For a radius's shifting instance and (where ring_inpl.inc marks it dual) its in-place instance, erosion and dilation + flag,
each compiled with mode 0 (neither), 1 (fused steps) and 3 (fused steps + slots one group late): VGPRs, scratch bytes,
spilled VGPRs and the s_nop count of the kernel.  A mode is ELIGIBLE for an instance when it adds no scratch and no spill
and leaves the VGPR count within the occupancy step of mode 0 (<= 128 / 168 / 256) for erosion and dilation alike; `inc`
takes the highest eligible mode (3 over 1 over 0) unless --deny lists the radius (timings decide: tools/window_ab.py).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neilpy_amd.build import CSRC, FLAGS, hipcc  # noqa: E402

SURVEY = os.path.join(ROOT, "profiles", "tuning", "r05_ring_fuse_survey.json")
SRC = """#include "morph_ring.h"
namespace smrf {
#if KIND == 1
static_assert(ring_tuned_inplace_dual<float>(RR) && SMRF_RING_INPLACE(float, RR), "no second instance");
constexpr int kNP = SMRF_RING_INPLACE_NP(float, RR);
#else
constexpr int kNP = SMRF_RING_NP(float, RR);
#endif
constexpr bool kInpl = RingCfg<float, RR, 256, kNP>::INPLACE;
__attribute__((used)) const int smrf_is_inplace = kInpl ? 1 : 0;
void* get() { return (void*)ring_kernel<float, RR, DILL, 256, kNP>; }
}
"""


def compile_one(job):
    R, kind, dil, mode, tmp = job
    src = os.path.join(tmp, "one_%d_%d_%d_%d.hip" % (R, kind, dil, mode))
    open(src, "w").write(SRC)
    out = src.replace(".hip", ".s")
    cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["-DRR=%d" % R, "-DKIND=%d" % kind, "-DDILL=%s" % ("true" if dil else "false"),
                                                           "-DSMRF_RING_FUSE_MODE(T,R,INPL)=%d" % mode, "--offload-device-only", "-S", src,
                                                           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "no second instance" in r.stderr:
            return (R, kind, dil, mode), None
        raise RuntimeError(r.stderr[-2000:])
    text = open(out).read()
    m = re.search(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    desc = m.group(2)
    rec = {"vgpr": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)),
           "scratch": int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1)),
           "spill": int(re.search(r"\.vgpr_spill_count:\s+(\d+)", text).group(1)),
           "np": int(re.search(r"ELi256ELi(\d+)EEEv", m.group(1)).group(1)),
           "inplace": "smrf_is_inplace" in text and bool(re.search(r"smrf_is_inplace.*?\.long\s+1", text, re.S)),
           "nops": len(re.findall(r"^\s+s_nop", text, re.M)), "lines": text.count("\n")}
    os.remove(out)
    os.remove(src)
    return (R, kind, dil, mode), rec


def step_of(v):
    return 128 if v <= 128 else 168 if v <= 168 else 256 if v <= 256 else 512


def survey(a):
    lo, hi = (int(v) for v in a.radii.split("-"))
    with tempfile.TemporaryDirectory() as tmp:
        jobs = [(R, kind, dil, mode, tmp) for R in range(lo, hi + 1) for kind in (0, 1) for dil in (0, 1) for mode in (0, 1, 3)]
        with ThreadPoolExecutor(max_workers=a.j) as ex:
            res = list(ex.map(compile_one, jobs))
    out = json.load(open(SURVEY)) if os.path.exists(SURVEY) else {}      # a partial --radii run updates its radii only
    for R in range(lo, hi + 1):
        out.pop(str(R), None)
    for (R, kind, dil, mode), rec in res:
        if rec is not None:
            out.setdefault(str(R), {}).setdefault(str(kind), {}).setdefault(str(dil), {})[str(mode)] = rec
    os.makedirs(os.path.dirname(SURVEY), exist_ok=True)
    with open(SURVEY, "w") as f:                          # one line per radius
        keys = sorted(out, key=int)
        f.write("{\n" + ",\n".join(" %s: %s" % (json.dumps(k), json.dumps(out[k], sort_keys=True, separators=(",", ":"))) for k in keys)
                + "\n}\n")
    for R in sorted((r for r in out if lo <= int(r) <= hi), key=int):
        for kind in sorted(out[R]):
            d = out[R][kind]
            print("R=%s %s NP=%d: " % (R, "in-place" if d["0"]["0"]["inplace"] else "shifting", d["0"]["0"]["np"]) + "  ".join(
                "mode %s: e %d/%d/%d d %d/%d/%d nops %d" % (m, d["0"][m]["vgpr"], d["0"][m]["scratch"], d["0"][m]["spill"], d["1"][m]["vgpr"],
                                                         d["1"][m]["scratch"], d["1"][m]["spill"], d["0"][m]["nops"]) for m in ("0", "1", "3")))


def eligible(d, mode):
    for dil in ("0", "1"):
        b, v = d[dil]["0"], d[dil][mode]
        if v["scratch"] > b["scratch"] or v["spill"] > b["spill"] or step_of(v["vgpr"]) > step_of(b["vgpr"]):
            return False
    return True


def inc(a):
    sv = json.load(open(SURVEY))
    deny = {int(v) for v in a.deny.split(",") if v}
    cap = {int(k): int(v) for k, v in (kv.split(":") for kv in a.cap.split(",") if kv)}
    tabs = {False: [0] * 65, True: [0] * 65}
    for R, kinds in sv.items():
        for kind, d in kinds.items():
            inplace = d["0"]["0"]["inplace"]
            mode = 3 if eligible(d, "3") else 1 if eligible(d, "1") else 0
            if int(R) in deny:
                mode = 0
            mode = min(mode, cap.get(int(R), 3))
            tabs[inplace][int(R)] = mode

    def row(t):
        return ",\n    ".join(", ".join(str(v) for v in t[i:i + 17]) for i in range(0, 65, 17))
    txt = ("// Fused window steps / delayed ring slots per fp32 radius (morph_ring.h: red_chain, RingCfg::SLOT_DELAY), for the shifting\n"
           "// and the in-place instance of a radius: 0 = neither, 1 = fused steps, 3 = fused steps + slots one group late.\n"
           "// Written by tools/ring_fuse_inc.py inc from profiles/tuning/r05_ring_fuse_survey.json: the highest mode that adds no\n"
           "// scratch, no spill and no occupancy step to the erosion and the dilation + flag instance%s%s.\n"
           % ((", radii switched off after timing: " + a.deny) if a.deny else "", (", capped: " + a.cap) if a.cap else "") +
           "inline constexpr unsigned char kRingFuseShiftF32[65] = {\n    %s};\n" % row(tabs[False]) +
           "inline constexpr unsigned char kRingFuseInplF32[65] = {\n    %s};\n" % row(tabs[True]) +
           "template <typename T> constexpr int ring_tuned_fuse(int r, bool inplace) {\n"
           "  return sizeof(T) == 4 && r >= 15 && r <= 64 ? (inplace ? kRingFuseInplF32[r] : kRingFuseShiftF32[r]) : 0;\n}\n")
    open(os.path.join(CSRC, "ring_fuse.inc"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["survey", "inc"])
    ap.add_argument("-j", type=int, default=8)
    ap.add_argument("--radii", default="15-64")
    ap.add_argument("--deny", default="")
    ap.add_argument("--cap", default="", help="R:mode,... upper bound per radius")
    a = ap.parse_args()
    (survey if a.cmd == "survey" else inc)(a)
