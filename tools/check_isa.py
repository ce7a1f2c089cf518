#!/usr/bin/env python3
"""Developer check of the generated ring / fused / chain kernels (no GPU needed): compiles every ring translation unit and chain.hip to
gfx950 assembly and reports (a) kernels with register spills, (b) fused LDS instructions (ds_read2* / ds_write2*: half
the LDS rate of the single forms on gfx950, MI355X_MICROARCH LDS table; the kernels keep hipcc from forming them with
inline-asm reads and staging stores).  Exit code 1 if either is found.

    python tools/check_isa.py [-j 8]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neilpy_amd.build import CSRC, FLAGS, RING_PARTS, hipcc  # noqa: E402


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_DEST_FIRST = re.compile(r"^(v_(?!cmp|cmpx|readlane|readfirstlane|nop)|ds_read|ds_bpermute|ds_permute|ds_swizzle|global_load|scratch_load|buffer_load|flat_load|"
                         r"global_atomic\w+ v|v_accvgpr_read)")


def _regs(tok):
    out = []
    for m in _VREG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lds_hazards(text):
    """Reads of (or writes to) a VGPR that an earlier ds_read is still filling: the kernels issue their table reads as
    inline asm and count the s_waitcnt themselves, so neither the compiler's own bookkeeping nor the hardware protects
    a register between the read's issue and the wait that covers it - a spill, a copy or a miscounted wait in that window
    would use stale data.  Linear scan per kernel: LDS operations retire in order, s_waitcnt lgkmcnt(N) leaves the N
    youngest outstanding (scalar loads only make it stricter)."""
    bad = []
    kern = None
    pending = []                                           # outstanding LDS ops, oldest first: lists of dest VGPRs
    for ln, line in enumerate(text.splitlines(), 1):
        if line.startswith("_ZN") and line.rstrip().endswith(":"):
            kern, pending = line.split(":")[0], []
            continue
        if kern is None:
            continue
        st = line.strip()
        if not st or st[0] in ";.":
            if st.startswith(".end_amdhsa_kernel") or st.startswith(".Lfunc_end"):
                kern = None
            continue
        st = st.split(";")[0].strip()
        op = st.split()[0]
        args = st[len(op):]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", args)
            if m:
                n = int(m.group(1))
                if n < len(pending):
                    pending = pending[len(pending) - n:] if n else []
            elif "vmcnt" not in args and "expcnt" not in args:   # bare immediate form: treat as full wait
                pending = []
            continue
        if op in ("s_barrier", "s_endpgm"):
            continue
        toks = [t.strip() for t in args.split(",")]
        dest_first = bool(_DEST_FIRST.match(op)) and "store" not in op and not op.startswith("ds_write")
        srcs, dsts = [], []
        for i, t in enumerate(toks):
            (dsts if (i == 0 and dest_first) else srcs).extend(_regs(t))
        busy = {r for d in pending for r in d}
        hit = [r for r in srcs + dsts if r in busy]
        if hit:
            bad.append("LDS read still in flight into v%d when `%s` uses it, in %s (asm line %d)" % (hit[0], st, kern, ln))
        if op.startswith("ds_"):
            pending.append(dsts if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) else [])
    return bad


DEFS = []


def one(job):
    part, f64, tmp = job
    if part < 0:                                           # the chained / table-free launches (both dtypes in one unit)
        out = os.path.join(tmp, "chain.s")
        cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + DEFS + ["--offload-device-only", "-S", os.path.join(CSRC, "chain.hip"),
                                                               "-o", out]
    else:
        out = os.path.join(tmp, "ring_%d_%d.s" % (f64, part))
        cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + DEFS + ["-DPART=%d" % part, "-DSMRF_F64=%d" % f64, "--offload-device-only",
                                                               "-S", os.path.join(CSRC, "ring_part.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    text = open(out).read()
    bad = []
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        if int(m.group(3)):
            bad.append("spill %s vgprs in %s (vgpr_count %s)" % (m.group(3), m.group(1), m.group(2)))
    kern = None
    fused = {}
    for line in text.splitlines():
        if line.startswith("_ZN4smrf") and line.rstrip().endswith(":"):
            kern = line.split(":")[0]
        mm = re.match(r"\s+(ds_(?:read|write)2\w*)", line)
        if mm and kern:
            fused[(kern, mm.group(1))] = fused.get((kern, mm.group(1)), 0) + 1
    bad += ["%d x %s in %s" % (n, op, k) for (k, op), n in fused.items()]
    hz = lds_hazards(text)
    bad += hz[:5] + (["... and %d more in-flight uses" % (len(hz) - 5)] if len(hz) > 5 else [])
    return part, f64, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--defs", default="", help="extra compiler flags (variant builds), space separated")
    ap.add_argument("--f32-only", action="store_true")
    a = ap.parse_args()
    DEFS[:] = [d for d in a.defs.split() if d]
    with tempfile.TemporaryDirectory() as tmp:
        jobs = [(p, f, tmp) for f in ((0,) if a.f32_only else (0, 1)) for p in range(RING_PARTS)] + [(-1, 0, tmp)]
        with ThreadPoolExecutor(max_workers=a.j) as ex:
            res = list(ex.map(one, jobs))
    n = 0
    for part, f64, bad in res:
        for b in bad:
            n += 1
            print("%s: %s" % ("chain" if part < 0 else "ring_%s_p%d" % ("f64" if f64 else "f32", part), b))
    print("%d finding(s) in %d translation units" % (n, len(res)))
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
