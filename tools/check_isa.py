#!/usr/bin/env python3
"""Developer check of the generated ring / fused kernels (no GPU needed): compiles every ring translation unit to
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


def one(job):
    part, f64, tmp = job
    out = os.path.join(tmp, "ring_%d_%d.s" % (f64, part))
    cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["-DPART=%d" % part, "-DSMRF_F64=%d" % f64, "--offload-device-only",
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
    return part, f64, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        jobs = [(p, f, tmp) for f in (0, 1) for p in range(RING_PARTS)]
        with ThreadPoolExecutor(max_workers=a.j) as ex:
            res = list(ex.map(one, jobs))
    n = 0
    for part, f64, bad in res:
        for b in bad:
            n += 1
            print("ring_%s_p%d: %s" % ("f64" if f64 else "f32", part, b))
    print("%d finding(s) in %d translation units" % (n, len(res)))
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
