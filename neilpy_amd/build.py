"""Build libsmrf_hip.so (gfx950) in-tree with hipcc.  ``python -m neilpy_amd.build [-j N]``.

The library cross-compiles without a GPU.  Objects go to ``neilpy_amd/_build/`` and the
shared library to ``neilpy_amd/_lib/libsmrf_hip.so`` (both git-ignored; the .so travels to the
GPU box with the repository snapshot).  A source newer than its object triggers a rebuild.
"""
import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIBDIR = os.path.join(PKG, "_lib")
LIB = os.path.join(LIBDIR, "libsmrf_hip.so")
RING_PARTS = 8

FLAGS = ["-std=c++20", "-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsmrf_hip cannot be built (there is no CPU fallback)")
    return exe


def units():
    """(object name, source, extra flags)"""
    out = [(n + ".o", os.path.join(CSRC, n + ".hip"), []) for n in ("core", "morph", "chain", "grid", "springs", "fda", "tail", "spline")]
    for f64 in (0, 1):
        for p in range(RING_PARTS):
            out.append(("ring_%s_p%d.o" % ("f64" if f64 else "f32", p), os.path.join(CSRC, "ring_part.hip"),
                        ["-DPART=%d" % p, "-DSMRF_F64=%d" % f64]))
    return out


def newest_header():
    t = os.path.getmtime(os.path.join(ROOT, "include", "smrf_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".inc")):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return t


def build(jobs=None, force=False, verbose=True, variant=None, defs=()):
    """``variant`` builds a side library ``_lib/variants/<variant>.so`` with extra ``defs`` (developer
    A/B and diagnostic builds; never loaded by the package)."""
    global OBJ, LIB
    if variant:
        OBJ = os.path.join(PKG, "_build", "variant_" + variant)
        LIB = os.path.join(LIBDIR, "variants", variant + ".so")
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    hdr = newest_header()
    todo = []
    objs = []
    for name, src, extra in units():
        obj = os.path.join(OBJ, name)
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            todo.append((obj, src, extra))

    def compile_one(job):
        obj, src, extra = job
        cmd = [cc] + FLAGS + list(defs) + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose:
            print("  built", os.path.basename(obj), flush=True)
        return obj

    jobs = jobs or min(8, os.cpu_count() or 1)
    if todo:
        if verbose:
            print("compiling %d translation units for gfx950 with %d jobs" % (len(todo), jobs), flush=True)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default=None)
    ap.add_argument("--defs", default="", help="extra compiler flags for a variant build, space separated")
    a = ap.parse_args()
    build(a.j, a.force, variant=a.variant, defs=tuple(a.defs.split()))
    sys.exit(0)
