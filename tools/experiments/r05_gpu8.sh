set -e
out=gpurun_out/r05g; mkdir -p $out
python -m pytest tests/test_fda.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY' > $out/fda_equal.log 2>&1
import sys, os, ctypes as C, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import neilpy_amd
from neilpy_amd import _lib
lib = _lib.load()
old = C.CDLL(os.path.abspath("neilpy_amd/_lib/variants/base.so"))
for f in ("smrf_fda_lsqr_f64", "smrf_fda_workspace_bytes"):
    getattr(old, f).restype, getattr(old, f).argtypes = getattr(lib, f).restype, getattr(lib, f).argtypes
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ok = True
for n, holes, seed in ((300, 0.3, 1), (513, 0.05, 2), (1024, 0.5, 3), (2049, 0.1, 4), (4097, 0.3, 5)):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=seed).astype(np.float64)).cuda()
    Z[torch.rand((n, n), device="cuda", generator=g) < holes] = float("nan")
    res = {}
    for name, L in (("cur", lib), ("r4", old)):
        A = Z.clone()
        nb = L.smrf_fda_workspace_bytes(n, n)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        istop, itn, nunk = C.c_int(0), C.c_int64(0), C.c_int64(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.smrf_fda_lsqr_f64(C.c_void_p(A.data_ptr()), n, n, 1e-6, 1e-6, 1e8, -1, C.byref(istop), C.byref(itn), C.byref(nunk), C.c_void_p(ws.data_ptr()), nb, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert rc == 0
        res[name] = (A, istop.value, itn.value, dt)
    eq = torch.equal(res["cur"][0], res["r4"][0]) and res["cur"][1:3] == res["r4"][1:3]
    ok = ok and eq
    print("fda n=%d holes %.2f: istop/itn %s vs %s, bit-equal %s, %.1f ms vs %.1f ms" % (n, holes, res["cur"][1:3], res["r4"][1:3], eq, res["cur"][3] * 1e3, res["r4"][3] * 1e3), flush=True)
print("fda split bit-identical to round 4 on every case:", ok)
sys.exit(0 if ok else 1)
PY
cat $out/fda_equal.log | grep -v amdgpu
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "band_lsqr or rehearsal" 2>&1 | tail -2
