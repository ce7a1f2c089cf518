set -e
out=gpurun_out/r05f; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rehearsal or rccl or springs or inpaint or smrf" > $out/pytest_band.log 2>&1 || { tail -40 $out/pytest_band.log; exit 1; }
tail -2 $out/pytest_band.log
python tools/lsqr_ab.py --size 8193 --occupancy 0.26 --reps 2 --libs neilpy_amd/_lib/variants/base.so > $out/lsqr_ab.log 2>&1; tail -2 $out/lsqr_ab.log
# band form as a whole raster on one rank: the same solve through the phases (4 phases, 2 reductions per iteration)
python - <<'PY' > $out/band_one_rank.log 2>&1
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import neilpy_amd
from neilpy_amd import sharded
n = 4097
g = torch.Generator(device="cuda").manual_seed(7)
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240).astype(np.float64)).cuda()
Z[torch.rand((n, n), device="cuda", generator=g) < 0.74] = float("nan")
ref = neilpy_amd.inpaint_nans_by_springs(Z)
st = dict(neilpy_amd.last_stats["inpaint"])
for rep in range(2):
    A = Z.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = sharded.inpaint_nans_by_springs_sharded(A, n, rank=0, world_size=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("band form on one rank:", out, "single device:", st, "max |diff|", float((A - ref).abs().max()), "equal", bool(torch.equal(A, ref)), "%.1f ms" % (dt * 1e3))
PY
tail -1 $out/band_one_rank.log
