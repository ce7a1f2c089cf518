#!/usr/bin/env python3
"""Run the row-band sharded SMRF stages (create_dem band -> springs -> progressive_filter) on N ranks
and check them against the golden vectors of one ISPRS sample.  Launch with torch.distributed.run;
``--backend gloo --share-gpu`` lets N ranks share one GPU (halos staged through the host).
Rank 0 prints one JSON line with the verdicts."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--sample", default="samp11")
ap.add_argument("--backend", default="nccl")
ap.add_argument("--share-gpu", action="store_true")
a = ap.parse_args()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from neilpy_amd import api, sharded  # noqa: E402
from neilpy_amd.affine import from_origin  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
lr = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))

G = os.path.join(ROOT, "tests", "golden")
smp = np.load(os.path.join(G, "samples.npz"))
gold = np.load(os.path.join(G, "smrf_%s.npz" % a.sample))
x, y, z = (smp[a.sample + "_" + k] / 100.0 for k in "xyz")
shape = tuple(int(v) for v in gold["shape"])
t = from_origin(gold["transform"][2], gold["transform"][5], 1, 1)
xd, yd, zd = (torch.from_numpy(v).to(dev) for v in (x, y, z))
b0, b1 = sharded.band_rows(shape[0], world, rank)

band, empty, n_out = sharded.create_dem_band(xd, yd, zd, tuple(~t)[:6], shape, rank=rank, world_size=world, bin_type="min")
zmin = np.where(gold["Zmin_centi"] == -2 ** 31, np.nan, gold["Zmin_centi"] / 100.0)
ok_dem = bool(np.array_equal(band.cpu().numpy(), zmin[b0:b1], equal_nan=True)) and n_out == 0

# the same raster from SHARDED points: this rank passes only its 1/N of the cloud (all-to-all by destination band)
p0, p1 = sharded.band_rows(len(x), world, rank)
band2, empty2, t2, shape2, (c0, c1) = sharded.create_dem_sharded(xd[p0:p1].contiguous(), yd[p0:p1].contiguous(),
                                                                  zd[p0:p1].contiguous(), 1, "min", rank=rank, world_size=world)
ok_dem = ok_dem and shape2 == shape and (c0, c1) == (b0, b1) and tuple(t2)[:6] == tuple(t)[:6] and \
    bool(torch.equal(torch.nan_to_num(band2, nan=-1.0), torch.nan_to_num(band, nan=-1.0))) and bool(torch.equal(empty2, empty))

istop, itn, nunk = sharded.inpaint_nans_by_springs_sharded(band, shape[0], rank=rank, world_size=world)
ok_lsqr = (istop, itn) == tuple(int(v) for v in gold["lsqr1"])
if "inpaint1" in gold.files:
    err = float(np.abs(band.cpu().numpy() - gold["inpaint1"][b0:b1]).max())
else:
    err = float("nan")

windows = np.arange(1, 19)
if world > 1 and (b1 - b0) < 2 * 18:
    windows = np.arange(1, (b1 - b0) // 2 + 1)
mask, when = sharded.progressive_filter_sharded(band, shape[0], windows, .15 * (windows * 1), rank=rank, world_size=world,
                                                return_when_dropped=True)
nbits = shape[0] * shape[1]
pf = np.unpackbits(gold["pf_mask_bits"])[:nbits].reshape(shape).astype(bool)
ok_pf = bool(len(windows) == 18 and np.array_equal(mask.cpu().numpy().astype(bool), pf[b0:b1]) and
             np.array_equal(when.cpu().numpy(), gold["pf_when_dropped"][b0:b1]))
res = torch.tensor([float(ok_dem), float(ok_lsqr), float(ok_pf), 0.0 if np.isnan(err) else err], dtype=torch.float64)
if world > 1:
    mn = res.clone()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN) if a.backend == "gloo" else None
    if a.backend == "nccl":
        mn = mn.to(dev)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        mn = mn.cpu()
    mx = res.clone().to(dev if a.backend == "nccl" else "cpu")
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    res = torch.tensor([mn[0], mn[1], mn[2], mx.cpu()[3]])
if rank == 0:
    print(json.dumps(dict(sample=a.sample, world=world, create_dem_band_ok=bool(res[0]), lsqr_itn_ok=bool(res[1]),
                          itn=itn, progressive_filter_ok=bool(res[2]), inpaint_max_abs_err=float(res[3]),
                          n_windows=int(len(windows)))), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
