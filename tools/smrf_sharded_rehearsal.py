#!/usr/bin/env python3
"""smrf_sharded on N ranks against the golden vectors of one ISPRS sample (developer / test tool).
Launch with torch.distributed.run; ``--backend gloo --share-gpu`` lets N ranks share one GPU (halos
and gathers staged through the host).  Rank 0 prints one JSON line with the verdicts."""
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
ap.add_argument("--points", default="replicated", choices=["replicated", "sharded"],
                help="sharded: every rank passes only its slice of the cloud (all-to-all gridding, local point flags)")
a = ap.parse_args()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from neilpy_amd import sharded  # noqa: E402

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
meta = json.load(open(os.path.join(G, "meta.json")))
x, y, z = (smp[a.sample + "_" + k] / 100.0 for k in "xyz")
kw = dict(meta["smrf_kwargs"])
p0, p1 = (0, len(x)) if a.points == "replicated" else sharded.band_rows(len(x), world, rank)
dtm, t, obj, pts, (b0, b1) = sharded.smrf_sharded(x[p0:p1], y[p0:p1], z[p0:p1], points=a.points, **kw)
shape = tuple(int(v) for v in gold["shape"])
nbits = shape[0] * shape[1]
want_obj = np.unpackbits(gold["object_cells_bits"])[:nbits].reshape(shape).astype(bool)
want_pts = np.unpackbits(gold["is_object_point_bits"])[:len(x)].astype(bool)
ok_obj = bool(np.array_equal(obj.cpu().numpy(), want_obj[b0:b1]))
ok_pts = bool(np.array_equal(pts.cpu().numpy(), want_pts[p0:p1]))
ok_t = tuple(float(v) for v in tuple(t)[:6]) == tuple(float(v) for v in gold["transform"])
if "Zpro" in gold.files:
    err = float(np.abs(dtm.cpu().numpy() - gold["Zpro"][b0:b1]).max())
else:
    err = abs(float(dtm.sum().item()))          # summed over ranks below and compared with the golden sum
from neilpy_amd import api  # noqa: E402
st = api.last_stats["sharded"]
ok_itn = (tuple(st["inpaint1"][:2]) == tuple(int(v) for v in gold["lsqr1"]) and
          tuple(st["inpaint2"][:2]) == tuple(int(v) for v in gold["lsqr2"]))
res = torch.tensor([float(ok_obj), float(ok_pts), float(ok_t), float(ok_itn)], dtype=torch.float64)
e = torch.tensor([err], dtype=torch.float64)
if world > 1:
    if a.backend == "nccl":
        res, e = res.to(dev), e.to(dev)
    dist.all_reduce(res, op=dist.ReduceOp.MIN)
    dist.all_reduce(e, op=dist.ReduceOp.SUM if "Zpro" not in gold.files else dist.ReduceOp.MAX)
res, e = res.cpu(), e.cpu()
if "Zpro" not in gold.files:
    e = torch.tensor([abs(float(e[0]) - float(gold["Zpro_sum"][0]))])
npts_obj = torch.tensor([float(pts.sum().item())], dtype=torch.float64)
if world > 1 and a.points == "sharded":
    if a.backend == "nccl":
        npts_obj = npts_obj.to(dev)
    dist.all_reduce(npts_obj, op=dist.ReduceOp.SUM)
if rank == 0:
    print(json.dumps(dict(sample=a.sample, world=world, points=a.points, object_cells_ok=bool(res[0]), is_object_point_ok=bool(res[1]),
                          transform_ok=bool(res[2]), lsqr_itn_ok=bool(res[3]),
                          dtm_err=float(e[0]), object_points=int(npts_obj.item()))), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
