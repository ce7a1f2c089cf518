#!/usr/bin/env python3
"""Developer check (round 4): the sharded entry points over a REAL RCCL process group of one rank - the only RCCL exercise a
one-GPU box allows (two ranks on one device are refused by RCCL).  It covers process-group creation with device_id, barrier,
all_reduce (the NaN count of progressive_filter_sharded, the two norms per LSQR iteration, the extent reduction and the
all_to_all of create_dem_sharded with itself as only peer); the neighbour send / recv has no peer at world size 1.

    python tools/rccl_single_rank_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # the repository this file sits in
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import neilpy_amd
from neilpy_amd import sharded
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(11)
print("backend", dist.get_backend())
n = 1024
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=3)).to(dev)
win = np.arange(1, 13); thr = .15 * win
m, _ = sharded.progressive_filter_sharded(Z, n, win, thr, rank=0, world_size=1)
ref = neilpy_amd.progressive_filter(Z, win, 1, .15)
t = torch.tensor([float(m.sum())], dtype=torch.float64, device=dev); dist.all_reduce(t); dist.barrier()
print("objects", int(t.item()), int(ref.sum().item()), bool(torch.equal(m.bool(), ref)))
# spring inpaint over the band path on one rank
A = torch.from_numpy(neilpy_amd.synth_dem(512, seed=4).astype(np.float64)).to(dev)
A[torch.rand((512, 512), device=dev) < 0.3] = float("nan")
B = A.clone()
st = sharded.inpaint_nans_by_springs_sharded(A, 512, rank=0, world_size=1)      # fills A in place; the all-reduces go over RCCL
neilpy_amd.inpaint_nans_by_springs(B, inplace=True)
print("springs (istop, itn, unknowns)", st, "max |band - single device|", float((A - B).abs().max()))
x = torch.rand(200000, dtype=torch.float64, device=dev) * 500; y = torch.rand(200000, dtype=torch.float64, device=dev) * 500
z = torch.rand(200000, dtype=torch.float64, device=dev) * 30
g1, _e, t1 = sharded.create_dem_sharded(x, y, z, 1.0, "min", rank=0, world_size=1)[:3]
g0, t0 = neilpy_amd.create_dem(x, y, z, 1.0, "min")
print("create_dem_sharded equal", bool(torch.equal(torch.nan_to_num(g1, nan=-1.0), torch.nan_to_num(g0, nan=-1.0))), tuple(t1)[:6] == tuple(t0)[:6])
# the halo exchange's own primitive - batch_isend_irecv of row blocks of a raster - with this rank as its own neighbour (RCCL
# runs a send and the matching recv of one rank inside a group call as a device copy): the P2P path of sharded._exchange
last = torch.arange(64 * 1024, dtype=torch.float32, device=dev).reshape(64, 1024)
recv = torch.zeros((8, 1024), dtype=torch.float32, device=dev)
ops = [dist.P2POp(dist.isend, last[10:18], 0), dist.P2POp(dist.irecv, recv, 0)]
for req in dist.batch_isend_irecv(ops):
    req.wait()
torch.cuda.synchronize()
print("self send/recv of a row block over RCCL equal", bool(torch.equal(recv, last[10:18])))
dist.destroy_process_group()
