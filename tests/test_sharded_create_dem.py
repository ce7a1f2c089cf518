"""create_dem with the POINTS sharded (neilpy_amd.sharded.create_dem_sharded), world 2 and 3, gloo on CPU.

The device operators cannot run here, so a NumPy stand-in with the same interface is injected (extent, bucket by
destination band, bin the received points); what is under test is the driver: the extent all-reduce, the routing rule
(a point goes to the rank whose rows it will be binned into), the counts + runs all_to_all and the out-of-raster
error on every rank.  The bands concatenated must equal the reference's own grid (tests/golden/create_dem.npz)
bit for bit.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port, golden


class NumpyPointOps:
    """CPU stand-in for sharded.HipPointOps (tests only)."""

    def extent(self, xd, yd):
        if xd.numel() == 0:
            return (np.inf, -np.inf, np.inf, -np.inf)
        x, y = xd.numpy(), yd.numpy()
        return float(x.min()), float(x.max()), float(y.min()), float(y.max())

    @staticmethod
    def _rows_cols(x, y, inv):
        ia, ib, ic, id_, ie, jf = inv
        return np.floor((x * ia + y * ib) + ic), np.floor((x * id_ + y * ie) + jf)

    def bucket(self, xd, yd, zd, inv, rows_total, nbands):
        from neilpy_amd.sharded import band_rows
        x, y, z = xd.numpy(), yd.numpy(), zd.numpy()
        _, r = self._rows_cols(x, y, inv)
        r = np.clip(np.nan_to_num(r, nan=0.0), 0, rows_total - 1).astype(np.int64)
        starts = np.array([band_rows(rows_total, nbands, k)[0] for k in range(nbands)])
        dest = np.searchsorted(starts, r, side="right") - 1
        order = np.argsort(dest, kind="stable")[::-1]                    # any order inside a run is allowed: reverse it
        order = order[np.argsort(dest[order], kind="stable")]
        counts = np.bincount(dest, minlength=nbands).astype(np.int64)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a[order]))   # noqa: E731
        return torch.from_numpy(counts), f(x), f(y), f(z)

    def bin_band(self, xd, yd, zd, inv, grid_shape, row0, rows_local, bin_type):
        ny, nx = grid_shape
        x, y, z = xd.numpy(), yd.numpy(), zd.numpy()
        c, r = self._rows_cols(x, y, inv)
        inside = (c >= 0) & (c < nx) & (r >= 0) & (r < ny)
        n_out = int((~inside).sum())
        keep = inside & ~np.isnan(z) & (r >= row0) & (r < row0 + rows_local)
        ri, ci = (r[keep] - row0).astype(np.int64), c[keep].astype(np.int64)
        band = np.full((rows_local, nx), np.inf if bin_type == "min" else -np.inf)
        (np.minimum if bin_type == "min" else np.maximum).at(band, (ri, ci), z[keep])
        empty = ~np.isfinite(band) & ~np.isin(band, z[keep][np.isinf(z[keep])])
        band[empty] = np.nan
        return torch.from_numpy(band), torch.from_numpy(empty.astype(np.uint8)), n_out


def _worker(rank, world, port, tag, out_dir, poison):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neilpy_amd import sharded
        g = golden("create_dem.npz")
        kw = json.loads(str(g[tag + "_kwargs_json"]))
        x, y, z = g[tag + "_x"], g[tag + "_y"], g[tag + "_z"]
        # an uneven split of the cloud, in file order: rank 0 gets the fewest points
        cut = [0] + [int(len(x) * (k + 1) ** 2 / world ** 2) for k in range(world)]
        sl = slice(cut[rank], cut[rank + 1])
        xs, ys, zs = (torch.from_numpy(np.ascontiguousarray(v[sl])) for v in (x, y, z))
        if poison and rank == world - 1:
            xs = xs.clone()
            xs[0] = float("nan")                                           # np.min propagates NaN: no raster
        try:
            band, empty, t, shape, (b0, b1) = sharded.create_dem_sharded(xs, ys, zs, kw.get("cellsize", 1), kw.get("bin_type", "max"),
                                                                         rank=rank, world_size=world, ops=NumpyPointOps())
            np.savez(os.path.join(out_dir, "r%d.npz" % rank), band=band.numpy(), empty=empty.numpy(), t=np.array(t[:6]),
                     shape=np.array(shape), b0=b0, b1=b1)
        except ValueError as e:
            np.savez(os.path.join(out_dir, "r%d.npz" % rank), error=np.array(str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tag", [(2, "min_cs1"), (3, "max_cs1"), (3, "min_cs0p3"), (2, "min_cs2p5"), (3, "nanz"),
                                       (2, "negcoords"), (3, "default")])
def test_create_dem_sharded_equals_reference(tmp_path, world, tag):
    g = golden("create_dem.npz")
    want = g[tag + "_I"]
    mp.spawn(_worker, args=(world, free_port(), tag, str(tmp_path), False), nprocs=world, join=True)
    got = np.full(want.shape, -1.0)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert tuple(d["shape"]) == want.shape
        assert np.array_equal(d["t"], g[tag + "_transform"])
        got[int(d["b0"]):int(d["b1"])] = d["band"]
        assert np.array_equal(d["empty"].astype(bool), np.isnan(d["band"]))
    assert np.array_equal(got, want, equal_nan=True)


def test_create_dem_sharded_nan_coordinate_raises_everywhere(tmp_path):
    mp.spawn(_worker, args=(2, free_port(), "min_cs1", str(tmp_path), True), nprocs=2, join=True)
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert "error" in d.files and "extent" in str(d["error"])
