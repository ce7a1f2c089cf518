"""Row-band sharding + halo exchange of progressive_filter, world_size 2 and 3, gloo on CPU.

The product compute (HIP) cannot run here, so the band operations are injected: an
oracle-backed BandOps that applies scipy's erosion / dilation to the extended band (halo rows
present) and crops.  What is under test is neilpy_amd.sharded: band partitioning, the grouped
halo exchange (sum of 2r rows per group of windows), the reflect handling at the raster's true borders and the flag/mask logic.
The sharded result must equal the single-process oracle bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port


class OracleBandOps:
    """BandOps for CPU tensors built on the oracle (tests only)."""

    def __init__(self):
        from oracle import smrf_oracle as orc
        self.orc = orc

    def _filter(self, src, src_row0, out_row0, out_rows, img_rows, radius, dilate):
        a = src.numpy()
        fp = self.orc.disk(radius)
        # scipy reflects at the array's ends: right at the raster's true borders; at interior band
        # edges the halo rows make the reflected part irrelevant for the rows we keep
        if src_row0 != 0 or src_row0 + a.shape[0] != img_rows:
            lo_ok = src_row0 == 0 or out_row0 - radius >= src_row0
            hi_ok = src_row0 + a.shape[0] == img_rows or out_row0 + out_rows + radius <= src_row0 + a.shape[0]
            assert lo_ok and hi_ok, "band lacks halo rows"
        f = self.orc.dilation if dilate else self.orc.erosion
        full = f(a, fp)
        return full[out_row0 - src_row0:out_row0 - src_row0 + out_rows]

    def erode(self, src, src_row0, dst, dst_row0, dst_rows, img_rows, radius):
        dst.copy_(torch.from_numpy(self._filter(src, src_row0, dst_row0, dst_rows, img_rows, radius, False)))

    def dilate_flag(self, eroded, er_row0, er_rows, last_band, opened_band, mask, when, thr, widx, band_row0,
                    band_nrows, img_rows, radius):
        opened = self._filter(eroded, er_row0, band_row0, band_nrows, img_rows, radius, True)
        opened_band.copy_(torch.from_numpy(opened))
        new_obj = (last_band.numpy() - opened) > np.float64(thr)
        mask.numpy()[new_obj] = 1
        if when is not None:
            when.numpy()[new_obj] = widx


def _worker(rank, world, port, shape, windows, dtype_name, out_dir, budget=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neilpy_amd import sharded
        from neilpy_amd.synth import synth_dem
        dtype = np.float32 if dtype_name == "f32" else np.float64
        Z = synth_dem(shape[1], seed=31, dtype=dtype, rows=shape[0])
        b0, b1 = sharded.band_rows(shape[0], world, rank)
        band = torch.from_numpy(np.ascontiguousarray(Z[b0:b1]))
        win = np.asarray(windows)
        thr = .15 * (win * 1)
        state = {}
        for _ in range(2):                       # second call reuses the buffers
            mask, when = sharded.progressive_filter_sharded(band, shape[0], win, thr, rank=rank, world_size=world,
                                                            ops=OracleBandOps(), return_when_dropped=True, state=state,
                                                            halo_budget=budget, overlap=(len(windows) % 2 == 0))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), mask=mask.numpy(), when=when.numpy(), b0=b0, b1=b1,
                 exchanges=state["exchanges"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,windows,dtype,budget,exchanges", [
    (2, (96, 70), [1, 2, 3, 5, 8], "f32", None, 1),          # one group: 38 halo rows <= the 48-row bands
    (2, (96, 70), [1, 2, 3, 5, 8], "f32", 0, 5),             # one exchange per window
    (2, (61, 40), [1, 4, 9, 12], "f64", None, 2),            # (1, 4, 9) | 12 on 30-row bands
    (3, (150, 33), [2, 1, 7, 11], "f32", 20, 2),             # (2, 1, 7) | 11 with a 20-row budget
])
def test_sharded_equals_single(tmp_path, world, shape, windows, dtype, budget, exchanges):
    from oracle import smrf_oracle as orc
    from neilpy_amd.synth import synth_dem
    port = free_port()
    mp.spawn(_worker, args=(world, port, shape, windows, dtype, str(tmp_path), budget), nprocs=world, join=True)
    Z = synth_dem(shape[1], seed=31, dtype=np.float32 if dtype == "f32" else np.float64, rows=shape[0])
    want_m, want_w = orc.progressive_filter(Z, np.asarray(windows), 1, .15, return_when_dropped=True)
    got_m = np.zeros(shape, np.uint8)
    got_w = np.zeros(shape, np.uint8)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        got_m[int(d["b0"]):int(d["b1"])] = d["mask"]
        got_w[int(d["b0"]):int(d["b1"])] = d["when"]
        assert int(d["exchanges"]) == exchanges
    assert np.array_equal(got_m.astype(bool), want_m)
    assert np.array_equal(got_w, want_w)


def test_band_too_short_raises():
    from neilpy_amd import sharded
    band = torch.zeros((10, 8))
    with pytest.raises(ValueError, match="halo"):
        sharded.progressive_filter_sharded(band, 20, np.array([6]), np.array([.9]), rank=0, world_size=2,
                                           ops=OracleBandOps())


def test_window_groups():
    from neilpy_amd.sharded import window_groups
    w = list(range(1, 51))
    g = window_groups(w, 2048)                               # the 8-GPU headline bands: 256-row budget
    assert [i for grp in g for i in grp] == list(range(50)) and len(g) == 12
    assert all(len(grp) == 1 or sum(2 * w[i] for i in grp) <= 256 for grp in g)
    assert len(window_groups(w, 8192)) == 6                  # 2 GPUs: 512 rows
    assert len(window_groups(w, 2048, 0)) == 50
    assert window_groups([0, 0, 3], 100) == [[0, 1, 2]]
    assert window_groups([], 100) == []
    assert window_groups([40, 1], 100) == [[0, 1]]           # budget = the band's 100 rows: 80 + 2 fit
    assert window_groups([40, 20], 100) == [[0], [1]]        # 80 + 40 do not


def test_band_rows_partition():
    from neilpy_amd.sharded import band_rows
    for m in (1, 7, 16384, 1000):
        for w in (1, 2, 3, 8):
            edges = [band_rows(m, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
