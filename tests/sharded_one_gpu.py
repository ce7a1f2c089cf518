"""neilpy_amd.sharded's row-band driver exercised on ONE GPU (test helper, no test in here).

Every rank's band runs in turn on the same device; the message a neighbour would send in exchange g is replaced by
the same rows of the surface entering group g, computed on the whole raster with the single-device opening.
Returns the per-band masks (and when_dropped) stitched back into whole rasters plus the exchange sizes seen.
"""
import numpy as np


def run_bands(nz, Z, windows, world, cellsize=1, slope=.15, return_when_dropped=False):
    import torch
    from neilpy_amd import sharded
    N = Z.shape[0]
    win = [int(w) for w in windows]
    thr = slope * (np.asarray(windows) * cellsize)
    min_band = min(sharded.band_rows(N, world, k)[1] - sharded.band_rows(N, world, k)[0] for k in range(world))
    groups = sharded.window_groups(win, min_band)
    halo, last = {}, Z
    for gi, grp in enumerate(groups):
        m = sum(2 * win[i] for i in grp)
        for k in range(world):
            b0, b1 = sharded.band_rows(N, world, k)
            if k > 0:
                halo[(gi, k, "up")] = last[b0 - m:b0].clone()
            if k < world - 1:
                halo[(gi, k, "down")] = last[b1:b1 + m].clone()
        for i in grp:
            last = nz.opening(last, radius=win[i])
    del last
    real = sharded._exchange
    mask = torch.empty(Z.shape, dtype=torch.uint8, device=Z.device)
    when = torch.empty(Z.shape, dtype=torch.uint8, device=Z.device) if return_when_dropped else None
    seen = []
    try:
        for k in range(world):
            b0, b1 = sharded.band_rows(N, world, k)
            calls = []

            def fake_exchange(dist, group, rank, world_size, send_up, recv_up, send_down, recv_down):
                gi = len(calls)
                calls.append(send_up.shape[0])
                if recv_up is not None:
                    recv_up.copy_(halo[(gi, k, "up")])
                if recv_down is not None:
                    recv_down.copy_(halo[(gi, k, "down")])
            sharded._exchange = fake_exchange
            m, w = sharded.progressive_filter_sharded(Z[b0:b1], N, windows, thr, rank=k, world_size=world,
                                                      return_when_dropped=return_when_dropped)
            mask[b0:b1].copy_(m)
            if when is not None:
                when[b0:b1].copy_(w)
            seen.append(calls)
    finally:
        sharded._exchange = real
    want_calls = [sum(2 * win[i] for i in g) for g in groups]
    assert all(c == want_calls for c in seen), (seen, want_calls)
    return mask.bool(), when, groups
