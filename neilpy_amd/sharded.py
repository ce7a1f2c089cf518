"""Row-band sharded progressive_filter: one process per GPU, halo exchange between neighbours.

The raster's rows are split into ``world_size`` contiguous bands.  A grey opening by ``disk(r)``
needs ``last`` on 2r rows either side of the band (r for the erosion the dilation reads, r more
for that erosion's own footprint), so every window does ONE exchange of 2r rows with each
neighbour (RCCL send/recv over the direct xGMI link; nearest-neighbour only, no collective),
recomputes the erosion on the r halo rows redundantly and then dilates + flags its own rows.
Image borders use scipy's reflect rule inside the kernels, exactly as on one GPU, so the result
is bit-identical to the single-device path (neilpy.py:1659-1680 semantics).

The communication uses ``torch.distributed`` point-to-point ops, so the same driver runs on
``nccl`` (= RCCL) with CUDA tensors and on ``gloo`` with CPU tensors; the compute is delegated
to a ``BandOps`` object.  The product default is :class:`HipBandOps` (libsmrf_hip); the CPU tests
inject an oracle-backed ops object to exercise the partitioning and halo logic without a GPU.
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["band_rows", "HipBandOps", "progressive_filter_sharded"]


def band_rows(img_rows, world_size, rank):
    """Global row range [b0, b1) owned by ``rank`` (bands differ by at most one row)."""
    base, rem = divmod(img_rows, world_size)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


class HipBandOps:
    """Band compute on the GPU through the C ABI (smrf_disk_filter_* / smrf_pf_dilate_flag_*)."""

    def __init__(self, impl=_lib.IMPL_AUTO):
        self.lib = _lib.load()
        _lib.require_gpu()
        self.impl = impl

    @staticmethod
    def _sfx(t):
        import torch
        return "f32" if t.dtype == torch.float32 else "f64"

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def erode(self, src, src_row0, dst, dst_row0, dst_rows, img_rows, radius):
        fn = getattr(self.lib, "smrf_disk_filter_" + self._sfx(src))
        _lib.check(fn(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), img_rows, src.shape[1], src.shape[1],
                      src_row0, src.shape[0], dst_row0, dst_rows, int(radius), 0, 0, self.impl, self._stream()))

    def dilate_flag(self, eroded, er_row0, er_rows, last_band, opened_band, mask, when, thr, widx, band_row0,
                    band_nrows, img_rows, radius):
        fn = getattr(self.lib, "smrf_pf_dilate_flag_" + self._sfx(eroded))
        cols = eroded.shape[1]
        _lib.check(fn(C.c_void_p(eroded.data_ptr()), C.c_void_p(last_band.data_ptr()),
                      C.c_void_p(opened_band.data_ptr()), C.c_void_p(mask.data_ptr()),
                      C.c_void_p(when.data_ptr()) if when is not None else C.c_void_p(0), float(thr), int(widx),
                      img_rows, cols, cols, er_row0, er_rows, band_row0, band_nrows, int(radius), 0, self.impl,
                      self._stream()))


def _exchange(dist, group, rank, world, send_up, recv_up, send_down, recv_down):
    """send_up -> rank-1 (its bottom halo), send_down -> rank+1 (its top halo).

    nccl (= RCCL) moves the CUDA row blocks directly over xGMI.  On a gloo group (the CPU tests,
    and multi-process rehearsals on a one-GPU box) CUDA tensors are staged through host memory.
    """
    def peer(r):
        return dist.get_global_rank(group, r) if group is not None else r
    staged = dist.get_backend(group) == "gloo" and any(t is not None and t.is_cuda
                                                       for t in (send_up, recv_up, send_down, recv_down))
    pairs = []
    if rank > 0:
        pairs.append((send_up, recv_up, peer(rank - 1)))
    if rank < world - 1:
        pairs.append((send_down, recv_down, peer(rank + 1)))
    ops, back = [], []
    for snd, rcv, p in pairs:
        if staged:
            h_s, h_r = snd.cpu(), rcv.cpu()
            back.append((rcv, h_r))
            snd, rcv = h_s, h_r
        ops.append(dist.P2POp(dist.isend, snd, p, group))
        ops.append(dist.P2POp(dist.irecv, rcv, p, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for dst, h in back:
        dst.copy_(h)


def progressive_filter_sharded(Z_band, img_rows, windows, thresholds, *, rank=None, world_size=None, group=None,
                               ops=None, return_when_dropped=False, state=None):
    """progressive_filter on this rank's row band ``Z_band`` (rows ``band_rows(img_rows, W, rank)``).

    Returns the band's ``(mask, when_dropped | None)`` as uint8 tensors on ``Z_band``'s device.
    ``thresholds`` = ``slope_threshold * (windows * cellsize)`` in float64, as on one device.
    Every band must have at least ``2 * max(windows)`` rows (a halo never spans two ranks).
    ``state`` (a dict) keeps the extended buffers between calls so a benchmark loop does not
    re-allocate.
    """
    import torch
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if ops is None:
        ops = HipBandOps()
    windows = [int(w) for w in np.asarray(windows).ravel()]
    thresholds = np.asarray(thresholds, dtype=np.float64).ravel()
    if len(windows) != thresholds.size:
        raise ValueError("windows and thresholds differ in length")
    b0, b1 = band_rows(img_rows, world_size, rank)
    nloc = b1 - b0
    if Z_band.shape[0] != nloc:
        raise ValueError("band has %d rows, expected %d" % (Z_band.shape[0], nloc))
    cols = Z_band.shape[1]
    rmax = max(windows) if windows else 0
    min_band = min(band_rows(img_rows, world_size, k)[1] - band_rows(img_rows, world_size, k)[0]
                   for k in range(world_size))
    if world_size > 1 and min_band < 2 * rmax:
        raise ValueError("row bands of %d rows are shorter than the 2*%d halo rows a window needs; "
                         "use fewer ranks for this raster" % (min_band, rmax))
    dev, dt = Z_band.device, Z_band.dtype
    H = 2 * rmax if world_size > 1 else 0
    e0, e1 = max(0, b0 - H), min(img_rows, b1 + H)          # rows the extended buffers can hold
    st = state if state is not None else {}
    key = (nloc, cols, H, str(dt), str(dev))
    if st.get("key") != key:
        st.clear()
        st["key"] = key
        st["ext"] = [torch.empty((e1 - e0, cols), dtype=dt, device=dev) for _ in range(2)]
        st["ero"] = torch.empty((e1 - e0, cols), dtype=dt, device=dev)
        st["mask"] = torch.empty((nloc, cols), dtype=torch.uint8, device=dev)
        st["when"] = torch.empty((nloc, cols), dtype=torch.uint8, device=dev)
    ext, ero, mask = st["ext"], st["ero"], st["mask"]
    when = st["when"] if return_when_dropped else None
    mask.zero_()
    if when is not None:
        when.zero_()
    off = b0 - e0                                            # band's first row inside an ext buffer
    cur = 0
    ext[cur][off:off + nloc].copy_(Z_band)
    for i, r in enumerate(windows):
        last = ext[cur]
        if world_size > 1 and r > 0:
            h = 2 * r
            lo, hi = max(0, b0 - h), min(img_rows, b1 + h)
            _exchange(dist, group, rank, world_size,
                      last[off:off + h], last[lo - e0:off] if rank > 0 else None,
                      last[off + nloc - h:off + nloc], last[off + nloc:hi - e0] if rank < world_size - 1 else None)
        else:
            lo, hi = b0, b1
            if world_size == 1:
                lo, hi = 0, img_rows
        # erosion on the band plus r rows either side (clipped at the image border)
        q0, q1 = max(0, b0 - r), min(img_rows, b1 + r)
        src = last[lo - e0:hi - e0]
        dst = ero[q0 - e0:q1 - e0]
        ops.erode(src, lo, dst, q0, q1 - q0, img_rows, r)
        nxt = ext[1 - cur]
        ops.dilate_flag(dst, q0, q1 - q0, last[off:off + nloc], nxt[off:off + nloc], mask, when, float(thresholds[i]),
                        i, b0, nloc, img_rows, r)
        if len(windows) > 1:
            cur = 1 - cur
    return mask, when
