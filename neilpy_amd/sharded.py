"""Row-band sharded progressive_filter: one process per GPU, halo exchange between neighbours.

The raster's rows are split into ``world_size`` contiguous bands.  A grey opening by ``disk(r)``
needs ``last`` on 2r rows either side of the band (r for the erosion the dilation reads, r more
for that erosion's own footprint), so a window costs 2r rows of the neighbours' bands.
Consecutive windows are grouped (``window_groups``): a group does ONE exchange of sum(2r) rows
with each neighbour (RCCL send/recv over the direct xGMI link; nearest-neighbour only, no
collective) and recomputes its shrinking margins redundantly, so windows 1..50 on 2048-row bands
take 12 exchanges instead of 50.
Image borders use scipy's reflect rule inside the kernels, exactly as on one GPU, so the result
is bit-identical to the single-device path (neilpy.py:1659-1680 semantics).

Overlap (``overlap=True``): the rows a group's exchange sends are the band's outermost sum(2r) rows of the
surface the previous group leaves.  That group's last dilation is split: the two edge strips run first, on
a side stream, followed by the exchange; the interior runs beside them on the main stream, which only
waits for the exchange before the next group's first erosion.

The communication uses ``torch.distributed`` point-to-point ops, so the same driver runs on
``nccl`` (= RCCL) with CUDA tensors and on ``gloo`` with CPU tensors; the compute is delegated
to a ``BandOps`` object.  The product default is :class:`HipBandOps` (libsmrf_hip); the CPU tests
inject an oracle-backed ops object to exercise the partitioning and halo logic without a GPU.
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["band_rows", "window_groups", "HipBandOps", "progressive_filter_sharded", "HipSpringsOps",
           "inpaint_nans_by_springs_sharded", "create_dem_band", "HipPointOps", "create_dem_sharded", "smrf_sharded"]


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

    def has_nan(self, band):
        """NaNs in this rank's rows?  (scipy's filters let the first visited footprint element decide NaN-ness; the
        kernels follow that rule only when told to, so the driver asks once per call and all-reduces the answer)"""
        cnt = C.c_int64(0)
        fn = getattr(self.lib, "smrf_count_nan_" + self._sfx(band))
        _lib.check(fn(C.c_void_p(band.data_ptr()), band.numel(), C.byref(cnt), self._stream()))
        return cnt.value > 0

    def can_fuse(self, t, radius):
        """a window of this radius can run as one fused opening + flag launch on a band (csrc/morph_fused.h).  Only the
        radii whose 4r warm-up rows are short against a band's segments: r <= 8."""
        import torch
        return 1 <= radius <= 8 and bool(self.lib.smrf_fused_open_supported(4 if t.dtype == torch.float32 else 8, int(radius)))

    def chain_len(self, t, radii, raster_cells):
        """how many of the windows at the head of ``radii`` one chained / table-free launch takes on a raster of this size
        (csrc/morph_chain.h; 0 = none)"""
        import torch
        # (the library applies its own SMRF_CHAIN / SMRF_FUSED switches here, the ones smrf_progressive_filter_* routes by)
        r = np.ascontiguousarray(np.asarray(radii[:4], dtype=np.int32))
        return int(self.lib.smrf_pf_chain_length(4 if t.dtype == torch.float32 else 8, r.ctypes.data_as(C.c_void_p), int(r.size),
                                                 int(raster_cells)))

    def chain_flag(self, last, last_row0, opened, mask, when, radii, thr, widx, out_row0, out_rows, img_rows):
        """the windows ``radii`` opened one after the other in ONE launch: ``opened`` = the last surface on global rows
        [out_row0, out_row0 + out_rows), every window's flags on those rows; ``last`` reaches sum(2r) rows beyond them"""
        fn = getattr(self.lib, "smrf_pf_chain_flag_" + self._sfx(last))
        cols = last.shape[1]
        r = np.ascontiguousarray(np.asarray(radii, dtype=np.int32))
        t = np.ascontiguousarray(np.asarray(thr, dtype=np.float64))
        w = np.ascontiguousarray(np.asarray(widx, dtype=np.int32))
        _lib.check(fn(C.c_void_p(last.data_ptr()), C.c_void_p(opened.data_ptr()), C.c_void_p(mask.data_ptr()),
                      C.c_void_p(when.data_ptr()) if when is not None else C.c_void_p(0), r.ctypes.data_as(C.c_void_p),
                      t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), int(r.size), img_rows, cols, cols, last_row0,
                      last.shape[0], out_row0, out_rows, self._stream()))

    def open_flag(self, last, last_row0, opened, mask, when, thr, widx, out_row0, out_rows, img_rows, radius):
        """opened = opening(last, disk(r)) on global rows [out_row0, out_row0 + out_rows) + the flag step, one launch;
        ``last`` holds global rows from ``last_row0`` and reaches 2r rows beyond the outputs"""
        fn = getattr(self.lib, "smrf_pf_open_flag_" + self._sfx(last))
        cols = last.shape[1]
        _lib.check(fn(C.c_void_p(last.data_ptr()), C.c_void_p(opened.data_ptr()), C.c_void_p(mask.data_ptr()),
                      C.c_void_p(when.data_ptr()) if when is not None else C.c_void_p(0), float(thr), int(widx), img_rows, cols,
                      cols, last_row0, last.shape[0], out_row0, out_rows, int(radius), self._stream()))

    def erode(self, src, src_row0, dst, dst_row0, dst_rows, img_rows, radius, nan_aware=0):
        fn = getattr(self.lib, "smrf_disk_filter_" + self._sfx(src))
        _lib.check(fn(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), img_rows, src.shape[1], src.shape[1],
                      src_row0, src.shape[0], dst_row0, dst_rows, int(radius), 0, int(nan_aware), self.impl, self._stream()))

    def dilate_flag(self, eroded, er_row0, er_rows, last_band, opened_band, mask, when, thr, widx, band_row0,
                    band_nrows, img_rows, radius, nan_aware=0):
        fn = getattr(self.lib, "smrf_pf_dilate_flag_" + self._sfx(eroded))
        cols = eroded.shape[1]
        _lib.check(fn(C.c_void_p(eroded.data_ptr()), C.c_void_p(last_band.data_ptr()),
                      C.c_void_p(opened_band.data_ptr()), C.c_void_p(mask.data_ptr()),
                      C.c_void_p(when.data_ptr()) if when is not None else C.c_void_p(0), float(thr), int(widx),
                      img_rows, cols, cols, er_row0, er_rows, band_row0, band_nrows, int(radius), int(nan_aware), self.impl,
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


def window_groups(windows, min_band_rows, budget=None):
    """Consecutive windows that share ONE halo exchange: a group needs sum(2r) rows of the
    neighbours' bands up front and recomputes its margins redundantly while they shrink by 2r per
    window.  Greedy: a group grows while its halo stays within ``budget`` rows (default: 1/16 of
    the shortest band, at least 256; a single window always fits).  Fewer, larger messages: the
    small radii, whose kernels are short, would otherwise pay one exchange latency each.  The
    default trades about 5 % redundant compute on a 2048-row band (tools/band_compute.py) for 12
    exchanges instead of 50 over windows 1..50."""
    if budget is None:
        budget = max(256, min_band_rows // 16)
    budget = min(budget, min_band_rows)
    groups, cur, need = [], [], 0
    for i, r in enumerate(windows):
        if cur and need + 2 * r > budget:
            groups.append(cur)
            cur, need = [], 0
        cur.append(i)
        need += 2 * r
    if cur:
        groups.append(cur)
    return groups


def progressive_filter_sharded(Z_band, img_rows, windows, thresholds, *, rank=None, world_size=None, group=None,
                               ops=None, return_when_dropped=False, state=None, halo_budget=None, overlap=False):
    """progressive_filter on this rank's row band ``Z_band`` (rows ``band_rows(img_rows, W, rank)``).

    Returns the band's ``(mask, when_dropped | None)`` as uint8 tensors on ``Z_band``'s device.
    ``thresholds`` = ``slope_threshold * (windows * cellsize)`` in float64, as on one device.
    Every band must have at least ``2 * max(windows)`` rows (a halo never spans two ranks).
    ``state`` (a dict) keeps the extended buffers between calls so a benchmark loop does not
    re-allocate (and reports ``state["exchanges"]``; with ``state["profile"] = True`` also ``exchange_events``, a pair
    of CUDA events around every exchange, ``exchange_bytes``, the bytes this rank sent in each, and ``groups``).  ``halo_budget``: rows of halo a group of
    consecutive windows may share in one exchange (see :func:`window_groups`; 0 = one exchange
    per window).  ``overlap=True`` splits every group's last dilation edge-first and posts the next group's exchange
    on a side stream beside the interior (module docstring).  Off by default: on a 2048-row band the two edge
    strips cost about 160 us per group, more than an exchange of up to about 0.45 ms returns
    (tools/band_compute.py, gpurun_out/r02/band_compute.log); it pays on slower links only.
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
    max_band = max(band_rows(img_rows, world_size, k)[1] - band_rows(img_rows, world_size, k)[0]
                   for k in range(world_size))
    if world_size > 1 and min_band < 2 * rmax:
        raise ValueError("row bands of %d rows are shorter than the 2*%d halo rows a window needs; "
                         "use fewer ranks for this raster" % (min_band, rmax))
    dev, dt = Z_band.device, Z_band.dtype
    groups = window_groups(windows, min_band, halo_budget) if world_size > 1 else [[i] for i in range(len(windows))]
    H = max([sum(2 * windows[i] for i in g) for g in groups], default=0) if world_size > 1 else 0
    e0, e1 = max(0, b0 - H), min(img_rows, b1 + H)          # rows the extended buffers can hold
    st = state if state is not None else {}
    key = (nloc, cols, H, str(dt), str(dev))
    if st.get("key") != key:
        keep = st.get("profile")
        st.clear()
        st["key"] = key
        if keep:
            st["profile"] = keep
        st["ext"] = [torch.empty((e1 - e0, cols), dtype=dt, device=dev) for _ in range(2)]
        st["ero"] = torch.empty((e1 - e0, cols), dtype=dt, device=dev)
        st["mask"] = torch.empty((e1 - e0, cols), dtype=torch.uint8, device=dev)
        st["when"] = torch.empty((e1 - e0, cols), dtype=torch.uint8, device=dev)
    ext, ero, mask = st["ext"], st["ero"], st["mask"]
    when = st["when"] if return_when_dropped else None
    mask.zero_()
    if when is not None:
        when.zero_()
    off = b0 - e0                                            # band's first row inside an ext buffer
    cur = 0
    ext[cur][off:off + nloc].copy_(Z_band)
    st["exchanges"] = 0
    # scipy's NaN rule (first visited footprint element decides) costs the kernels a read per cell: only when some
    # band holds a NaN, as the single-device path decides with one count
    nan_aware = 0
    if hasattr(ops, "has_nan"):
        nan_aware = int(bool(ops.has_nan(Z_band)))
        if world_size > 1 and dist.is_initialized():
            flag = torch.tensor([nan_aware], dtype=torch.int32, device=dev if dist.get_backend(group) != "gloo" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            nan_aware = int(flag.item())
    kw = {"nan_aware": nan_aware} if nan_aware else {}
    # overlap (CUDA tensors): the exchange a group needs is posted on a side stream as soon as the previous group's
    # last dilation has produced the band's edge rows, and runs beside that dilation's interior
    side = st.get("side")
    if side is None and Z_band.is_cuda and world_size > 1:
        side = st["side"] = torch.cuda.Stream(device=dev)
    M_of = [sum(2 * windows[i] for i in g) if world_size > 1 else 0 for g in groups]
    st["groups"] = [[windows[i] for i in g] for g in groups]

    profile = bool(st.get("profile")) and Z_band.is_cuda      # bench.py: events either side of every exchange
    if profile:
        st["exchange_events"], st["exchange_bytes"] = [], []

    def exchange(last, M):
        lo, hi = max(0, b0 - M), min(img_rows, b1 + M)
        if profile:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _exchange(dist, group, rank, world_size,
                  last[off:off + M], last[lo - e0:off] if rank > 0 else None,
                  last[off + nloc - M:off + nloc], last[off + nloc:hi - e0] if rank < world_size - 1 else None)
        if profile:
            ev1.record()
            st["exchange_events"].append((ev0, ev1))
            st["exchange_bytes"].append(M * cols * Z_band.element_size() * ((rank > 0) + (rank < world_size - 1)))
        st["exchanges"] += 1

    posted = None                                            # event of an exchange posted ahead for the next group
    for gi, grp in enumerate(groups):
        # rows of `last` the whole group needs beyond the band: every opening eats 2r of them
        M = M_of[gi]
        if M > 0:
            if posted is not None:
                if posted is not True:
                    torch.cuda.current_stream().wait_event(posted)
                posted = None
            else:
                exchange(ext[cur], M)
        gpos = 0
        while gpos < len(grp):
            i = grp[gpos]
            gpos += 1
            r = windows[i]
            last = ext[cur]
            # runs of small windows inside a group as ONE launch (HipBandOps.chain_flag): the margin they eat is the sum of
            # theirs; not with NaNs (no NaN rule there) and not where the group's last window is split edge-first (overlap)
            if (world_size > 1 and not nan_aware and not overlap and hasattr(ops, "chain_len")):
                # sized by the cells a launch of this group marches (a band plus its still-valid margin), not by the whole
                # raster: the chain 4, 5 and the table-free R = 10 only pay from 20 Mi cells up (chain.hip, min_cells).
                # The SAME figure on every rank - the longest band with a two-sided margin - so that all ranks route a
                # window the same way (an edge rank's one-sided margin would otherwise put it on the other side of a size
                # threshold than its neighbour; every route gives the same bits, but the ranks' launches should not differ).
                # A chain whose halo is not shorter than the raster has no kernel (smrf_pf_chain_flag_* refuses it).
                k = ops.chain_len(last, [windows[j] for j in grp[gpos - 1:]], min(img_rows, max_band + 2 * M) * cols)
                if k >= 1 and sum(2 * windows[j] for j in grp[gpos - 1:gpos - 1 + k]) >= img_rows:
                    k = 0
                if k >= 1:
                    members = grp[gpos - 1:gpos - 1 + k]
                    lo, hi = max(0, b0 - M), min(img_rows, b1 + M)
                    M -= sum(2 * windows[j] for j in members)
                    o0, o1 = max(0, b0 - M), min(img_rows, b1 + M)
                    nxt = ext[1 - cur]
                    ops.chain_flag(last[lo - e0:hi - e0], lo, nxt[o0 - e0:o1 - e0], mask[o0 - e0:o1 - e0],
                                   when[o0 - e0:o1 - e0] if when is not None else None, [windows[j] for j in members],
                                   [float(thresholds[j]) for j in members], members, o0, o1 - o0, img_rows)
                    gpos += k - 1
                    if len(windows) > 1:
                        cur = 1 - cur
                    continue
            if world_size == 1:
                lo, hi, q0, q1, o0, o1 = 0, img_rows, 0, img_rows, 0, img_rows
            else:
                lo, hi = max(0, b0 - M), min(img_rows, b1 + M)          # rows of `last` that are valid now
                q0, q1 = max(0, b0 - M + r), min(img_rows, b1 + M - r)  # erosion: r rows inside them
                M -= 2 * r
                o0, o1 = max(0, b0 - M), min(img_rows, b1 + M)          # opening: r more rows inside (the band at the end)
            dst = ero[q0 - e0:q1 - e0]
            nxt = ext[1 - cur]
            fused = not nan_aware and hasattr(ops, "can_fuse") and ops.can_fuse(last, r)
            if not fused:
                ops.erode(last[lo - e0:hi - e0], lo, dst, q0, q1 - q0, img_rows, r, **kw)

            def dilate(y0, y1):
                # margin rows are flagged too (same values the neighbour computes for them); only the band's are returned
                if fused:                                    # small disk: opening + flag of these rows in one launch
                    ops.open_flag(last[lo - e0:hi - e0], lo, nxt[y0 - e0:y1 - e0], mask[y0 - e0:y1 - e0],
                                  when[y0 - e0:y1 - e0] if when is not None else None, float(thresholds[i]), i, y0, y1 - y0,
                                  img_rows, r)
                    return
                ops.dilate_flag(dst, q0, q1 - q0, last[y0 - e0:y1 - e0], nxt[y0 - e0:y1 - e0], mask[y0 - e0:y1 - e0],
                                when[y0 - e0:y1 - e0] if when is not None else None, float(thresholds[i]), i, y0, y1 - y0,
                                img_rows, r, **kw)
            Mn = M_of[gi + 1] if (overlap and i == grp[-1] and gi + 1 < len(groups)) else 0
            if Mn > 0 and o0 == b0 and o1 == b1 and nloc >= 2 * Mn:
                # last window of the group: the Mn edge rows either side first, then the next group's exchange beside
                # the interior (the neighbours need exactly those rows of the opened surface)
                if side is not None:
                    ready = torch.cuda.Event()
                    ready.record()                                       # the erosion is enqueued before this point
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        dilate(b0, b0 + Mn)
                        dilate(b1 - Mn, b1)
                        exchange(nxt, Mn)
                        posted = torch.cuda.Event()
                        posted.record()
                    if nloc > 2 * Mn:
                        dilate(b0 + Mn, b1 - Mn)
                else:                                                    # CPU tensors (gloo tests): same split, in order
                    dilate(b0, b0 + Mn)
                    dilate(b1 - Mn, b1)
                    exchange(nxt, Mn)
                    posted = True
                    if nloc > 2 * Mn:
                        dilate(b0 + Mn, b1 - Mn)
            else:
                dilate(o0, o1)
            if len(windows) > 1:
                cur = 1 - cur
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)        # nothing of this call is left running on the side stream
    return mask[off:off + nloc], (when[off:off + nloc] if when is not None else None)


# ------------------------------------------------------------------------------------------
# inpaint_nans_by_springs over row bands: 1-row halos + one scalar all-reduce per reduction
# ------------------------------------------------------------------------------------------
# phases of smrf_springs_band_phase (include/smrf_hip.h; 9 is retired)
PH_MASK, PH_RHS, PH_BNORM, PH_ATU, PH_INIT_ALFA, PH_AV, PH_BETA_RHO, PH_ATUXW, PH_ALFA_TESTS = range(9)
PH_SCATTER = 10


class HipSpringsOps:
    """The band phases of libsmrf_hip's LSQR (smrf_springs_band_*) on this rank's rows."""

    def __init__(self, A_band, has_above, has_below):
        import torch
        self.lib = _lib.load()
        _lib.require_gpu()
        if A_band.dtype != torch.float64 or not A_band.is_contiguous() or not A_band.is_cuda:
            raise TypeError("the band must be a contiguous float64 CUDA tensor")
        self.A = A_band
        self.rows, self.cols = A_band.shape
        self.flags = (int(bool(has_above)), int(bool(has_below)))
        self.nbytes = self.lib.smrf_springs_band_workspace_bytes(self.rows, self.cols)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=A_band.device)
        lay = (C.c_int64 * 8)()
        _lib.check(self.lib.smrf_springs_band_layout(self.rows, self.cols, lay))
        ld = int(lay[6])                                      # cells between plane rows (>= cols: padded to cache lines)
        n2 = (self.rows + 2) * ld
        self.v = self.ws[lay[0]:lay[0] + n2 * 8].view(torch.float64).view(self.rows + 2, ld)[:, :self.cols]
        self.uv = self.ws[lay[1]:lay[1] + n2 * 8].view(torch.float64).view(self.rows + 2, ld)[:, :self.cols]
        self.hole = self.ws[lay[2]:lay[2] + n2].view(self.rows + 2, ld)[:, :self.cols]
        self.abelow = self.ws[lay[3]:lay[3] + self.cols * 8].view(torch.float64)
        self.red2 = self.ws[lay[4]:lay[4] + 16].view(torch.float64)      # [|v|^2, |dk|^2] of the ATUXW phase
        self.red = self.red2[:1]                                          # the single sum of the other phases

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def begin(self, atol, btol, conlim, iter_lim):
        _lib.check(self.lib.smrf_springs_band_begin(self.rows, self.cols, atol, btol, conlim, iter_lim,
                                                    C.c_void_p(self.ws.data_ptr()), self.nbytes, self._stream()))

    def phase(self, ph):
        _lib.check(self.lib.smrf_springs_band_phase(ph, C.c_void_p(self.A.data_ptr()), self.rows, self.cols,
                                                    self.flags[0], self.flags[1], C.c_void_p(self.ws.data_ptr()),
                                                    self.nbytes, self._stream()))

    def status(self):
        istop, itn, nunk, done = C.c_int(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        _lib.check(self.lib.smrf_springs_band_status(C.c_void_p(self.ws.data_ptr()), self.rows, self.cols, C.byref(istop),
                                                     C.byref(itn), C.byref(nunk), C.byref(done), self._stream()))
        return istop.value, itn.value, nunk.value, bool(done.value)


def _shift(dist, group, rank, world, send, recv, up):
    """up=True: every rank sends `send` to rank-1 and receives `recv` from rank+1 (down: the reverse)."""
    dst, src = (rank - 1, rank + 1) if up else (rank + 1, rank - 1)

    def peer(r):
        return dist.get_global_rank(group, r) if group is not None else r
    staged = dist.get_backend(group) == "gloo" and send.is_cuda
    ops, back = [], None
    if 0 <= dst < world:
        ops.append(dist.P2POp(dist.isend, send.cpu() if staged else send.contiguous(), peer(dst), group))
    if 0 <= src < world:
        if staged:
            back = recv.cpu()
            ops.append(dist.P2POp(dist.irecv, back, peer(src), group))
        else:
            ops.append(dist.P2POp(dist.irecv, recv, peer(src), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if back is not None:
        recv.copy_(back)


def _allreduce_sum(dist, group, t):
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def inpaint_nans_by_springs_sharded(A_band, img_rows, *, rank=None, world_size=None, group=None, ops=None,
                                    atol=1e-6, btol=1e-6, conlim=1e8, iter_lim=-1, poll=8):
    """Spring infill (SciPy-LSQR semantics) of a raster split into row bands; ``A_band`` (this rank's
    rows, float64) is filled in place.  Returns ``(istop, itn, n_unknown)`` - equal on every rank.

    Per LSQR iteration: rank r sends the first row of ``v`` up and the last row of ``uv`` down (one
    row each, nearest neighbour) and TWO all-reduces replace the three norms: ``|u|^2`` alone (beta
    must exist before ``v = S^T u - beta v``), then ``[|v|^2, |dk|^2]`` as one 2-element buffer:
    the plane rotation's ``rho`` needs only ``rhobar`` and ``beta`` (csrc/lsqr_core.h: rho_step), so
    ``dk = w / rho`` (lsqr.py:459) - and with it the x and w steps - ride in the pass that makes
    ``v`` (csrc/springs.hip: atuxw_kernel, the single-device solver's kernel; round 5).  Two
    synchronisation points per iteration is LSQR's own minimum (beta, then alfa).  The scalar
    recurrence runs replicated on every rank from the same reduced sums, so all ranks stop at the
    same iteration.  Summation order differs from one device (block partials per band), so
    values agree to ~1e-12 and ``itn`` is normally identical (hard part 1).
    """
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    b0, b1 = band_rows(img_rows, world_size, rank)
    if A_band.shape[0] != b1 - b0:
        raise ValueError("band has %d rows, expected %d" % (A_band.shape[0], b1 - b0))
    multi = world_size > 1
    if ops is None:
        ops = HipSpringsOps(A_band, rank > 0, rank < world_size - 1)
    n = ops.rows

    def reduce():
        if multi:
            _allreduce_sum(dist, group, ops.red)

    def reduce2():
        if multi:
            _allreduce_sum(dist, group, ops.red2)

    ops.begin(atol, btol, conlim, iter_lim)
    ops.phase(PH_MASK)
    if multi:
        _shift(dist, group, rank, world_size, ops.hole[1], ops.hole[n + 1], up=True)
        _shift(dist, group, rank, world_size, ops.A[0], ops.abelow, up=True)
    reduce()
    ops.phase(PH_RHS)
    reduce()
    ops.phase(PH_BNORM)
    if multi:
        _shift(dist, group, rank, world_size, ops.uv[n], ops.uv[0], up=False)
    ops.phase(PH_ATU)
    reduce()
    ops.phase(PH_INIT_ALFA)
    istop, itn, nunk, done = ops.status()
    while not done:
        for _ in range(poll):
            if multi:
                _shift(dist, group, rank, world_size, ops.v[1], ops.v[n + 1], up=True)
            ops.phase(PH_AV)                                 # u_k = S v_{k-1} - alfa u_{k-1}
            reduce()
            ops.phase(PH_BETA_RHO)                           # beta_k; rho_k, t1_k, 1/rho_k need nothing else
            if multi:
                _shift(dist, group, rank, world_size, ops.uv[n], ops.uv[0], up=False)
            ops.phase(PH_ATUXW)                              # w_{k-1}, dk_k, (x_k every second k), v_k
            reduce2()                                        # [|v|^2, |dk|^2] as one buffer
            ops.phase(PH_ALFA_TESTS)                         # alfa_k, rest of the rotation, stopping tests
        istop, itn, nunk, done = ops.status()
    if nunk > 0:
        ops.phase(PH_SCATTER)
    return istop, itn, nunk


def create_dem_band(xd, yd, zd, inv_affine, grid_shape, *, rank, world_size, bin_type='min'):
    """This rank's row band of create_dem's raster from the FULL point set (points are replicated or
    streamed to every rank; no exchange): returns (float64 band, uint8 empty mask, n_outside).

    The binning kernel takes the band as (row0, rows_local) and ignores points of other bands, so N
    ranks read the points N times but write disjoint rows (the all-to-all of points by
    destination band is the alternative when the points do not fit every rank)."""
    import torch
    lib = _lib.load()
    ny, nx = grid_shape
    b0, b1 = band_rows(ny, world_size, rank)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    keys = torch.empty((b1 - b0, nx), dtype=torch.int64, device=xd.device)
    n_out = torch.zeros(1, dtype=torch.int64, device=xd.device)
    grid = torch.empty((b1 - b0, nx), dtype=torch.float64, device=xd.device)
    empty = torch.empty((b1 - b0, nx), dtype=torch.uint8, device=xd.device)
    h_inv = (C.c_double * 6)(*[float(v) for v in inv_affine])
    is_max = 1 if bin_type == 'max' else 0
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.smrf_grid_clear_u64(p(keys), keys.numel(), st))
    _lib.check(lib.smrf_grid_bin_f64(p(xd), p(yd), p(zd), xd.numel(), h_inv, None, p(keys), ny, nx, b0, b1 - b0, is_max,
                                     p(n_out), st))
    _lib.check(lib.smrf_grid_finalize_f64(p(keys), p(grid), p(empty), keys.numel(), is_max, st))
    return grid, empty, int(n_out.item())


class HipPointOps:
    """Point-side device operators of the sharded create_dem (libsmrf_hip, include/smrf_hip.h)."""

    def __init__(self):
        self.lib = _lib.load()
        _lib.require_gpu()

    @staticmethod
    def _st():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def extent(self, xd, yd):
        """(xmin, xmax, ymin, ymax) of this rank's points; (+inf, -inf, +inf, -inf) for none"""
        import torch
        if xd.numel() == 0:
            return (np.inf, -np.inf, np.inf, -np.inf)
        ws = torch.empty(4 * 1024, dtype=torch.float64, device=xd.device)
        ext = (C.c_double * 4)()
        _lib.check(self.lib.smrf_points_extent_f64(C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), xd.numel(), ext,
                                                   C.c_void_p(ws.data_ptr()), ws.numel() * 8, self._st()))
        return tuple(float(v) for v in ext)

    def bucket(self, xd, yd, zd, inv, rows_total, nbands):
        """points grouped by destination row band: (counts int64[nbands] on the device, x, y, z packed runs)"""
        import torch
        h_inv = (C.c_double * 6)(*[float(v) for v in inv])
        p = lambda t: C.c_void_p(t.data_ptr())                  # noqa: E731
        counts = torch.zeros(nbands, dtype=torch.int64, device=xd.device)
        _lib.check(self.lib.smrf_points_band_count_f64(p(xd), p(yd), xd.numel(), h_inv, rows_total, nbands, p(counts), self._st()))
        cursors = torch.cumsum(counts, 0) - counts               # exclusive prefix sum: each band's first slot
        ox, oy, oz = torch.empty_like(xd), torch.empty_like(yd), torch.empty_like(zd)
        _lib.check(self.lib.smrf_points_band_pack_f64(p(xd), p(yd), p(zd), xd.numel(), h_inv, rows_total, nbands, p(cursors),
                                                      p(ox), p(oy), p(oz), self._st()))
        return counts, ox, oy, oz

    def bin_band(self, xd, yd, zd, inv, grid_shape, row0, rows_local, bin_type):
        """(float64 band, uint8 empty mask, points outside the raster) from the points this rank received"""
        import torch
        ny, nx = grid_shape
        p = lambda t: C.c_void_p(t.data_ptr())                  # noqa: E731
        keys = torch.empty((rows_local, nx), dtype=torch.int64, device=xd.device)
        n_out = torch.zeros(1, dtype=torch.int64, device=xd.device)
        grid = torch.empty((rows_local, nx), dtype=torch.float64, device=xd.device)
        empty = torch.empty((rows_local, nx), dtype=torch.uint8, device=xd.device)
        h_inv = (C.c_double * 6)(*[float(v) for v in inv])
        is_max = 1 if bin_type == 'max' else 0
        _lib.check(self.lib.smrf_grid_clear_u64(p(keys), keys.numel(), self._st()))
        _lib.check(self.lib.smrf_grid_bin_f64(p(xd), p(yd), p(zd), xd.numel(), h_inv, None, p(keys), ny, nx, row0, rows_local,
                                              is_max, p(n_out), self._st()))
        _lib.check(self.lib.smrf_grid_finalize_f64(p(keys), p(grid), p(empty), keys.numel(), is_max, self._st()))
        return grid, empty, int(n_out.item())


def _a2a(dist, group, out, inp, out_splits, in_splits):
    """all_to_all_single; gloo stages CUDA tensors through the host (rehearsals on a one-GPU box)"""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        ho = out.cpu()
        dist.all_to_all_single(ho, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def create_dem_sharded(xd, yd, zd, cellsize=1, bin_type='max', *, rank=None, world_size=None, group=None, ops=None):
    """create_dem (neilpy/neilpy.py:1110-1166) with the POINTS sharded: every rank passes its own 1/N of the cloud
    and gets its row band of the raster - nothing is replicated (the all-to-all form).

    1. extents: local min/max (device reduction), a count of ranks with an undefined extent (all-reduce sum) and one
       4-double all-reduce(min) -> the same edges, shape and
       transform on every rank (:1117-1124, :1141);
    2. every point is routed to the rank whose band holds ``floor(row)`` of ``~t * (x, y)`` - bucketed on the device
       (count, prefix sum, pack), the counts and then the three coordinate runs exchanged with ``all_to_all_single``
       (RCCL; about 24 B per point cross the links once);
    3. the received points are binned into the band with the same 64-bit ``atomicMin`` as on one device (:1151-1156).
    Returns ``(band float64, empty uint8, transform, (ny, nx), (b0, b1))``.  Bit-identical to the rows
    ``b0:b1`` of the single-device raster: min / max do not depend on the order points arrive in.
    """
    import torch
    import torch.distributed as dist
    from . import api
    from .affine import from_origin
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if bin_type not in ('max', 'min'):
        raise ValueError('This type not supported.')                  # neilpy.py:1158
    if ops is None:
        ops = HipPointOps()
    multi = world_size > 1
    gloo = multi and dist.get_backend(group) == "gloo"
    xmin, xmax, ymin, ymax = ops.extent(xd, yd)
    # A rank whose extent is undefined (a NaN / inf coordinate, or no points) must stop EVERY rank.  MIN does not carry
    # a NaN reliably (RCCL's float min is fmin-like and drops it, gloo's std::min keeps it or not by operand order), so
    # the ranks agree on an explicit count of bad extents first and only reduce finite values.
    bad = int(not np.isfinite([xmin, xmax, ymin, ymax]).all())
    if multi:
        nb = torch.tensor([bad], dtype=torch.int32, device="cpu" if gloo else xd.device)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM, group=group)
        bad = int(nb.item())
    if bad:
        raise ValueError("zero-size or non-finite point set: the raster's extent is undefined")
    if multi:
        e = torch.tensor([xmin, ymin, -xmax, -ymax], dtype=torch.float64, device="cpu" if gloo else xd.device)
        dist.all_reduce(e, op=dist.ReduceOp.MIN, group=group)
        xmin, ymin, xmax, ymax = float(e[0]), float(e[1]), -float(e[2]), -float(e[3])
    xedges, yedges = api._edges_from_extent(np.float64(xmin), np.float64(xmax), np.float64(ymin), np.float64(ymax), cellsize)
    nx, ny = len(xedges) - 1, len(yedges) - 1
    t = from_origin(xedges[0], yedges[0], cellsize, cellsize)
    inv = tuple(~t)[:6]
    if multi and ny < world_size:
        raise ValueError("a raster of %d rows cannot be split over %d ranks" % (ny, world_size))
    b0, b1 = band_rows(ny, world_size, rank)
    if multi:
        counts, sx, sy, sz = ops.bucket(xd, yd, zd, inv, ny, world_size)
        cnt = counts.cpu() if (gloo or not counts.is_cuda) else counts
        got = torch.empty_like(cnt)
        dist.all_to_all_single(got, cnt, group=group)                  # how many points every rank sends me
        in_splits = [int(v) for v in counts.cpu().tolist()]
        out_splits = [int(v) for v in got.cpu().tolist()]
        n_in = sum(out_splits)
        rx, ry, rz = (torch.empty(n_in, dtype=torch.float64, device=xd.device) for _ in range(3))
        for dst_t, src_t in ((rx, sx), (ry, sy), (rz, sz)):
            _a2a(dist, group, dst_t, src_t, out_splits, in_splits)
        xd, yd, zd = rx, ry, rz
    band, empty, n_out = ops.bin_band(xd, yd, zd, inv, (ny, nx), b0, b1 - b0, bin_type)
    if multi:
        o = torch.tensor([n_out], dtype=torch.int64, device="cpu" if gloo else band.device)
        dist.all_reduce(o, op=dist.ReduceOp.SUM, group=group)
        n_out = int(o.item())
    if n_out > 0:
        raise ValueError("invalid entry in coordinates array")       # np.ravel_multi_index, neilpy.py:1151
    return band, empty, t, (ny, nx), (b0, b1)


# ------------------------------------------------------------------------------------------
# the whole smrf() over row bands
# ------------------------------------------------------------------------------------------
def _all_gather_rows(dist, group, part, sizes):
    """Concatenate every rank's leading-dimension block (``sizes[k]`` rows on rank k) on every rank.
    Blocks are padded to the largest one (all_gather wants equal shapes); gloo stages CUDA tensors
    through the host."""
    import torch
    world = len(sizes)
    if world == 1:
        return part
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
    buf[:part.shape[0]].copy_(part)
    staged = dist.get_backend(group) == "gloo" and part.is_cuda
    src = buf.cpu() if staged else buf
    outs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(outs, src, group=group)
    full = torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)
    return full.to(part.device) if staged else full


def smrf_sharded(x, y, z, cellsize=1, windows=5, slope_threshold=.15, elevation_threshold=.5, elevation_scaler=1.25,
                 low_filter_slope=5, low_outlier_fill=False, *, rank=None, world_size=None, group=None,
                 points="replicated"):
    """neilpy.smrf (neilpy/neilpy.py:1685-1808) with the raster split into row bands, one rank per GPU.

    ``points="replicated"``: every rank passes the FULL point set (or reads it from the same file) and bins it
    into its own rows (create_dem_band, no exchange).  ``points="sharded"``: every rank passes only its own part of
    the cloud; the points are routed to the rank that owns their row with one all-to-all (create_dem_sharded) and the
    tail classifies the rank's own points - nothing is replicated but the DTM the spline needs.
    Both spring inpaints and both progressive filters run on this rank's band
    (inpaint_nans_by_springs_sharded, progressive_filter_sharded).  The tail's
    bicubic spline is global along both axes, so the DTM bands are all-gathered and every rank
    solves the spline on the whole raster ("replicas only") but evaluates and tests only
    its own 1/N of the points; the point flags are all-gathered.

    Returns ``(dtm_band, transform, object_cells_band, is_object_point, (b0, b1))``: the rank's rows
    ``b0:b1`` of the DTM (float64 CUDA) and of the object raster (bool CUDA), and the point flags (bool CUDA): of
    ALL points, the same on every rank (replicated), or of the rank's own points (sharded).
    """
    import torch
    import torch.distributed as dist
    from . import api
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if np.isscalar(windows):
        windows = np.arange(windows) + 1
    windows = np.asarray(windows)
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)    # noqa: E731
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)              # noqa: E731
    if points not in ("replicated", "sharded"):
        raise ValueError("points must be 'replicated' or 'sharded'")
    xd, yd, zd = api._points_to_device(x, y, z)
    if points == "sharded":
        band, empty, t, (ny, nx), (b0, b1) = create_dem_sharded(xd, yd, zd, cellsize, 'min', rank=rank,
                                                                world_size=world_size, group=group)   # :1741
    else:
        xedges, yedges = api._dem_edges(xd, yd, cellsize)                          # :1117-1124, same on every rank
        nx, ny = len(xedges) - 1, len(yedges) - 1
        from .affine import from_origin
        t = from_origin(xedges[0], yedges[0], cellsize, cellsize)
        b0, b1 = band_rows(ny, world_size, rank)
        band, empty, n_out = create_dem_band(xd, yd, zd, tuple(~t)[:6], (ny, nx), rank=rank, world_size=world_size,
                                             bin_type='min')                       # :1741
        if n_out > 0:
            raise ValueError("invalid entry in coordinates array")
    stats = {"inpaint1": inpaint_nans_by_springs_sharded(band, ny, rank=rank, world_size=world_size, group=group)}
    neg = torch.empty_like(band)
    _lib.check(lib.smrf_negate_f64(p(band), p(neg), band.numel(), st()))
    low, _ = progressive_filter_sharded(neg, ny, np.array([1]), low_filter_slope * (np.array([1]) * cellsize),
                                        rank=rank, world_size=world_size, group=group)                 # :1744
    low = low.clone()
    del neg
    if low_outlier_fill:                                                           # :1747-1749
        _lib.check(lib.smrf_mask_apply_f64(p(band), p(low), None, None, None, band.numel(), st()))
        stats["inpaint1b"] = inpaint_nans_by_springs_sharded(band, ny, rank=rank, world_size=world_size, group=group)
    obj, _ = progressive_filter_sharded(band, ny, windows, slope_threshold * (windows * cellsize), rank=rank,
                                        world_size=world_size, group=group)        # :1752-1755
    obj = obj.contiguous()
    object_cells = torch.empty_like(obj)
    _lib.check(lib.smrf_mask_apply_f64(p(band), p(empty), p(low), p(obj), p(object_cells), band.numel(), st()))   # :1762-1763
    stats["inpaint2"] = inpaint_nans_by_springs_sharded(band, ny, rank=rank, world_size=world_size, group=group)  # :1764
    # tail: the spline needs the whole DTM; the points are split evenly
    sizes = [band_rows(ny, world_size, k)[1] - band_rows(ny, world_size, k)[0] for k in range(world_size)]
    Zpro = _all_gather_rows(dist, group, band, sizes)
    npts = xd.numel()
    if points == "sharded":                                  # every rank classifies its own points, nothing to gather
        if npts:
            flags = api._classify_points_device(Zpro, t, cellsize, xd, yd, zd, elevation_threshold, elevation_scaler)[2]
        else:
            flags = torch.empty(0, dtype=torch.uint8, device=xd.device)
        api.last_stats["sharded"] = stats
        return band, t, object_cells.bool(), flags.bool(), (b0, b1)
    psz = [band_rows(npts, world_size, k)[1] - band_rows(npts, world_size, k)[0] for k in range(world_size)]
    p0, p1 = band_rows(npts, world_size, rank)
    if p1 > p0:
        part = api._classify_points_device(Zpro, t, cellsize, xd[p0:p1].contiguous(), yd[p0:p1].contiguous(),
                                           zd[p0:p1].contiguous(), elevation_threshold, elevation_scaler)[2]
    else:
        part = torch.empty(0, dtype=torch.uint8, device=xd.device)
    flags = _all_gather_rows(dist, group, part, psz)
    api.last_stats["sharded"] = stats
    return band, t, object_cells.bool(), flags.bool(), (b0, b1)
