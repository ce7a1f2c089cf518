"""Host <-> device copies of large arrays through pinned staging buffers.

``torch.from_numpy(a).cuda()`` / ``t.cpu()`` on pageable memory run at 2-3 GB/s on the GPU boxes;
staging the same bytes through pinned memory in 16 MB chunks overlaps the host memcpy of one chunk with
the DMA of another.  One thread's memcpy moves about 10 GB/s - less than the link - so the memcpys of up to
five chunks run side by side on a small worker pool (1.6 GB array on a GPU box: 22 -> 51 GB/s up, 8 -> 16 GB/s down, where
the fresh pages of the NumPy result are the limit; gpurun_out/r02/xfer_probe.log) (NumPy releases the GIL in a contiguous copy; the workers
touch host memory only, every HIP call stays on the caller's thread).  Small arrays take the direct path.
The drop-in functions use this at their NumPy boundary (points in, rasters out); everything between
stays in HBM.
"""
import collections
import threading

import numpy as np

STAGE_BYTES = 16 << 20
SMALL = 4 << 20
NBUF = 6                # staging buffers per (thread, device): one or two under DMA, the others being filled / drained
_tls = threading.local()
_pool = None
_pool_lock = threading.Lock()


def _workers():
    global _pool
    if _pool is None:
        with _pool_lock:
            if _pool is None:
                from concurrent.futures import ThreadPoolExecutor
                _pool = ThreadPoolExecutor(max_workers=NBUF - 1, thread_name_prefix="neilpy_amd_xfer")
    return _pool


def _copy(dst, src):
    dst[...] = src


def staging(device_index=None):
    """``(buffers, events)``: NBUF pinned uint8 buffers of STAGE_BYTES and one event each, allocated once per
    (thread, device): two threads transferring at once never share a staging chunk, and an event is only ever
    recorded on the device it belongs to."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    pool = getattr(_tls, "pool", None)
    if pool is None:
        pool = _tls.pool = {}
    if device_index not in pool:
        with torch.cuda.device(device_index):
            pool[device_index] = ([torch.empty(STAGE_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(NBUF)],
                                  [torch.cuda.Event() for _ in range(NBUF)])
    return pool[device_index]


def to_device(arr, device=None):
    """C-contiguous NumPy array -> CUDA tensor of the same dtype and shape."""
    import torch
    arr = np.ascontiguousarray(arr)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if arr.nbytes < SMALL:
        return torch.from_numpy(arr).to(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(index):                           # copies and events on the destination's streams
        return _to_device_staged(arr, torch.device("cuda", index))


def _to_device_staged(arr, device):
    import torch
    out = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0].reshape(-1)).dtype, device=device)
    src = arr.reshape(-1).view(np.uint8)
    dst = out.reshape(-1).view(torch.uint8)
    bufs, events = staging(device.index)
    views = [b.numpy() for b in bufs]
    pool = _workers()
    filling = collections.deque()                            # (future, slot, position, bytes), oldest first

    def send_oldest():
        fut, slot, p, n = filling.popleft()
        fut.result()                                         # the chunk is in its pinned buffer
        dst[p:p + n].copy_(bufs[slot][:n], non_blocking=True)
        events[slot].record()
    pos = k = 0
    while pos < src.size:
        n = min(STAGE_BYTES, src.size - pos)
        slot = k % NBUF
        if len(filling) == NBUF - 1:
            send_oldest()
        events[slot].synchronize()                           # the DMA that last read this buffer is done
        filling.append((pool.submit(_copy, views[slot][:n], src[pos:pos + n]), slot, pos, n))
        pos += n
        k += 1
    while filling:
        send_oldest()
    return out


def to_host(t):
    """CUDA tensor -> NumPy array of the same dtype and shape."""
    import torch
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes < SMALL or not t.is_cuda:
        return t.cpu().numpy()
    with torch.cuda.device(t.device):                        # the copies run on the tensor's device: so must the events
        return _to_host_staged(t, nbytes)


def _to_host_staged(t, nbytes):
    import torch
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    src = t.reshape(-1).view(torch.uint8)
    dst = out.reshape(-1).view(np.uint8)
    bufs, events = staging(t.device.index)
    views = [b.numpy() for b in bufs]
    pool = _workers()
    chunks = [(p, min(STAGE_BYTES, nbytes - p)) for p in range(0, nbytes, STAGE_BYTES)]
    for e in events:
        e.synchronize()
    draining = {}                                            # slot -> future of the memcpy out of that buffer
    ahead = 2                                                # chunks under DMA ahead of the one being drained
    issued = 0
    try:
        for k, (p, n) in enumerate(chunks):
            while issued < min(len(chunks), k + ahead):
                slot = issued % NBUF
                if slot in draining:
                    draining.pop(slot).result()              # the buffer's previous chunk has left it
                q, m = chunks[issued]
                bufs[slot][:m].copy_(src[q:q + m], non_blocking=True)
                events[slot].record()
                issued += 1
            slot = k % NBUF
            events[slot].synchronize()
            draining[slot] = pool.submit(_copy, dst[p:p + n], views[slot][:n])
    finally:
        for f in draining.values():
            f.result()
    return out
