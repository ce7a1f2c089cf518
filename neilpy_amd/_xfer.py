"""Host <-> device copies of large arrays through two pinned staging buffers.

``torch.from_numpy(a).cuda()`` / ``t.cpu()`` on pageable memory run at 2-3 GB/s on the GPU boxes;
staging the same bytes through pinned memory in 16 MB chunks, with the host memcpy of one chunk
overlapping the DMA of the other, reaches PCIe rate.  Small arrays take the direct path.
The drop-in functions use this at their NumPy boundary (points in, rasters out); everything between
stays in HBM.
"""
import threading

import numpy as np

STAGE_BYTES = 16 << 20
SMALL = 4 << 20
_tls = threading.local()


def staging(device_index=None):
    """``(buffers, events)``: two pinned uint8 buffers of STAGE_BYTES and one event each, allocated once per
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
            pool[device_index] = ([torch.empty(STAGE_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                                  [torch.cuda.Event() for _ in range(2)])
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
    pos = k = 0
    while pos < src.size:
        n = min(STAGE_BYTES, src.size - pos)
        events[k % 2].synchronize()                          # the DMA that last read this buffer is done
        bufs[k % 2].numpy()[:n] = src[pos:pos + n]
        dst[pos:pos + n].copy_(bufs[k % 2][:n], non_blocking=True)
        events[k % 2].record()
        pos += n
        k += 1
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
    chunks = [(p, min(STAGE_BYTES, nbytes - p)) for p in range(0, nbytes, STAGE_BYTES)]
    for e in events:
        e.synchronize()

    def issue(k):
        p, n = chunks[k]
        bufs[k % 2][:n].copy_(src[p:p + n], non_blocking=True)
        events[k % 2].record()
    issue(0)
    for k, (p, n) in enumerate(chunks):
        if k + 1 < len(chunks):
            issue(k + 1)                                     # DMA of the next chunk runs during this memcpy
        events[k % 2].synchronize()
        dst[p:p + n] = bufs[k % 2].numpy()[:n]
    return out
