"""Device scoping of the public entry points.

libsmrf_hip launches on the stream it is handed and keeps per-device launch geometry keyed by
``hipGetDevice()``; PyTorch's "current stream" is the current DEVICE's.  A CUDA tensor that lives on
another device than the current one would therefore be launched on with the wrong device's stream.
Every public function runs under ``torch.cuda.device(<device of its CUDA tensor arguments>)``;
tensors on two different devices in one call are refused.  NumPy-only calls use the current device.
"""
import functools

from ._lib import SmrfHipError


def is_tensor(a):
    return type(a).__module__.startswith("torch") and hasattr(a, "data_ptr")


def common_device(values):
    """The device shared by the CUDA tensors among ``values`` (None if there is none); raises
    :class:`SmrfHipError` when they live on different devices."""
    dev = None
    for v in values:
        if is_tensor(v) and getattr(v, "is_cuda", False):
            if dev is None:
                dev = v.device
            elif v.device != dev:
                raise SmrfHipError("arguments live on different devices (%s and %s): neilpy_amd runs one call on "
                                   "one device" % (dev, v.device))
    return dev


def device_scoped(fn):
    """Run ``fn`` with the device of its CUDA tensor arguments current (streams, allocations, launches)."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        dev = common_device(list(args) + list(kw.values()))
        if dev is None:
            return fn(*args, **kw)
        import torch
        if dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapper
