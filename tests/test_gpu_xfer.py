"""The NumPy boundary's staged host <-> device copies (neilpy_amd/_xfer.py): every byte arrives, for sizes around the chunk
boundaries, from several threads at once, and next to the LAS reader that shares the pinned buffers."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nz():
    import neilpy_amd
    return neilpy_amd


@pytest.mark.parametrize("nbytes", [(4 << 20) + 16, (16 << 20), (16 << 20) + 32, 5 * (16 << 20), 7 * (16 << 20) + 4096 + 16])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.uint8])
def test_round_trip(nz, nbytes, dtype):
    import torch
    from neilpy_amd import _xfer
    rng = np.random.default_rng(nbytes % 1000)
    a = rng.integers(0, 255, size=nbytes, dtype=np.uint8).view(dtype)
    if dtype is not np.uint8:
        a = a.reshape(-1, 2)                                  # a 2-D shape survives the trip
    t = _xfer.to_device(a)
    assert t.is_cuda and tuple(t.shape) == a.shape
    assert np.array_equal(t.cpu().numpy().view(np.uint8), a.view(np.uint8))
    back = _xfer.to_host(t)
    assert back.dtype == a.dtype and back.shape == a.shape and np.array_equal(back.view(np.uint8), a.view(np.uint8))
    m = _xfer.to_host(t.view(torch.uint8).reshape(-1) > 127)  # bool masks leave through the same path
    assert m.dtype == np.bool_ and np.array_equal(m, a.view(np.uint8).reshape(-1) > 127)


def test_threads_do_not_share_staging(nz):
    from neilpy_amd import _xfer
    res, errs = {}, []

    def work(seed):
        try:
            rng = np.random.default_rng(seed)
            a = rng.random(6_000_000)                         # 48 MB: three chunks
            for _ in range(3):
                b = _xfer.to_host(_xfer.to_device(a))
                if not np.array_equal(a, b):
                    raise AssertionError("thread %d: bytes differ" % seed)
            res[seed] = True
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs and len(res) == 4, errs
