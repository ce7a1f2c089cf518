import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

SAMPLES = ["samp11", "samp12", "samp21", "samp22", "samp23", "samp24", "samp31", "samp41",
           "samp42", "samp51", "samp52", "samp53", "samp54", "samp61", "samp71"]
SMALL_SAMPLES = ["samp21", "samp24", "samp11", "samp41"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


_samples = None


def load_sample(name):
    """(x, y, z, g) of an ISPRS sample, decoded from the lossless centi-unit fixture."""
    global _samples
    if _samples is None:
        _samples = golden("samples.npz")
    s = _samples
    return (s[name + "_x"] / 100.0, s[name + "_y"] / 100.0, s[name + "_z"] / 100.0, s[name + "_g"])


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].astype(bool).reshape(shape)


def zmin_from_centi(zc):
    return np.where(zc == -2 ** 31, np.nan, zc / 100.0)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")


def switch(monkeypatch, name, value):
    """set (or with None remove) one of the library's SMRF_* environment switches for the rest of the test and have the
    library read them again - it reads them once, at load (smrf_switches_reload)"""
    from neilpy_amd import _lib
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    _lib.reload_switches()


@pytest.fixture(autouse=True)
def _switches_follow_the_environment():
    """after monkeypatch has restored the environment (it is torn down before this autouse fixture), the library's cached
    switches are read again, so no test inherits another one's routing"""
    yield
    from neilpy_amd import _lib
    if _lib._lib is not None:
        _lib.reload_switches()


def free_port():
    """a TCP port nobody listens on right now (multi-process tests must not collide on a constant)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
