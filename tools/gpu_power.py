#!/usr/bin/env python3
"""Package power, power cap and shader clock of one GPU from the amdgpu hwmon files (measurement helper; bench.py's
``roofline.power`` and tools/experiments/power_probe.py).  The files are world-readable on the GPU box:
``/sys/bus/pci/devices/<bdf>/hwmon/hwmon*/power1_input`` (uW, the socket's current package power), ``power1_cap`` (uW) and
``freq1_input`` (Hz, label sclk).  ``Sampler`` reads them from a thread every ``period`` seconds between start() and stop()."""
import glob
import os
import threading
import time


def hwmon_dir(device_index=0):
    """the hwmon directory of torch's cuda:<device_index>, or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:  # noqa: BLE001
        return None
    found = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
    return found[0] if found else None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:  # noqa: BLE001
        return None


class Sampler:
    def __init__(self, device_index=0, period=0.01):
        self.dir = hwmon_dir(device_index)
        self.period = period
        self.samples = []                                  # (time, watts, sclk GHz)
        self._stop = threading.Event()
        self._th = None

    @property
    def ok(self):
        return self.dir is not None and read_int(os.path.join(self.dir, "power1_input")) is not None

    def cap_w(self):
        v = read_int(os.path.join(self.dir, "power1_cap")) if self.dir else None
        return None if v is None else v / 1e6

    def _run(self):
        pw, fq = os.path.join(self.dir, "power1_input"), os.path.join(self.dir, "freq1_input")
        while not self._stop.is_set():
            w, f = read_int(pw), read_int(fq)
            if w is not None:
                self.samples.append((time.time(), w / 1e6, None if f is None else f / 1e9))
            time.sleep(self.period)

    def start(self):
        if self.ok:
            self._stop.clear()
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def stop(self):
        if self._th is not None:
            self._stop.set()
            self._th.join()
            self._th = None
        return self

    def between(self, t0, t1):
        """median / max package power and median shader clock of the samples taken in [t0, t1]"""
        import numpy as np
        sel = [s for s in self.samples if t0 <= s[0] <= t1]
        if not sel:
            return None
        w = np.array([s[1] for s in sel])
        f = np.array([s[2] for s in sel if s[2] is not None])
        return {"socket_w": float(np.median(w)), "socket_w_max": float(w.max()), "sclk_ghz": float(np.median(f)) if len(f) else None,
                "samples": len(sel)}
