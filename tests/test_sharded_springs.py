"""Sharded spring inpaint (LSQR over row bands): halo / all-reduce protocol on CPU with gloo.

The band phases are provided by a NumPy port of the kernels (NumpySpringsOps below, tests only);
neilpy_amd.sharded.inpaint_nans_by_springs_sharded drives them exactly as it drives the HIP phases.
The sharded solve must stop at the reference's iteration and agree with the golden outputs.
"""
import os
import sys
from math import sqrt

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port, golden


class NumpySpringsOps:
    """CPU stand-in for HipSpringsOps: same planes (one halo row above and below), same phases."""

    def __init__(self, A_band, has_above, has_below):
        self.A = A_band                                   # torch float64 (rows, cols), filled in place
        self.rows, self.cols = A_band.shape
        self.ha, self.hb = bool(has_above), bool(has_below)
        n2 = (self.rows + 2, self.cols)
        self.x, self.w = np.zeros(n2), np.zeros(n2)
        self.v, self.uh, self.uv = torch.zeros(n2, dtype=torch.float64), np.zeros(n2), torch.zeros(n2, dtype=torch.float64)
        self.hole = torch.zeros(n2, dtype=torch.uint8)
        self.abelow = torch.zeros(self.cols, dtype=torch.float64)
        self.red2 = torch.zeros(2, dtype=torch.float64)   # [|v|^2, |dk|^2] of phase 7; the other phases use red[0]
        self.red = self.red2[:1]
        self.sc = {}

    def begin(self, atol, btol, conlim, iter_lim):
        self.sc = dict(atol=atol, btol=btol, ctol=1 / conlim if conlim > 0 else 0.0, iter_lim=iter_lim, itn=0, istop=0,
                       done=False, anorm=0.0, ddnorm=0.0, xxnorm=0.0, z=0.0, cs2=-1.0, sn2=0.0, alfa=0.0, beta=0.0,
                       ia=1.0, ib=1.0, beta_pos=False, nunk=0)

    def _stopped(self):
        return self.sc["done"] or self.sc["istop"] != 0

    def _aty(self):
        """S^T u_s on the own rows, the four springs of a cell in the kernels' order (up, left, right, down)"""
        s, n = self.sc, self.rows
        own = slice(1, n + 1)
        uv = self.uv.numpy()
        y = np.zeros((n, self.cols))
        up = s["ib"] * uv[0:n]
        if not self.ha:
            up[0] = 0.0
        y = y - up
        y[:, 1:] = y[:, 1:] - s["ib"] * self.uh[own][:, :-1]
        y[:, :-1] = y[:, :-1] + s["ib"] * self.uh[own][:, :-1]
        dn = s["ib"] * uv[own]
        if not self.hb:
            dn[-1] = 0.0
        return y + dn

    def phase(self, ph):
        s, n = self.sc, self.rows
        own = slice(1, n + 1)
        A = self.A.numpy()
        hole, v, uv = self.hole.numpy(), self.v.numpy(), self.uv.numpy()
        if ph == 0:
            hole[own] = np.isnan(A)
            self.red[0] = float(hole[own].sum())
        elif ph == 1:
            s["nunk"] = int(self.red[0])
            if s["iter_lim"] < 0:
                s["iter_lim"] = 2 * s["nunk"]
            if s["nunk"] == 0:
                s["done"] = True
            h = hole[own].astype(bool)
            K = np.where(h, 0.0, A)
            hb = hole[2:n + 2].astype(bool)                       # hole of the row below each own row
            Kb = np.vstack([K[1:], np.where(hb[-1:], 0.0, self.abelow.numpy()[None, :])])
            eh = np.zeros_like(K)
            act = h[:, :-1] | h[:, 1:]
            eh[:, :-1] = np.where(act, K[:, 1:] - K[:, :-1], 0.0)
            ev = np.where(h | hb, Kb - K, 0.0)
            if not self.hb:
                ev[-1] = 0.0
            self.uh[own], uv[own] = eh, ev
            self.x[:], self.w[:], v[:] = 0.0, 0.0, 0.0
            self.red[0] = float((eh * eh).sum() + (ev * ev).sum())
        elif ph == 2:
            b = sqrt(float(self.red[0]))
            s.update(bnorm=b, beta=b, beta_pos=b > 0, ib=1 / b if b > 0 else 1.0, alfa=0.0, ia=1.0)
        elif ph == 3:                                          # set-up: the first v = S^T u (v = 0 before)
            if self._stopped() or not s["beta_pos"]:
                return
            h = hole[own].astype(bool)
            nv = np.where(h, self._aty() - s["beta"] * (s["ia"] * v[own]), v[own])
            v[own] = nv
            self.red[0] = float((nv[h] * nv[h]).sum())
        elif ph == 4:
            if s["done"]:
                return
            a = sqrt(float(self.red[0])) if s["beta_pos"] else 0.0
            s.update(alfa=a, ia=1 / a if a > 0 else 1.0, rhobar=a, phibar=s["beta"])
            if a * s["beta"] == 0:
                s["done"] = True
        elif ph == 5:                                          # u = S v - alfa u
            if self._stopped():
                return
            h = hole[own].astype(bool)
            hb = hole[2:n + 2].astype(bool)
            vs = s["ia"] * v[own]
            vb = s["ia"] * v[2:n + 2]
            tot = 0.0
            act = h[:, :-1] | h[:, 1:]
            nu = (vs[:, :-1] - vs[:, 1:]) - s["alfa"] * (s["ib"] * self.uh[own][:, :-1])
            self.uh[own][:, :-1] = np.where(act, nu, self.uh[own][:, :-1])
            tot += float((nu[act] ** 2).sum())
            actv = h | hb
            if not self.hb:
                actv[-1] = False
            nuv = (vs - vb) - s["alfa"] * (s["ib"] * uv[own])
            uv[own] = np.where(actv, nuv, uv[own])
            tot += float((nuv[actv] ** 2).sum())
            self.red[0] = tot
        elif ph == 6:                                          # beta, then the part of the rotation that needs only rhobar and beta
            if self._stopped():
                return
            b = sqrt(float(self.red[0]))
            s.update(beta=b, beta_pos=b > 0)
            if b > 0:
                s["ib"] = 1 / b
                s["anorm"] = sqrt(s["anorm"] ** 2 + s["alfa"] ** 2 + b ** 2)
            else:
                s["ib"] = 1.0
            from oracle.smrf_oracle import _sym_ortho
            cs, sn, rho = _sym_ortho(s["rhobar"], s["beta"])
            phi = cs * s["phibar"]
            s.update(cs=cs, sn=sn, rho=rho, phi=phi, t1_prev=s.get("t1", 0.0), t1=phi / rho, ir=1 / rho)
        elif ph == 7:                                          # w_{k-1}, dk_k, (x every second iteration: both steps), v_k
            if self._stopped():
                return
            h = hole[own].astype(bool)
            vs = s["ia"] * v[own]
            wo = self.w[own].copy()
            first, xupd = s["itn"] == 0, (s["itn"] & 1) != 0
            wn = vs if first else vs + s["t2"] * wo
            if xupd:
                self.x[own] = np.where(h, (self.x[own] + s["t1_prev"] * wo) + s["t1"] * wn, self.x[own])
            self.w[own] = np.where(h, wn, self.w[own])
            dk = s["ir"] * wn
            self.red2[1] = float((dk[h] ** 2).sum())
            if s["beta_pos"]:
                nv = np.where(h, self._aty() - s["beta"] * vs, v[own])
                v[own] = nv
                self.red[0] = float((nv[h] * nv[h]).sum())
        elif ph == 8:                                          # alfa, the rest of the rotation, the stopping tests
            if self._stopped():
                return
            if s["beta_pos"]:
                a = sqrt(float(self.red[0]))
                s.update(alfa=a, ia=1 / a if a > 0 else 1.0)
            cs, sn, rho, phi = s["cs"], s["sn"], s["rho"], s["phi"]
            theta = sn * s["alfa"]
            s["rhobar"] = -cs * s["alfa"]
            s["phibar"] = sn * s["phibar"]
            s["tau"] = sn * phi
            s["t2"] = -theta / rho
            delta, gambar = s["sn2"] * rho, -s["cs2"] * rho
            rhs = phi - delta * s["z"]
            zbar = rhs / gambar
            s["xnorm"] = sqrt(s["xxnorm"] + zbar ** 2)
            gamma = sqrt(gambar ** 2 + theta ** 2)
            s.update(cs2=gambar / gamma, sn2=theta / gamma, z=rhs / gamma)
            s["xxnorm"] = s["xxnorm"] + s["z"] ** 2
            EPS = np.finfo(np.float64).eps
            nd = sqrt(float(self.red2[1]))                          # |dk|^2, all-reduced with |v|^2
            s["ddnorm"] += nd * nd
            s["itn"] += 1
            acond = s["anorm"] * sqrt(s["ddnorm"])
            rnorm = sqrt(s["phibar"] ** 2)
            arnorm = s["alfa"] * abs(s["tau"])
            test1 = rnorm / s["bnorm"]
            test2 = arnorm / (s["anorm"] * rnorm + EPS)
            test3 = 1 / (acond + EPS)
            t1 = test1 / (1 + s["anorm"] * s["xnorm"] / s["bnorm"])
            rtol = s["btol"] + s["atol"] * s["anorm"] * s["xnorm"] / s["bnorm"]
            istop = 0
            if s["itn"] >= s["iter_lim"]: istop = 7
            if 1 + test3 <= 1: istop = 6
            if 1 + test2 <= 1: istop = 5
            if 1 + t1 <= 1: istop = 4
            if test3 <= s["ctol"]: istop = 3
            if test2 <= s["atol"]: istop = 2
            if test1 <= rtol: istop = 1
            s["istop"] = istop
        elif ph == 10:                                         # a solve that stopped at an odd iteration owes x its last step
            h = hole[own].astype(bool)
            xs = self.x[own] + s["t1"] * self.w[own] if (s["itn"] & 1) else self.x[own]
            A[h] = xs[h]

    def status(self):
        s = self.sc
        return s["istop"], s["itn"], s["nunk"], bool(s["done"] or s["istop"] != 0 or s["itn"] >= s["iter_lim"] >= 0)


def _worker(rank, world, port, tag, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neilpy_amd import sharded
        g = golden("inpaint.npz")
        A = g[tag + "_in"]
        b0, b1 = sharded.band_rows(A.shape[0], world, rank)
        band = torch.from_numpy(np.ascontiguousarray(A[b0:b1]))
        ops = NumpySpringsOps(band, rank > 0, rank < world - 1)
        istop, itn, nunk = sharded.inpaint_nans_by_springs_sharded(band, A.shape[0], rank=rank, world_size=world, ops=ops)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), band=band.numpy(), b0=b0, b1=b1, istop=istop, itn=itn, nunk=nunk)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tag", [(1, "hole40"), (2, "occ10"), (2, "borders"), (3, "hole40"), (2, "allnan"),
                                       (2, "nonan"), (3, "occ60")])
def test_sharded_springs_equals_reference(tmp_path, world, tag):
    g = golden("inpaint.npz")
    want = g[tag + "_out"]
    istop, itn = (int(v) for v in g[tag + "_lsqr"])
    port = free_port()
    mp.spawn(_worker, args=(world, port, tag, str(tmp_path)), nprocs=world, join=True)
    got = np.empty_like(want)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        got[int(d["b0"]):int(d["b1"])] = d["band"]
        assert (int(d["istop"]), int(d["itn"])) == (istop, itn), (r, int(d["istop"]), int(d["itn"]))
        assert int(d["nunk"]) == int(np.isnan(g[tag + "_in"]).sum())
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8)
