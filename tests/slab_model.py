"""numpy model of the library's slab-decomposed second-derivative pipeline
(pf_api.hip: KY layout -> x-pass 1->3 -> all-to-all -> y-pass 3->6 on the P
received blocks -> z-pass with kz factors, 1/N^3 and the DC mode), one process
per rank, exchange through torch.distributed.  TEST INFRASTRUCTURE: it pins the
decomposition algebra (block addressing, wavenumber offsets, which factor is
applied in which pass) on CPU with gloo; the HIP kernels follow the same maps."""
from __future__ import annotations

import numpy as np

PI = 3.14159265358979323846


def signed(n):
    i = np.arange(n)
    return np.where(i > n // 2, i - n, i).astype(np.float64) * (2.0 * PI / n)


def alltoall(dist, torch, send_blocks):
    """send_blocks[q] goes to rank q; returns the list received from every rank.
    gloo has no all_to_all: all_gather everything and pick (test sizes only)."""
    P = dist.get_world_size()
    r = dist.get_rank()
    mine = torch.from_numpy(np.ascontiguousarray(np.stack(send_blocks)))
    gathered = [torch.empty_like(mine) for _ in range(P)]
    dist.all_gather(gathered, mine)
    return [gathered[p][r].numpy() for p in range(P)]


def hess_band(n, rs, eps=2.0 ** -60):
    """pf_api.hip hess_band: modes beyond |k| = sqrt(-2 ln eps)/rs carry a window weight below eps and are dropped"""
    if rs <= 0.0:
        return 1 << 30
    kb = np.sqrt(-2.0 * np.log(eps)) / rs * n / (2.0 * PI)
    return int(kb) + 1 if kb < n // 2 - 1 else 1 << 30


def band_rows(n, P, p, band):
    """pf_api.hip band_rows: the in-band local ky rows [lo, hi) of rank p's slab (one interval for P >= 2)"""
    nyl = n // P
    y0, y1 = p * nyl, (p + 1) * nyl
    if y0 <= band and y1 > n - band:      # an odd number of ranks: the middle slab holds rows of both ends of the band -- one piece, the whole slab
        return 0, nyl
    if y0 <= band:
        return 0, min(band + 1, y1) - y0
    if y1 > n - band:
        return max(n - band, y0) - y0, nyl
    return 0, 0


def hessian_slab(dist, torch, dk_xslab, rs, prune=False):
    """dk_xslab: this rank's boundary slab [nxl][n][nzh] (x-slab, as kdensity[0]).
    Returns the six real fields of this rank's x-slab, order 11,22,33,12,13,23.
    prune: the band-limited form -- out-of-band kx are not read, only in-band (ky, kz) columns are transformed, and only
    the in-band rows x columns of every block go through the exchange (exchange_band / band_zpitch in pf_api.hip)."""
    P, r = dist.get_world_size(), dist.get_rank()
    nxl, n, nzh = dk_xslab.shape
    nyl = n // P
    k1 = signed(n)
    kz = (2.0 * PI / n) * np.arange(nzh)
    # boundary (x-slab) -> KY (y-slab of k-space): pf_set_density's to_blocks + exchange
    recv = alltoall(dist, torch, [dk_xslab[:, q * nyl:(q + 1) * nyl, :] for q in range(P)])
    dk_ky = np.concatenate(recv, axis=0)                      # [n (x)][nyl][nzh]
    dc = np.array([dk_xslab[0, 0, 0].real / n ** 3 if r == 0 else 0.0])
    t = torch.from_numpy(dc)
    dist.all_reduce(t)                                        # pf_set_density: DC mode to every rank
    dc = float(t[0])
    # x-pass (KY layout): Green prefactor with the GLOBAL ky of this y-slab (outer_offset = r*nyl)
    kx = k1[:, None, None]
    ky = k1[r * nyl:(r + 1) * nyl][None, :, None]
    k2 = kx ** 2 + ky ** 2 + kz[None, None, :] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        pre = np.where(k2 != 0.0, np.exp(-0.5 * k2 * rs * rs) / k2, 0.0)
    phi = dk_ky * pre
    band = hess_band(n, rs) if prune else 1 << 30
    if band < n // 2:
        sx = np.abs(np.where(np.arange(n) > n // 2, np.arange(n) - n, np.arange(n)))
        phi = phi * (sx <= band)[:, None, None]               # band_e: out-of-band kx never loaded
    A = [np.fft.ifft(phi * m, axis=0) * n for m in (1.0, kx, kx * kx)]
    # all-to-all of each field: send block q = x in slab q (contiguous in KY), receive [p][nxl][nyl][nzh]
    R = []
    for a in A:
        if band < n // 2:
            # only rows [lo, hi) of this rank and columns kz <= band travel; every rank knows everybody's interval
            lo, hi = band_rows(n, P, r, band)
            nb = min(band + 1, nzh)
            pad = np.zeros((nxl, nyl, nzh), dtype=a.dtype)    # gloo model: fixed-size messages, zeros stand for "not sent"
            send = []
            for q in range(P):
                blk = pad.copy()
                blk[:, lo:hi, :nb] = a[q * nxl:(q + 1) * nxl, lo:hi, :nb]
                send.append(blk)
            blocks = alltoall(dist, torch, send)
            for p_, blk in enumerate(blocks):                  # what was never sent is never read (y-pass band_e mask)
                plo, phi_ = band_rows(n, P, p_, band)
                keep = np.zeros((nyl,), dtype=bool)
                keep[plo:phi_] = True
                assert not np.any(blk[:, ~keep, :]) and not np.any(blk[:, :, nb:])
        else:
            blocks = alltoall(dist, torch, [a[q * nxl:(q + 1) * nxl] for q in range(P)])
        R.append(np.concatenate(blocks, axis=1))              # y = p*nyl + yl  -> [nxl][n][nzh]
    kyf = k1[None, :, None]
    iy = lambda f: np.fft.ifft(f, axis=1) * n                 # noqa: E731
    B = [iy(R[2]), iy(R[0] * kyf * kyf), iy(R[0]), iy(R[1] * kyf), iy(R[1]), iy(R[0] * kyf)]
    zmul = [1.0, 1.0, kz * kz, 1.0, kz, kz]
    out = []
    for b, m in zip(B, zmul):
        out.append(np.fft.irfft(b * m, n=n, axis=2) * n / n ** 3 + dc)
    return out
