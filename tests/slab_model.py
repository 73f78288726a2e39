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


def hessian_slab(dist, torch, dk_xslab, rs):
    """dk_xslab: this rank's boundary slab [nxl][n][nzh] (x-slab, as kdensity[0]).
    Returns the six real fields of this rank's x-slab, order 11,22,33,12,13,23."""
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
    A = [np.fft.ifft(phi * m, axis=0) * n for m in (1.0, kx, kx * kx)]
    # all-to-all of each field: send block q = x in slab q (contiguous in KY), receive [p][nxl][nyl][nzh]
    R = []
    for a in A:
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
