"""Independent numpy/pocketfft restatement of the hot-path formulas
(SURVEY.md section 2b / 8a / Appendix A).  TEST INFRASTRUCTURE: it pins the C
oracle's field-level path (own FFT, k-loop, LPT bookkeeping) against a second
implementation that shares no code with it.  SURVEY.md Appendix C.6 records
that exactly these formulas reproduce the running reference bit-for-bit in
Fmax/Rmax at N=32.
"""
from __future__ import annotations

import numpy as np

PI = 3.14159265358979323846


def kvecs(n):
    idx = np.arange(n)
    s = np.where(idx > n // 2, idx - n, idx).astype(np.float64)
    k1 = (2.0 * PI / n) * s
    kx = k1[:, None, None]
    ky = k1[None, :, None]
    kz = ((2.0 * PI / n) * np.arange(n // 2 + 1, dtype=np.float64))[None, None, :]
    return kx, ky, kz


def derivative(dk, rs, a, b, growth=1.0):
    """compute_derivative (src/fmax-pfft.c:255-441): a,b in {0,1,2,3};
    (a>0,b>0) second derivative, (a>0,b=0) first derivative (x i)."""
    n = dk.shape[0]
    kx, ky, kz = kvecs(n)
    comp = [np.ones((1, 1, 1)), kx, ky, kz]
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        g = comp[a] * comp[b] / k2 * np.exp(-0.5 * k2 * rs * rs) * growth
    g = np.where(k2 != 0.0, g, 1.0)
    c = dk * g
    if (a == 0) != (b == 0):
        c = c * 1j  # (re,im) <- (-im, re)
    return np.fft.irfftn(c, s=(n, n, n), axes=(0, 1, 2))


PAIRS = [(1, 1), (2, 2), (3, 3), (1, 2), (1, 3), (2, 3)]  # storage order, src/LPT.c:36-44


def hessian(dk, rs):
    return [derivative(dk, rs, a, b) for (a, b) in PAIRS]


def _spline_coeffs(x, y):
    # natural cubic spline second-derivative coefficients c_i (GSL cspline)
    n = len(x)
    h = np.diff(x)
    A = np.zeros((n, n))
    rhs = np.zeros(n)
    A[0, 0] = 1.0
    A[-1, -1] = 1.0
    for i in range(1, n - 1):
        A[i, i - 1] = h[i - 1]
        A[i, i] = 2.0 * (h[i - 1] + h[i])
        A[i, i + 1] = h[i]
        rhs[i] = 3.0 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
    return np.linalg.solve(A, rhs)


class Spline:
    def __init__(self, x, y):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.c = _spline_coeffs(self.x, self.y)

    def __call__(self, v):
        x, y, c = self.x, self.y, self.c
        v = np.asarray(v, dtype=np.float64)
        i = np.clip(np.searchsorted(x, v, side="right") - 1, 0, len(x) - 2)
        dx = x[i + 1] - x[i]
        dy = y[i + 1] - y[i]
        d = v - x[i]
        b = dy / dx - dx * (c[i + 1] + 2.0 * c[i]) / 3.0
        dd = (c[i + 1] - c[i]) / (3.0 * dx)
        inner = y[i] + d * (b + d * (c[i] + d * dd))
        lo = y[0] + (v - x[0]) * (y[1] - y[0]) / (x[1] - x[0])
        hi = y[-1] + (v - x[-1]) * (y[-1] - y[-2]) / (x[-1] - x[-2])
        return np.where(v < x[0], lo, np.where(v > x[-1], hi, inner))


def ell_classic(l1, l2, l3):
    """src/collapse_times.c:114-221, vectorised (generic branches only: the
    measure-zero |l1|<1e-20 / |den|<1e-20 ladders return -0.1 / are asserted absent)."""
    with np.errstate(all="ignore"):
        de = l1 + l2 + l3
        det = l1 * l2 * l3
        den = det / 126.0 + 5.0 * l1 * de * (de - l1) / 84.0
        rden = 1.0 / den
        a1 = 3.0 * l1 * (de - l1) / 14.0 * rden
        a1_2 = a1 * a1
        a2 = l1 * rden
        a3 = -1.0 * rden
        q = (a1_2 - 3.0 * a2) / 9.0
        r = (2.0 * a1_2 * a1 - 9.0 * a1 * a2 + 27.0 * a3) / 54.0
        disc = r * r - q * q * q
        # case 1
        fr = np.abs(r)
        sq1 = np.power(np.sqrt(np.where(disc > 0, disc, 0.0)) + fr, 0.333333333333333)
        e1 = -fr / r * (sq1 + q / sq1) - a1 / 3.0
        e1 = np.where(e1 < 0.0, -0.1, e1)
        # case 2
        sq = 2.0 * np.sqrt(q)
        t = np.arccos(2.0 * r / q / sq)
        s1 = -sq * np.cos(t * (1.0 / 3)) - a1 * (1.0 / 3)
        s2 = -sq * np.cos((t + 2.0 * PI) * (1.0 / 3)) - a1 * (1.0 / 3)
        s3 = -sq * np.cos((t + 4.0 * PI) * (1.0 / 3)) - a1 * (1.0 / 3)
        s1 = np.where(s1 < 0.0, 1e10, s1)
        s2 = np.where(s2 < 0.0, 1e10, s2)
        s3 = np.where(s3 < 0.0, 1e10, s3)
        e2 = np.where(s1 < s2, s1, s2)
        e2 = np.where(s3 < e2, s3, e2)
        e2 = np.where(e2 == 1e10, -0.1, e2)
        ell = np.where(disc > 0, e1, e2)
        ell = np.where(np.abs(l1) < 1e-20, -0.1, ell)
        assert not np.any((np.abs(den) < 1e-20) & (np.abs(l1) >= 1e-20)), "degenerate den branch hit"
        corr = (de > 0.0) & (ell > 0.0)
        inv = 1.0 / de
        ell = np.where(corr, ell - 0.364 * inv * np.exp(-6.5 * (l1 - l2) * inv - 2.8 * (l2 - l3) * inv), ell)
    return ell


def inverse_collapse_time(d, spline):
    """src/collapse_times.c:679-776 on arrays d[0..5] -> F (NaN where the
    reference's acos argument leaves [-1,1], quirk Q4)."""
    with np.errstate(all="ignore"):
        mu1 = d[0] + d[1] + d[2]
        mu1_2 = mu1 * mu1
        mu2 = 0.5 * mu1_2
        mu2 = mu2 - 0.5 * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        add0, add1, add2 = d[3] * d[3], d[4] * d[4], d[5] * d[5]
        mu2 = mu2 - (add0 + add1 + add2)
        mu3 = d[0] * d[1] * d[2] + 2.0 * d[3] * d[4] * d[5] - d[0] * add2 - d[1] * add1 - d[2] * add0
        q = (mu1_2 - 3.0 * mu2) / 9.0
        r = -(2.0 * mu1_2 * mu1 - 9.0 * mu1 * mu2 + 27.0 * mu3) / 54.0
        failm = (q * q * q < r * r) | (q < 0.0)
        sq = 2.0 * np.sqrt(q)
        t = np.arccos(2.0 * r / q / sq)
        x1 = -sq * np.cos(t * (1.0 / 3.0)) + mu1 * (1.0 / 3.0)
        x2 = -sq * np.cos((t + 2.0 * PI) * (1.0 / 3.0)) + mu1 * (1.0 / 3.0)
        x3 = -sq * np.cos((t + 4.0 * PI) * (1.0 / 3.0)) + mu1 * (1.0 / 3.0)
        diag = q == 0.0
        x1 = np.where(diag, d[0], x1)
        x2 = np.where(diag, d[1], x2)
        x3 = np.where(diag, d[2], x3)
        hi = np.maximum(np.maximum(x1, x2), x3)
        lo = np.minimum(np.minimum(x1, x2), x3)
        mid = x1 + x2 + x3 - lo - hi
        bc = ell_classic(hi, mid, lo)
        igm = 1.0 / np.power(10.0, spline(np.log10(np.where(bc > 0, bc, 1.0)))) - 1.0
        f = np.where(bc > 0.0, 1.0 + igm, 0.0)
        f = np.where(np.isnan(bc), np.nan, f)
        f = np.where(failm & ~diag, -10.0, f)
    return f


def sweep(dk, radii_cells, spline):
    """compute_fmax loop (src/fmax.c:66-150): float32 running max with the
    promoted-float comparison (quirk Q2)."""
    n = dk.shape[0]
    fmax = np.full((n, n, n), -10.0, dtype=np.float32)
    rmax = np.full((n, n, n), -1, dtype=np.int32)
    tv = []
    hes = None
    for i, rs in enumerate(radii_cells):
        hes = hessian(dk, rs)
        delta = hes[0] + hes[1] + hes[2]
        tv.append(np.sum(delta * delta) / n ** 3)
        f = inverse_collapse_time(hes, spline)
        upd = fmax.astype(np.float64) < f
        fmax = np.where(upd, f.astype(np.float32), fmax)
        rmax = np.where(upd, np.int32(i), rmax)
    return fmax, rmax, np.array(tv), hes


def lpt(dk, hes, growth):
    """src/LPT.c:32-235 + compute_displacements (src/fmax.c:292-367).
    growth = [D, D2, D31 (signed), D32].  Returns dict of 4 x [3] float32 fields."""
    n = dk.shape[0]
    s0, s1, s2, s3, s4, s5 = hes
    src2 = s0 * s1 + s0 * s2 + s1 * s2 - s3 * s3 - s4 * s4 - s5 * s5
    src31 = 3.0 * (s0 * (s1 * s2 - s5 * s5) - s3 * (s3 * s2 - s4 * s5) + s4 * (s3 * s5 - s4 * s1))
    src32 = 2.0 * (s0 + s1 + s2) * src2
    k2lpt = np.fft.rfftn(src2, axes=(0, 1, 2))
    for i, (a, b) in enumerate(PAIRS):
        ider = a if a == b else a + b + 1
        phi2 = derivative(k2lpt, 0.0, a, b)
        src32 = src32 - 2.0 * (1.0 if ider <= 3 else 2.0) * phi2 * hes[ider - 1]
    k31 = np.fft.rfftn(src31, axes=(0, 1, 2))
    k32 = np.fft.rfftn(src32, axes=(0, 1, 2))
    out = {}
    for name, spec, g in (("Vel", dk, growth[0]), ("Vel_2LPT", k2lpt, growth[1]),
                          ("Vel_3LPT_1", k31, growth[2]), ("Vel_3LPT_2", k32, growth[3])):
        out[name] = np.stack([derivative(spec, 0.0, a, 0, g).astype(np.float32) for a in (1, 2, 3)], axis=-1)
    out["_k"] = (k2lpt, k31, k32)
    return out
