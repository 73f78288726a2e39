"""Reader of tests/golden/lpt_analytic.json (closed-form LPT known answers, made by tests/golden/make_lpt_analytic.py):
turns the exact coefficient tables into the input half-spectrum and the expected fields on an n^3 grid.

A table is a list of [m, re, im]: f(x) = sum_m (re + i im) exp(2 pi i m.x / n).  The evaluation below is a plain Fourier
sum in float64 (a cosine table indexed by m.x mod n, so the phases are exact), cross-checked against the 40-digit samples
the generator stored.
"""
from __future__ import annotations

import json
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "lpt_analytic.json")
VEL_NAMES = ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2")


def load():
    with open(PATH) as f:
        return json.load(f)["cases"]


def case(name):
    for c in load():
        if c["name"] == name:
            return c
    raise KeyError(name)


def _f(s):
    return float(Fraction(s))


def evaluate(table, n, x0=0, nx=None):
    """the real field of a coefficient table on the x-slab [x0, x0+nx) of the n^3 grid, float64"""
    nx = n if nx is None else nx
    out = np.zeros((nx, n, n))
    j = np.arange(n)
    ct, st = np.cos(2 * np.pi * j / n), np.sin(2 * np.pi * j / n)
    # exact quadrant values
    for q, (cv, sv) in ((0, (1.0, 0.0)), (n // 4, (0.0, 1.0)), (n // 2, (-1.0, 0.0)), (3 * n // 4, (0.0, -1.0))):
        if n % 4 == 0 or q in (0, n // 2):
            ct[q], st[q] = cv, sv
    xs = np.arange(x0, x0 + nx)
    for m, re, im in table:
        ph = (m[0] * xs[:, None, None] + m[1] * j[None, :, None] + m[2] * j[None, None, :]) % n
        re, im = _f(re), _f(im)
        if re != 0.0:
            out += re * ct[ph]
        if im != 0.0:
            out -= im * st[ph]
    return out


def density_spectrum(c, n):
    """kdensity[0] of the case on the n^3 grid: [n][n][n/2+1] complex128 = n^3 c[m] (the reference's r2c is unnormalised,
    src/fmax-pfft.c:191-200); both members of a conjugate pair in the kz = 0 plane are stored"""
    assert n // 2 > 3 * 2, "third-order products of |m_i| <= 2 waves must stay below Nyquist"
    dk = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    for m, re, im in c["delta"]:
        if m[2] < 0:
            continue
        dk[m[0] % n, m[1] % n, m[2]] = (_f(re) + 1j * _f(im)) * float(n) ** 3
    return dk


def spectrum_of(table, n):
    """half-spectrum [n][n][n/2+1] of a coefficient table (n^3 c[m])"""
    s = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    for m, re, im in table:
        if m[2] < 0:
            continue
        s[m[0] % n, m[1] % n, m[2]] = (_f(re) + 1j * _f(im)) * float(n) ** 3
    return s


def growth(c):
    return np.array([_f(g) for g in c["growth"]])


def expected(c, n, x0=0, nx=None):
    """dict of expected float64 fields: d[6], s2, s3a, s3b, and Vel*[3] in grid units (N / 2 pi applied)"""
    e = dict(d=[evaluate(t, n, x0, nx) for t in c["d"]], s2=evaluate(c["s2"], n, x0, nx), s3a=evaluate(c["s3a"], n, x0, nx),
             s3b=evaluate(c["s3b"], n, x0, nx))
    fac = n / (2 * np.pi)
    for k in VEL_NAMES:
        e[k] = np.stack([fac * evaluate(t, n, x0, nx) for t in c["vel"][k]], axis=-1)
    return e


def check_sample(c, e):
    """the float64 evaluation against the generator's 40-digit samples (n = 16 only)"""
    s = c["sample"]
    cells = np.array(s["cells"])
    ix = (cells[:, 0], cells[:, 1], cells[:, 2])
    worst = 0.0
    for i in range(6):
        worst = max(worst, np.max(np.abs(e["d"][i][ix] - np.array(s["d"][i]))))
    for k in ("s2", "s3a", "s3b"):
        worst = max(worst, np.max(np.abs(e[k][ix] - np.array(s[k]))))
    for k in VEL_NAMES:
        for a in range(3):
            worst = max(worst, np.max(np.abs(e[k][..., a][ix] - np.array(s["vel"][k][a]))))
    return worst


def fp32_close(got_f32, want_f64, extra_abs=0.0):
    """a stored fp32 product against the exact value: within one fp32 ulp of it everywhere (the fp64 result feeding the
    conversion may sit a few 1e-16 from the exact value, which moves a tie), and the correctly rounded value on almost all
    cells.  Returns (max error in ulp, fraction of cells not equal to the correctly rounded value)."""
    want32 = want_f64.astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(want32), np.float32(1e-30))).astype(np.float64)
    err = np.abs(got_f32.astype(np.float64) - want_f64) - extra_abs
    off = (got_f32 != want32) & (np.abs(got_f32.astype(np.float64) - want32.astype(np.float64)) > extra_abs)  # (zero crossings aside)
    return float(np.max(err / ulp)), float(np.mean(off))


def scale_dependent_expected(c, n, x0=0, nx=None):
    """Vel* of a SCALE_DEPENDENT build: every mode of the unit-growth displacement tables times the multiplier of its |k|
    (generator's 40-digit values of InterpolateGrowth / GrowingMode*, src/cosmo.c:1728-1819, at k = 2 pi |m| / n rad/cell)"""
    sd = c["scale_dependent"]
    gk = sd["gk"][str(n)]
    fac = n / (2 * np.pi)
    out = {}
    for o, k in enumerate(VEL_NAMES):
        g_rat = Fraction(c["growth"][o])
        comps = []
        for t in c["vel"][k]:
            scaled = []
            for m, re, im in t:
                w = gk[o][str(m[0] ** 2 + m[1] ** 2 + m[2] ** 2)]
                scaled.append([m, Fraction(re) / g_rat * Fraction(w), Fraction(im) / g_rat * Fraction(w)])
            comps.append(fac * evaluate(scaled, n, x0, nx))
        out[k] = np.stack(comps, axis=-1)
    return out


def scale_dependent_tables(c):
    sd = c["scale_dependent"]
    return dict(T=[np.array([_f(t) for t in row]) for row in sd["T"]], logkmin=_f(sd["logkmin"]), dlogk=_f(sd["dlogk"]), sign=sd["sign"])
