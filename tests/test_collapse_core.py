"""Unit test of the device collapse header (pinocchio_amd/csrc/pf_collapse_core.h)
compiled for the host, against the reference known answers and the oracle."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from pinocchio_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_emul", "collapse_emul.cpp")
SO = os.path.join(HERE, "cpu_emul", "libcollapse_emul.so")
dp = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def emul():
    hdrs = [os.path.join(HERE, "..", "pinocchio_amd", "csrc", h) for h in ("pf_collapse_core.h", "pf_sng_core.h")]
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max([os.path.getmtime(SRC)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.emul_collapse.argtypes = [dp, dp, C.c_int, dp, C.c_long, dp, dp]
    L.emul_ell_classic.restype = C.c_double
    L.emul_ell_classic.argtypes = [C.c_double] * 3
    L.emul_spline.argtypes = [dp, dp, C.c_int, dp, C.c_long, dp]
    return L


def test_kat(emul):
    with open(os.path.join(HERE, "golden", "collapse_kat.json")) as f:
        kat = json.load(f)
    for case in kat["ell_classic"]:
        assert emul.emul_ell_classic(*case["l"]) == pytest.approx(case["bc"], rel=4e-16, abs=0)
    x, y = synth.invgrow_table("eds")
    d = np.array([c["d"] for c in kat["inverse_collapse_time"]], dtype=np.float64)
    F = np.empty(len(d)); lam = np.empty((len(d), 3))
    assert emul.emul_collapse(_dp(x), _dp(y), len(x), _dp(d), len(d), _dp(F), _dp(lam)) == 0
    for i, case in enumerate(kat["inverse_collapse_time"]):
        assert F[i] == pytest.approx(case["F"], rel=1e-14, abs=1e-15)


def test_random_hessians_match_oracle_bitwise(emul):
    """same libm, same operation order -> identical doubles, including sentinels and NaNs"""
    rng = np.random.default_rng(5)
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(8, 1)
    o.set_invgrow(x, y)
    n = 20000
    d = rng.standard_normal((n, 6)) * np.array([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
    d[:50, 3:] = 0.0            # diagonal tensors
    d[50:60] = 0.0              # all-zero
    d[60:70, :3] = 0.4; d[60:70, 3:] = 0.0   # q == 0 branch
    F = np.empty(n); lam = np.empty((n, 3))
    assert emul.emul_collapse(_dp(x), _dp(y), len(x), _dp(d), n, _dp(F), _dp(lam)) == 0
    want = np.array([o.inverse_collapse_time(row)[0] for row in d])
    same = (F == want) | (np.isnan(F) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]
    assert (F > 1).sum() > 100 and (F == 0).sum() > 100


def test_spline(emul):
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(8, 1)
    o.set_invgrow(x, y)
    v = np.concatenate([np.linspace(x[0] - 1, x[-1] + 1, 3001), x])
    out = np.empty_like(v)
    assert emul.emul_spline(_dp(x), _dp(y), len(x), _dp(v), len(v), _dp(out)) == 0
    want = np.array([o.spline_eval(t) for t in v])
    assert np.array_equal(out, want)


@pytest.mark.parametrize("with_table", [1, 0])
def test_fast_flavour_agrees_to_ulps(emul, with_table):
    """sincos-rotation / cbrt / exp10 forms of the three hot spots -- and the inverse growing mode from the polynomial table
    of the spline (pf_gtab.h: what the kernels run), or from the series forms --: same values to ~1e-15 where the
    cubic is well conditioned, identical sentinels and zeros"""
    emul.emul_collapse_fast.argtypes = [dp, dp, C.c_int, dp, C.c_long, dp, dp, C.c_int]
    rng = np.random.default_rng(6)
    x, y = synth.invgrow_table("lcdm")
    n = 200000
    d = rng.standard_normal((n, 6)) * np.array([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
    d[:50, 3:] = 0.0
    d[50:60] = 0.0
    F0 = np.empty(n); F1 = np.empty(n); lam = np.empty((n, 3))
    assert emul.emul_collapse(_dp(x), _dp(y), len(x), _dp(d), n, _dp(F0), _dp(lam)) == 0
    assert emul.emul_collapse_fast(_dp(x), _dp(y), len(x), _dp(d), n, _dp(F1), _dp(lam), with_table) == (2 if with_table else 0)
    nan = np.isnan(F0) & np.isnan(F1)
    assert np.array_equal(F0 == -10.0, F1 == -10.0) and np.array_equal(F0 == 0.0, F1 == 0.0)
    rel = np.abs(F1 - F0) / np.maximum(1.0, np.abs(F0))
    rel[nan] = 0
    print("fast vs exact: median %.2e  99.9%% %.2e  max %.2e  bitwise equal %.3f" % (
        np.median(rel), np.quantile(rel, 0.999), rel.max(), np.mean(F0 == F1)))
    # ulp-level differences of the calls, amplified where the cubic is ill-conditioned (same statistics as the
    # device libm against glibc)
    assert np.median(rel) < 5e-15
    assert np.quantile(rel, 0.999) < 1e-10
    assert rel.max() < 1e-5
    # stored value: fp32
    assert np.mean(F0.astype(np.float32) != F1.astype(np.float32)) < 2e-4


@pytest.mark.parametrize("kind", ["lcdm", "eds", "wiggly", "dense"])
def test_inverse_growth_table_of_the_fast_flavour(emul, kind):
    """pf_gtab.h: 10^(-S(log10 D)) as piecewise degree-7 polynomials in D -- against the composite evaluated by mpmath from the
    same knots and cspline coefficients (50 digits): within 2e-14 everywhere inside the table, 3e-15 on the knot intervals (the
    reference's own chain log10 -> gsl spline -> pow carries ~4e-15 of rounding), NaN (= "take the series forms") outside; knots too dense for the
    start table are refused"""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    emul.emul_gtab.argtypes = [dp, dp, C.c_int, dp, C.c_long, dp, dp, dp]
    x, y = synth.invgrow_table("eds" if kind == "eds" else "lcdm")
    if kind == "wiggly":     # a table whose natural spline oscillates at its ends: intervals are split where the fit asks for it
        y = y + 1e-4 * np.sin(40.0 * x)
    if kind == "dense":      # ten times the knots: more than one bin of the start table can hold
        xx = np.linspace(x[0], x[-1], 500)
        y = np.interp(xx, x, y); x = xx
    n = len(x)
    rng = np.random.default_rng(12)
    D = 10.0 ** rng.uniform(x[0] - 0.3, 3.3, 20000)
    D[:4] = [10.0 ** x[0], np.nextafter(10.0 ** x[0], 0), 1023.0, 1e4]
    Y = np.empty(len(D)); info = np.empty(8); bcd = np.empty(3 * n)
    rc = emul.emul_gtab(_dp(x), _dp(y), n, _dp(D), len(D), _dp(Y), _dp(info), _dp(bcd))
    if kind == "dense":
        assert rc == 1 and info[6] == 0.0 and np.isnan(Y).all()
        return
    assert rc == 0 and info[6] == 1.0 and info[5] <= 2e-14, info
    lo, hi = info[3], info[4]
    assert lo == pytest.approx(10.0 ** x[0], rel=1e-15) and hi >= 1024.0
    inside = (D >= lo) & (D < hi)
    assert np.array_equal(np.isnan(Y), ~inside) and inside.sum() > 15000
    c, b, d = bcd[:n], bcd[n:2 * n], bcd[2 * n:]
    slope = (mp.mpf(y[-1]) - mp.mpf(y[-2])) / (mp.mpf(x[-1]) - mp.mpf(x[-2]))
    worst = worst_knots = 0.0
    for Di, Yi in zip(D[inside][:6000], Y[inside][:6000]):
        xl = mp.log10(mp.mpf(float(Di)))
        if xl > x[-1]:
            S = mp.mpf(y[-1]) + (xl - mp.mpf(x[-1])) * slope
        else:
            j = min(max(int(np.searchsorted(x, float(xl), side="right")) - 1, 0), n - 2)
            dx = xl - mp.mpf(x[j])
            S = mp.mpf(y[j]) + dx * (mp.mpf(b[j]) + dx * (mp.mpf(c[j]) + dx * mp.mpf(d[j])))
        want = mp.power(10, -S)
        err = float(abs((mp.mpf(float(Yi)) - want) / want))
        worst = max(worst, err)
        if xl <= x[-1]:
            worst_knots = max(worst_knots, err)
    assert worst <= 2e-14 and worst_knots <= 3e-15, (worst, worst_knots)


def test_tabulated_ct_header_matches_oracle_bitwise(emul):
    """delta sampling, node splines (shared LDL^t factors + GSL's substitutions) and the bilinear-of-splines lookup of
    the device header on the host against the oracle's restatement of src/collapse_times.c:780-1231: identical doubles"""
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(8, 2)
    o.set_invgrow(x, y)
    var = 2.3
    tab, dv = o.ct_build(0, var)
    rng = np.random.default_rng(11)
    n = 5000
    ampl = np.sqrt(var)
    d = rng.uniform(-9.0, 9.0, n)              # beyond +-7: my_spline_eval's linear extrapolation
    xx = rng.uniform(0.0, 4.0, n)              # beyond 3.5: index clamp
    yy = rng.uniform(0.0, 4.0, n)
    lam = np.stack([(d + 2 * xx + yy) / 3.0 * ampl, (d - xx + yy) / 3.0 * ampl, (d - xx - 2 * yy) / 3.0 * ampl], axis=1).copy()
    F = np.empty(n); delta = np.empty(100)
    emul.emul_ct.argtypes = [dp, C.c_double, dp, C.c_long, dp, dp, C.c_int]
    # 0: BILINEAR_SPLINE, the source's define; 1, 2: the -DTRILINEAR and -DALL_SPLINE flavours (:1153-1216; the 4 x 4 natural
    # splines and the bicubic patch of gsl_spline2d in the header's closed form against the oracle's general-size restatement)
    for flavour in (0, 1, 2):
        assert emul.emul_ct(_dp(np.ascontiguousarray(tab)), ampl, _dp(lam), n, _dp(delta), _dp(F), flavour) == 0
        assert np.array_equal(delta, dv)
        o.set_ct_interpolation(flavour)
        want = np.array([o.interpolate_collapse_time(*l) for l in lam])
        assert np.array_equal(F, want), (flavour, np.abs(F - want).max())


def test_ell_sng_header_matches_oracle_bitwise(emul):
    """the RKF45 / step-control / accept-reject loop of pf_sng_core.h on the host against oracle/pf_sng.c (written
    separately from the same GSL algorithm): identical collapse epochs, including 'no collapse' and the sentinel"""
    rng = np.random.default_rng(21)
    n = 400
    d = rng.uniform(-3.0, 6.0, n); x = rng.uniform(0.0, 3.0, n); y = rng.uniform(0.0, 3.0, n)
    lam = np.stack([(d + 2 * x + y) / 3.0, (d - x + y) / 3.0, (d - x - 2 * y) / 3.0], axis=1).copy()
    lam[:5] = [[1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [2.0, 2.0, -1.0], [-1.0, -1.0, -1.0], [1e-3, 1e-3, 1e-3]]
    L = oracle_lib.lib()
    emul.emul_ell_sng.argtypes = [dp, C.c_long, C.c_double, dp, dp]
    hoc = 100.0 / 299792.458
    # Omega0, OmegaLambda, OmegaRad, OmegaK, FR0, H_over_c, size: the last case is MOD_GRAV_FR (|f_R0| = 1e-5, R = 2 Mpc)
    for cosmo, din in ((np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]), 1e-5), (np.array([0.25, 0.75, 0.0, 0.0, 0.0, 0.0, 0.0]), 1.28e-5),
                       (np.array([0.3, 0.6, 0.0, 0.1, 0.0, 0.0, 0.0]), 1.2e-5), (np.array([0.25, 0.75, 0.0, 0.0, 1e-5, hoc, 2.0]), 1.28e-5)):
        got = np.empty(n)
        emul.emul_ell_sng(_dp(lam), n, din, _dp(cosmo), _dp(got))
        want = np.array([L.orc_ell_sng(l[0], l[1], l[2], din, _dp(cosmo)) for l in lam])
        assert np.array_equal(got, want)
        assert (want > 0).sum() > 100 and (want == 0).sum() > 20


def test_spline_interval_table_finds_the_bisection_interval(emul):
    """the start-index table of k_collapse's spline lookup against plain bisection: identical values at the knots, next to
    them (one ulp either side), at the bin edges of both table geometries and at random abscissae.  Knot sets: the growth
    table (direct form: bins of width 2^k, no loop), knots ON power-of-two multiples (direct form, with knots exactly on bin
    edges: the sliver below an edge where the bin index comes out one too high), spacings over 24 orders of magnitude (two
    knots in a bin: the walk form), and a three-knot spline"""
    emul.emul_spline_lut.argtypes = [dp, dp, C.c_int, dp, C.c_long, dp]
    rng = np.random.default_rng(2)
    x1, y1 = synth.invgrow_table("lcdm")
    x2 = np.cumsum(np.concatenate([[0.0], rng.uniform(1e-6, 1.0, 209) ** 4])) - 3.0   # spacings over 24 orders of magnitude
    y2 = np.sin(x2)
    x3 = -4.0 + 0.03125 * np.arange(210)                                              # knots on multiples of 2^-5
    y3 = np.cos(x3)
    for x, y, form in ((x1, y1, 2), (x2, y2, 0), (x3, y3, 2), (x1[:3], y1[:3], 2)):
        n = len(x)
        edges = x[0] + (x[-1] - x[0]) / 4096.0 * np.arange(4097)
        w = 2.0 ** np.ceil(np.log2((x[-1] - x[0]) / 4094.0) + 1e-12)
        e2 = np.floor(x[0] / w) * w + w * np.arange(4097)                              # the direct form's bin edges
        e2 = e2[(e2 >= x[0]) & (e2 <= x[-1])]
        v = np.concatenate([x, np.nextafter(x, -np.inf), np.nextafter(x, np.inf), edges, np.nextafter(edges, -np.inf),
                            e2, np.nextafter(e2, -np.inf), np.nextafter(e2, np.inf), rng.uniform(x[0] - 0.5, x[-1] + 0.5, 20000)])
        a = np.empty(len(v)); b = np.empty(len(v))
        assert emul.emul_spline(_dp(x), _dp(y), n, _dp(v), len(v), _dp(a)) == 0
        assert emul.emul_spline_lut(_dp(x), _dp(y), n, _dp(v), len(v), _dp(b)) == form
        assert np.array_equal(a, b)


def test_division_by_constant_is_correctly_rounded(emul):
    """pf_div_const (multiply + two fma) against the IEEE division it stands in for: identical doubles"""
    emul.emul_div_const.argtypes = [dp, C.c_long, dp, dp]
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.integers(-30, 30, 200000), [0.0, -0.0, 9.0, 54.0, 1e-310, 1.7e308],
                        np.arange(1.0, 2000.0)])
    a = np.empty(len(x)); b = np.empty(len(x))
    emul.emul_div_const(_dp(x), len(x), _dp(a), _dp(b))
    keep = np.abs(x) > 1e-290            # the residual trick needs x c and the residual to stay normal
    assert np.array_equal(a[keep], (x / 9.0)[keep]) and np.array_equal(b[keep], (x / 54.0)[keep])
    assert np.allclose(a[~keep], (x / 9.0)[~keep], rtol=1e-15, atol=1e-320)


def test_log10_of_the_fast_flavour(emul):
    """pf_log10_pos against the correctly rounded value (numpy longdouble): within 2 ulp over the whole positive range,
    relative to log10 itself also next to x = 1 where the result passes through zero"""
    emul.emul_log10.argtypes = [dp, C.c_long, dp]
    rng = np.random.default_rng(4)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 200000), 10.0 ** rng.uniform(-5, 2, 400000), 1.0 + rng.uniform(-1e-3, 1e-3, 100000),
                        1.0 + rng.uniform(-1e-9, 1e-9, 1000), [1.0, 2.0, 0.5, 10.0, 0.1, 0.70710678118654752, 1.4142135623730951, 1e-5, 5.0]])
    got = np.empty(len(x))
    emul.emul_log10(_dp(x), len(x), _dp(got))
    want = (np.log(x.astype(np.longdouble)) / np.log(np.longdouble(10))).astype(np.float64)
    ulp = np.spacing(np.abs(want))
    bad = np.abs(got - want) > 2 * ulp
    assert not bad.any(), (x[bad][:5], got[bad][:5], want[bad][:5])
    assert got[len(x) - 9] == 0.0          # log10(1)


@pytest.mark.parametrize("form", ["emul_cos3", "emul_cos3_tab"])
def test_cosine_triple_of_the_fast_flavour(emul, form):
    """pf_cos3_of_acos(x) = cos((acos x + 2 pi k) / 3), k = 0, 1, 2, without acos and sincos (one polynomial, two square roots;
    emul_cos3_tab: the polynomial from the table of 32 short ones, pf_c3tab.h, the form the cell kernels run)
    against mpmath (40 digits): c_1 within 2 ulp everywhere on [-1, 1]; c_2, c_3 within 2 ulp of their own magnitude scale (they
    are -c_1/2 -+ sqrt(3)/2 sin: absolute error a few 1e-16) where the triple is well conditioned, and -- next to x = 1, where
    1 - x carries the rounding of x itself -- within the change a 1-ulp change of x makes; NaN in all three outside [-1, 1]"""
    import mpmath as mp
    mp.mp.dps = 40
    cos3 = getattr(emul, form)
    cos3.argtypes = [dp, C.c_long, dp]
    rng = np.random.default_rng(33)
    x = np.concatenate([rng.uniform(-1, 1, 20000), 1.0 - 10.0 ** rng.uniform(-16, 0, 5000), -1.0 + 10.0 ** rng.uniform(-16, 0, 5000),
                        10.0 ** rng.uniform(-300, -1, 1000), -10.0 ** rng.uniform(-300, -1, 1000),
                        [0.0, 1.0, -1.0, 0.5, -0.5, np.nextafter(1.0, 0.0), np.nextafter(-1.0, 0.0)]])
    edges = 2.0 * (np.arange(33) / 32.0) ** 2 - 1.0   # the ends of the table's pieces (y = j / 32), and their neighbours
    x = np.concatenate([x, edges, np.nextafter(edges, 2.0), np.nextafter(edges, -2.0), np.nextafter(np.nextafter(edges, 2.0), 2.0)])
    x = np.clip(x, -1.0, 1.0)
    got = np.empty((len(x), 3))
    cos3(_dp(x), len(x), _dp(got))
    want = np.empty((len(x), 3))
    for i, v in enumerate(x):
        t = mp.acos(mp.mpf(float(v)))
        want[i] = [float(mp.cos((t + 2 * mp.pi * k) / 3)) for k in range(3)]
    eps = np.finfo(float).eps
    # conditioning: d c_k / d x = sin((t + 2 pi k)/3) / (3 sin t): a 1-ulp change of x moves c_k by this much
    t = np.arccos(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        cond = np.abs(np.sin((t[:, None] + 2 * np.pi * np.arange(3)) / 3)) / (3 * np.maximum(np.sin(t), 1e-300))[:, None] * np.spacing(np.abs(x))[:, None]
    tol = 2.5 * eps + 2.0 * np.where(np.isfinite(cond), cond, 0.0)
    err = np.abs(got - want)
    bad = err > tol
    assert not bad.any(), (x[bad.any(axis=1)][:5], got[bad.any(axis=1)][:5], want[bad.any(axis=1)][:5])
    assert (np.abs(got[:, 0] - want[:, 0]) <= 2 * np.spacing(want[:, 0])).all()           # the largest root: always well conditioned
    # the roots really solve the Chebyshev cubic, and they come out ordered c1 >= c3 >= c2
    res = 4 * got ** 3 - 3 * got - x[:, None]
    assert np.max(np.abs(res)) < 4e-15    # (the residual itself is evaluated in double: |f'| <= 9)
    assert (got[:, 0] >= got[:, 2]).all() and (got[:, 2] >= got[:, 1]).all()
    out = np.empty(9)
    cos3(_dp(np.array([1.0000001, -1.5, np.nan])), 3, _dp(out))
    assert np.all(np.isnan(out))


def test_exp_and_exp10_of_the_fast_flavour(emul):
    """pf_exp_series / pf_exp10_series against the correctly rounded values (mpmath, 40 digits): within 1 ulp over the
    ranges the solver uses and over the whole finite range; exact at 0 and at integer powers of ten; inf / 0 beyond"""
    import mpmath as mp
    mp.mp.dps = 40
    emul.emul_exp.argtypes = [dp, C.c_long, dp, dp]
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(-40, 0.5, 20000), rng.uniform(-3.5, 3.5, 20000), rng.uniform(-300, 300, 5000),
                        rng.uniform(-1e-3, 1e-3, 2000), 10.0 ** rng.uniform(-300, -3, 1000), -10.0 ** rng.uniform(-300, -3, 1000),
                        0.5 * np.log(2.0) * (2 * np.arange(-20, 21) + 1), [0.0, -0.0, 1.0, -1.0]])
    e = np.empty(len(x)); e10 = np.empty(len(x))
    emul.emul_exp(_dp(x), len(x), _dp(e), _dp(e10))
    want = np.array([float(mp.exp(mp.mpf(float(v)))) for v in x])
    want10 = np.array([float(mp.power(10, mp.mpf(float(v)))) for v in x])
    ok = (want > 1e-300) & np.isfinite(want)
    ok10 = (want10 > 1e-300) & np.isfinite(want10)
    assert (np.abs(e - want)[ok] <= np.spacing(want[ok])).all()
    assert (np.abs(e10 - want10)[ok10] <= np.spacing(want10[ok10])).all()
    k = np.arange(-300.0, 301.0)
    p10 = np.empty(len(k))
    emul.emul_exp(_dp(k), len(k), _dp(np.empty(len(k))), _dp(p10))
    exact = np.array([float(mp.power(10, int(v))) for v in k])
    assert (np.abs(p10 - exact) <= np.spacing(exact)).all()
    print("10^k exact for", int(np.sum(p10 == exact)), "of", len(k), "integers; |k| <= 22:", bool(np.array_equal(p10[278:323], exact[278:323])))
    edge = np.array([0.0, -746.0, -1e10, -np.inf, 710.0, 1e10, np.inf, -324.0, 309.0])
    a = np.empty(len(edge)); b = np.empty(len(edge))
    emul.emul_exp(_dp(edge), len(edge), _dp(a), _dp(b))
    assert a[0] == 1.0 and b[0] == 1.0
    assert np.all(a[1:4] == 0.0) and np.all(np.isinf(a[4:7]))
    assert np.all(b[[1, 2, 3, 7]] == 0.0) and np.all(np.isinf(b[[4, 5, 6, 8]]))



def test_pow_third_of_the_fast_flavour(emul):
    """pf_pow_third = cbrt(x) (1 - d ln x) with ln x from the exponent and a quadratic in the mantissa: the factor beside
    the cube root against x^-d in 40 digits (d = 1/3 - 0.333333333333333, the reference's exponent is not 1/3), over 60
    decades.  The cube root itself is the library's (glibc's here, off by up to 3 ulp; the device library's on the GPU,
    where test_fast_flavour_elementary_functions_on_the_device holds the product to 2 ulp of x^0.333333333333333)."""
    import mpmath as mp
    mp.mp.dps = 40
    emul.emul_pow_third.argtypes = [dp, C.c_long, dp]
    libm = C.CDLL("libm.so.6")
    libm.cbrt.restype = C.c_double
    libm.cbrt.argtypes = [C.c_double]
    rng = np.random.default_rng(31)
    x = np.concatenate([10.0 ** rng.uniform(-30, 30, 30000), rng.uniform(0.5, 2.0, 5000), 2.0 ** np.arange(-60.0, 61.0)])
    got = np.empty(len(x))
    emul.emul_pow_third(_dp(x), len(x), _dp(got))
    d = mp.mpf(1) / 3 - mp.mpf(0.333333333333333)
    cb = np.array([libm.cbrt(float(v)) for v in x])
    factor = np.array([float(mp.mpf(float(g)) / mp.mpf(float(c)) - 1) for g, c in zip(got, cb)])
    want = np.array([float(mp.power(mp.mpf(float(v)), -d) - 1) for v in x])
    assert np.max(np.abs(factor - want)) <= 2.3e-16        # one rounding of the factor, one of the product
    assert np.max(np.abs(want)) > 2e-14                     # and the factor is not 1: 2.4e-14 at 1e30


def test_invariants_and_lpt_contraction_of_the_zpass(emul):
    """pf_invariants + pf_eigen_from_invariants (what k_c2r_invariants stores and k_collapse_inv starts from) give the
    eigenvalues of pf_ordered_eigenvalues bit for bit (same operations), and those are the eigenvalues of the tensor
    (numpy); pf_lpt3b_accumulate is s - 2 sum_ab phi2_ab h_ab in the reference's order of the components."""
    iv = C.POINTER(C.c_int)
    emul.emul_invariants.argtypes = [dp, C.c_long, dp, dp, iv]
    emul.emul_lpt3b.argtypes = [dp, dp, dp, C.c_long, dp]
    rng = np.random.default_rng(77)
    n = 20000
    d = rng.standard_normal((n, 6)) * 10.0 ** rng.integers(-3, 3, (n, 1))
    mu = np.empty((n, 3)); lam = np.empty((n, 3)); ok = np.empty(n, dtype=np.int32)
    emul.emul_invariants(_dp(d), n, _dp(mu), _dp(lam), ok.ctypes.data_as(iv))
    T = np.zeros((n, 3, 3))
    T[:, 0, 0], T[:, 1, 1], T[:, 2, 2], T[:, 0, 1], T[:, 0, 2], T[:, 1, 2] = d.T
    T[:, 1, 0], T[:, 2, 0], T[:, 2, 1] = T[:, 0, 1], T[:, 0, 2], T[:, 1, 2]
    assert np.allclose(mu[:, 0], np.trace(T, axis1=1, axis2=2), rtol=1e-13, atol=0)
    assert np.allclose(mu[:, 2], np.linalg.det(T), rtol=1e-9, atol=1e-12 * np.abs(d).max(axis=1) ** 3)
    good = ok == 1
    assert good.mean() > 0.99
    ref = np.linalg.eigvalsh(T)[:, ::-1]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.max(np.abs(lam[good] - ref[good]) / scale[good]) < 1e-7     # the trigonometric formula near double roots
    assert np.median(np.abs(lam[good] - ref[good]) / scale[good]) < 1e-15
    s = rng.standard_normal(n); ph = rng.standard_normal((n, 6)); h = rng.standard_normal((n, 6))
    out = np.empty(n)
    emul.emul_lpt3b(_dp(s), _dp(ph), _dp(h), n, _dp(out))
    w = np.array([2.0, 2.0, 2.0, 4.0, 4.0, 4.0])
    assert np.allclose(out, s - (w * ph * h).sum(axis=1), rtol=1e-13, atol=1e-13)
    # operation order of the reference: 11,12,13,22,23,33, each term 2 (or 4) * phi2 * h, subtracted one by one
    want = s.copy()
    for c in (0, 3, 4, 1, 5, 2):
        want -= (2.0 if c < 3 else 4.0) * ph[:, c] * h[:, c]
    assert np.array_equal(out, want)

