"""Pins the CPU oracle (oracle/pf_oracle.c): reference known answers
(SURVEY.md Appendix D), GSL-spline restatement vs scipy, own FFT vs pocketfft,
and the whole field-level path vs the independent numpy restatement."""
import json
import os

import ctypes as C

import numpy as np
import pytest

import np_restatement as npr
import oracle_lib
from pinocchio_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLD, "collapse_kat.json")) as f:
        return json.load(f)


def _eds_oracle(n=8):
    o = oracle_lib.Oracle(n, 1)
    x, y = synth.invgrow_table("eds")
    o.set_invgrow(x, y)
    return o


def test_ell_classic_kat(kat):
    for case in kat["ell_classic"]:
        got = oracle_lib.ell_classic(*case["l"])
        assert got == pytest.approx(case["bc"], rel=4e-16, abs=0), (case, got)


def test_inverse_collapse_time_kat(kat):
    o = _eds_oracle()
    for case in kat["inverse_collapse_time"]:
        f, eig, fail = o.inverse_collapse_time(case["d"])
        assert fail == 0
        # the EdS spline is the identity up to spline round-off: F = 1/b_c
        assert f == pytest.approx(case["F"], rel=1e-14, abs=1e-15), (case, f)
        if "degenerate" in case["branch"]:
            # acos conditioning: reference itself is only good to 1e-8 here
            assert np.allclose(eig, case["eig"], rtol=0, atol=5e-8)
        else:
            assert np.allclose(eig, case["eig"], rtol=1e-15, atol=1e-15), (case, eig)


def test_fail_sentinel_and_nan_quirks():
    o = _eds_oracle()
    # q<0 cannot happen for real symmetric input, q^3<r^2 by round-off can: sentinel -10, fail=0 (quirk Q4)
    f, _, fail = o.inverse_collapse_time([1.0, 1.0, 1.0 + 1e-9, 0, 0, 0])
    assert fail == 0 and (f == -10.0 or f > 0)


def test_spline_matches_scipy_natural():
    from scipy.interpolate import CubicSpline
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(8, 1)
    o.set_invgrow(x, y)
    cs = CubicSpline(x, y, bc_type="natural")
    xs = np.linspace(x[0], x[-1], 2001)
    got = np.array([o.spline_eval(v) for v in xs])
    assert np.max(np.abs(got - cs(xs))) < 5e-14
    # knots are reproduced exactly
    for i in (0, 1, 57, 208, 209):
        assert o.spline_eval(x[i]) == pytest.approx(y[i], abs=1e-15)
    # linear extrapolation outside the knots (src/cosmo.c:2016-2027)
    lo = y[0] + (x[0] - 1.0 - x[0]) * (y[1] - y[0]) / (x[1] - x[0])
    assert o.spline_eval(x[0] - 1.0) == pytest.approx(lo, rel=1e-15)
    hi = y[-1] + 0.5 * (y[-1] - y[-2]) / (x[-1] - x[-2])
    assert o.spline_eval(x[-1] + 0.5) == pytest.approx(hi, rel=1e-15)
    # and against the numpy restatement's dense-solve spline
    sp = npr.Spline(x, y)
    assert np.max(np.abs(got - sp(xs))) < 5e-14


def test_inverse_growing_mode_eds_identity():
    o = _eds_oracle()
    for d in (0.05, 0.3, 1.0, 1.4):
        assert o.inverse_growing_mode(d) == pytest.approx(1.0 / d - 1.0, rel=1e-13)


@pytest.mark.parametrize("n", [8, 16, 32])
def test_fft_against_pocketfft(n):
    rng = np.random.default_rng(n)
    o = oracle_lib.Oracle(n, 2)
    real = rng.standard_normal((n, n, n))
    spec = o.r2c(real)
    ref = np.fft.rfftn(real, axes=(0, 1, 2))
    assert np.max(np.abs(spec - ref)) < 1e-12 * np.max(np.abs(ref))
    # c2r of a NON-Hermitian half-spectrum: same semantics as irfftn (x,y c2c then z c2r)
    junk = rng.standard_normal((n, n, n // 2 + 1)) + 1j * rng.standard_normal((n, n, n // 2 + 1))
    got = o.c2r(junk)
    want = np.fft.irfftn(junk, s=(n, n, n), axes=(0, 1, 2)) * n ** 3
    assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want))


@pytest.mark.parametrize("n,rs", [(16, 0.0), (16, 1.5), (32, 2.8)])
def test_second_derivatives_vs_numpy(n, rs):
    dk = synth.make_density(n, seed=7 + n)
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk)
    got = o.second_derivatives(rs)
    want = npr.hessian(dk, rs)
    scale = max(np.max(np.abs(w)) for w in want)
    for g, w in zip(got, want):
        assert np.max(np.abs(g - w)) < 1e-13 * scale


@pytest.mark.parametrize("n,kind", [(16, "eds"), (32, "lcdm")])
def test_full_sweep_and_lpt_vs_numpy(n, kind):
    dk = synth.make_density(n, seed=synth.SEED)
    radii = synth.radii_ladder(6) / 4.0
    radii[-1] = 0.0
    x, y = synth.invgrow_table(kind)
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk)
    o.set_invgrow(x, y)
    o.set_growth(g)
    tv = o.compute_fmax(radii, do_lpt=True)
    p = o.products()

    fmax, rmax, tv_np, hes = npr.sweep(dk, radii, npr.Spline(x, y))
    assert np.allclose(tv, tv_np, rtol=1e-12)
    # Fmax: float32 of (nearly) the same double; allow 1 ulp on a tiny fraction
    diff = np.abs(p["Fmax"].astype(np.float64) - fmax.astype(np.float64))
    # tolerance: 2 ulp(fp32) evaluated at max(|F|,1); cells with F << 1 come from an
    # ill-conditioned (cancelling) cubic root and are never used downstream (F>=1 only)
    ulp = np.spacing(np.maximum(np.abs(fmax), 1.0).astype(np.float32)).astype(np.float64)
    assert np.all(diff <= 2 * ulp)
    assert np.mean(diff > 0) < 1e-3
    assert np.mean(p["Rmax"] != rmax) < 1e-3
    assert p["Fmax"].max() > 1.0  # something collapses

    d = npr.lpt(dk, hes, g)
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        a, b = p[name].astype(np.float64), d[name].astype(np.float64)
        amp = np.max(np.abs(b))
        assert amp > 0
        assert np.max(np.abs(a - b)) <= 4e-7 * amp, name
    k2, k31, k32 = d["_k"]
    for which, kk in enumerate((k2, k31, k32)):
        got = o.kvector(which)
        assert np.max(np.abs(got - kk)) < 1e-11 * np.max(np.abs(kk))

    # Fmax PDF (src/fmax.c:509-550)
    h = o.fmax_pdf()
    assert int(h.sum()) == n ** 3
    xf = np.clip((p["Fmax"].astype(np.float64) * 10.0).astype(np.int64), 0, 209)
    assert np.array_equal(h, np.bincount(xf.ravel(), minlength=210).astype(np.uint64))


def test_thread_count_independence():
    n = 16
    dk = synth.make_density(n, seed=3)
    radii = np.array([2.0, 1.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    res = []
    for nt in (1, 4):
        o = oracle_lib.Oracle(n, nt)
        o.set_density(dk)
        o.set_invgrow(x, y)
        tv = o.compute_fmax(radii, do_lpt=True)
        res.append((tv, o.products()))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-13)
    for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(res[0][1][name], res[1][1][name]), name


def _interpolate_growth(k, T, logkmin=-3.0, dlogk=0.5):
    """InterpolateGrowth, SCALE_DEPENDENT branch (src/cosmo.c:1728-1755), vectorised"""
    T = np.asarray(T)
    nk = len(T)
    kmin, kmax = 10.0 ** logkmin, 10.0 ** (logkmin + (nk - 1) * dlogk)
    dk = (np.log10(np.maximum(k, kmin)) - logkmin) / dlogk
    kk = np.clip(dk.astype(np.int64), 0, nk - 2)
    fr = dk - kk
    v = fr * T[kk + 1] + (1 - fr) * T[kk]
    v = np.where(k < kmin, T[0], v)
    return np.where(k > kmax, T[-1], v)


def test_scale_dependent_growth_and_per_radius_splines_vs_numpy():
    """the SCALE_DEPENDENT additions of the oracle: growth per mode from the k-binned tables (|k| in rad/cell as in
    src/fmax-pfft.c:339-364) against a numpy restatement, and SPLINE_INVGROW[ismooth] selection"""
    n = 16
    dk = synth.make_density(n, seed=12)
    radii = np.array([2.0, 1.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    j = np.arange(10)
    tabs = [np.log10(abs(g[o]) * (1.0 + 0.05 * (o + 1) * j)) for o in range(4)]
    signs = [1.0, 1.0, -1.0, 1.0]
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk); o.set_invgrow(x, y)
    for k in range(4):
        o.set_growth_table(k + 1, tabs[k], sign=signs[k])
    o.compute_fmax(radii, do_lpt=True)
    p = o.products()
    kx, ky, kz = npr.kvecs(n)
    kmod = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2)
    gk = [signs[k] * 10.0 ** _interpolate_growth(kmod, tabs[k]) for k in range(4)]
    assert gk[0].max() / gk[0].min() > 1.05            # the grid's |k| range crosses several bins
    hes = npr.hessian(dk, 0.0)
    d = npr.lpt(dk, hes, gk)
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        a, b = p[name].astype(np.float64), d[name].astype(np.float64)
        assert np.max(np.abs(a - b)) <= 4e-7 * np.max(np.abs(b)), name
    # a constant table is the scalar multiplier
    o2 = oracle_lib.Oracle(n, 2)
    o2.set_density(dk); o2.set_invgrow(x, y); o2.set_growth(g)
    o2.compute_fmax(radii, do_lpt=True)
    p2 = o2.products()
    for k in range(4):
        o.set_growth_table(k + 1, np.full(10, np.log10(abs(g[k]))), sign=signs[k])
    o.displacements(compute_sources=False)
    pc = o.products()
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(pc[name] - p2[name])) <= 2e-7 * np.max(np.abs(p2[name]))
    # per-radius splines: identical ones change nothing, a different one at radius 1 changes Fmax where Rmax == 1
    for i in range(3):
        o2.set_invgrow_radius(i, x, y)
    o2.compute_fmax(radii, do_lpt=False)
    assert np.array_equal(o2.products()["Fmax"], p2["Fmax"])
    x3, y3 = synth.invgrow_table("lcdm", omega0=0.4)
    o2.set_invgrow_radius(1, x3, y3)
    o2.compute_fmax(radii, do_lpt=False)
    p3 = o2.products()
    per = [npr.sweep(dk, radii[i:i + 1], npr.Spline(*(x3, y3) if i == 1 else (x, y)))[0] for i in range(3)]
    want = np.full((n, n, n), -10.0, dtype=np.float32)
    rwant = np.full((n, n, n), -1, dtype=np.int32)
    for i in range(3):                                  # the running max of src/collapse_times.c:640-652
        upd = per[i] > want
        want = np.where(upd, per[i], want)
        rwant = np.where(upd, np.int32(i), rwant)
    assert np.any(p3["Fmax"] != p2["Fmax"])
    ulp = np.spacing(np.maximum(np.abs(want), 1.0).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(p3["Fmax"].astype(np.float64) - want) <= 2 * ulp)
    assert np.mean(p3["Rmax"] != rwant) < 1e-3


def test_select_sorted_is_the_fragmentation_order():
    """orc_select_sorted against numpy: Fmax >= Flast, descending Fmax, ties by index"""
    n = 16
    o = oracle_lib.Oracle(n, 2)
    o.set_density(synth.make_density(n, seed=4))
    o.set_invgrow(*synth.invgrow_table("lcdm"))
    o.compute_fmax(np.array([2.0, 1.0, 0.0]), do_lpt=False)
    F = o.products()["Fmax"].ravel()
    for flast in (1.0, 1.5, -20.0, 1e9):
        idx, f = o.select_sorted(flast)
        sel = np.flatnonzero(F >= np.float32(flast))
        order = sel[np.lexsort((sel, -F[sel].astype(np.float64)))]
        assert np.array_equal(idx, order.astype(np.uint32)) and np.array_equal(f, F[order])
    assert len(o.select_sorted(-20.0)[0]) == n ** 3 and len(o.select_sorted(1e9)[0]) == 0


def test_tabulated_ct_restatement_vs_scipy():
    """TABULATED_CT (row f-4; pinned end to end by the reference's f(R) run in test_hmf256_kat.py; here unit by unit):
    the delta sampling, the table of ell() and the bilinear-of-splines interpolation against scipy's natural spline"""
    from scipy.interpolate import CubicSpline
    n = 16
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(n, 4)
    o.set_invgrow(x, y)
    var = 1.7
    tab, dv = o.ct_build(0, var)
    # sampling (src/collapse_times.c:836-876): starts at -CT_RANGE_D, finest (1.2 ref_interval) around CT_DELTA0 = -1
    assert dv[0] == -7.0 and np.all(np.diff(dv) > 0) and 6.0 < dv[-1] < 8.0
    steps = np.diff(dv)
    fine = dv[:-1][steps <= steps.min() * (1 + 1e-12)]      # the plateau of smallest steps straddles CT_DELTA0
    assert fine.min() < -1.0 < fine.max() + steps.min() and steps.max() / steps.min() > 2.0
    # table nodes are ell() of the (delta, x, y) grid scaled by sqrt(variance)
    ampl, bin_x = np.sqrt(var), 3.5 / 50
    rng = np.random.default_rng(3)
    for _ in range(200):
        iy, ix, idd = rng.integers(0, 50), rng.integers(0, 50), rng.integers(0, 100)
        xx, yy = ix * bin_x, iy * bin_x
        l = [(dv[idd] + 2 * xx + yy) / 3.0 * ampl, (dv[idd] - xx + yy) / 3.0 * ampl, (dv[idd] - xx - 2 * yy) / 3.0 * ampl]
        bc = o.L.orc_ell_classic(*l)
        want = 1.0 + o.L.orc_inverse_growing_mode(o.h, bc) if bc > 0 else 0.0
        assert tab[iy, ix, idd] == want
    assert (tab > 0).mean() > 0.3 and (tab == 0).mean() > 0.05
    # interpolation: four natural splines in delta, bilinear in (x, y)
    for _ in range(300):
        d, xx, yy = rng.uniform(-6.5, 6.5), rng.uniform(0, 3.4), rng.uniform(0, 3.4)
        l1, l2, l3 = (d + 2 * xx + yy) / 3.0 * ampl, (d - xx + yy) / 3.0 * ampl, (d - xx - 2 * yy) / 3.0 * ampl
        dd, x2, y2 = (l1 + l2 + l3) / ampl, (l1 - l2) / ampl, (l2 - l3) / ampl
        ix, iy = min(int(x2 / bin_x), 48), min(int(y2 / bin_x), 48)
        fx, fy = x2 / bin_x - ix, y2 / bin_x - iy
        sp = lambda i, j: float(CubicSpline(dv, tab[j, i], bc_type="natural")(dd))
        want = (1 - fx) * (1 - fy) * sp(ix, iy) + fx * (1 - fy) * sp(ix + 1, iy) + (1 - fx) * fy * sp(ix, iy + 1) + fx * fy * sp(ix + 1, iy + 1)
        got = o.interpolate_collapse_time(l1, l2, l3)
        assert abs(got - want) <= 1e-10 * max(1.0, abs(want))
    # beyond the delta range my_spline_eval extrapolates linearly from the end knots
    l = [(9.0 + 0.2) / 3.0 * ampl, (9.0 - 0.1) / 3.0 * ampl, (9.0 - 0.1) / 3.0 * ampl]
    t0 = tab[0, 1, :]  # not exactly at a node; only continuity matters here
    assert np.isfinite(o.interpolate_collapse_time(*l))
    # the sweep with the table in place: same cells collapse, F close to the direct solve where F >= 1
    dk = synth.make_density(n, seed=12)
    radii = np.array([2.0, 0.0])
    o.set_density(dk)
    tv = o.compute_fmax(radii, do_lpt=False)
    direct = o.products()["Fmax"].copy()
    o.set_tabulated_ct(tv)          # the reference uses the expected variance; the measured one will do here
    o.compute_fmax(radii, do_lpt=False)
    tabbed = o.products()["Fmax"]
    both = (direct >= 1.0) & (tabbed >= 1.0)
    assert both.mean() > 0.05 and np.mean((direct >= 1.0) != (tabbed >= 1.0)) < 0.02
    assert np.median(np.abs(tabbed[both] - direct[both]) / direct[both]) < 5e-3
    o.set_tabulated_ct([])
    o.compute_fmax(radii, do_lpt=False)
    assert np.array_equal(o.products()["Fmax"], direct)


def test_trilinear_and_all_spline_table_interpolation_vs_independent_construction():
    """-DTRILINEAR / -DALL_SPLINE (src/collapse_times.c:1153-1216; tests/Readme_Pinocchio_tests_V5_1.txt offers the three
    flavours).  PARITY UNPINNED: no output of the reference made with either exists, and gsl_spline2d's bicubic is restated
    from GSL's published source.  Checked here against constructions that share no code with it: numpy's trilinear weights;
    scipy natural splines for the node derivatives (rows, columns, rows of the column derivatives, as GSL's bicubic_init)
    and the tensor product of cubic Hermite basis functions for the patch -- the bicubic patch through sixteen corner
    values is unique, so any correct formula gives the same polynomial."""
    from scipy.interpolate import CubicSpline
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(16, 4)
    o.set_invgrow(x, y)
    var = 1.3
    tab, dv = o.ct_build(0, var)
    ampl, bin_x = np.sqrt(var), 3.5 / 50
    rng = np.random.default_rng(8)
    pts = [(rng.uniform(-6.5, 6.5), rng.uniform(0, 3.4), rng.uniform(0, 3.4)) for _ in range(150)]
    pts += [(-1.0, 0.01, 0.02), (0.5, 3.40, 0.03), (2.0, 0.03, 3.42), (3.0, 3.41, 3.41), (-7.5, 1.0, 1.0), (8.5, 0.5, 0.2)]   # edge cells, beyond the delta range

    def lam(d, xx, yy):
        return (d + 2 * xx + yy) / 3.0 * ampl, (d - xx + yy) / 3.0 * ampl, (d - xx - 2 * yy) / 3.0 * ampl

    def node_spline(i, j, dd):      # my_spline_eval: natural spline inside, linear extrapolation from the end knots outside
        t = tab[j, i]
        if dd < dv[0]:
            return t[0] + (dd - dv[0]) * (t[1] - t[0]) / (dv[1] - dv[0])
        if dd > dv[-1]:
            return t[-1] + (dd - dv[-1]) * (t[-1] - t[-2]) / (dv[-1] - dv[-2])
        return float(CubicSpline(dv, t, bc_type="natural")(dd))

    o.set_ct_interpolation(1)
    for d, xx, yy in pts:
        l1, l2, l3 = lam(d, xx, yy)
        dd, x2, y2 = (l1 + l2 + l3) / ampl, (l1 - l2) / ampl, (l2 - l3) / ampl
        ix, iy = min(int(x2 / bin_x), 48), min(int(y2 / bin_x), 48)
        idd = int(np.clip(np.searchsorted(dv, dd, side="right") - 1, 0, 98))
        w = [(dd - dv[idd]) / (dv[idd + 1] - dv[idd]), x2 / bin_x - ix, y2 / bin_x - iy]
        want = sum(tab[iy + c, ix + b, idd + a] * (w[0] if a else 1 - w[0]) * (w[1] if b else 1 - w[1]) * (w[2] if c else 1 - w[2])
                   for a in (0, 1) for b in (0, 1) for c in (0, 1))
        got = o.interpolate_collapse_time(l1, l2, l3)
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (d, xx, yy)
    # at a node the trilinear value is the table entry itself
    l = lam(dv[40], 7 * bin_x, 9 * bin_x)
    assert abs(o.interpolate_collapse_time(*l) - tab[9, 7, 40]) <= 1e-9 * max(1.0, tab[9, 7, 40])

    h00 = lambda t: 2 * t ** 3 - 3 * t ** 2 + 1
    h10 = lambda t: t ** 3 - 2 * t ** 2 + t
    h01 = lambda t: -2 * t ** 3 + 3 * t ** 2
    h11 = lambda t: t ** 3 - t ** 2
    o.set_ct_interpolation(2)
    for d, xx, yy in pts:
        l1, l2, l3 = lam(d, xx, yy)
        dd, x2, y2 = (l1 + l2 + l3) / ampl, (l1 - l2) / ampl, (l2 - l3) / ampl
        ix, iy = min(int(x2 / bin_x), 48), min(int(y2 / bin_x), 48)
        ixs = 0 if ix == 0 else (46 if ix >= 48 else ix - 1)
        iys = 0 if iy == 0 else (46 if iy >= 48 else iy - 1)
        xs, ys = (np.arange(4) + ixs) * bin_x, (np.arange(4) + iys) * bin_x
        z = np.array([[node_spline(ixs + i, iys + j, dd) for i in range(4)] for j in range(4)])       # z[j, i]
        zx = np.array([CubicSpline(xs, z[j], bc_type="natural")(xs, 1) for j in range(4)])
        zy = np.array([CubicSpline(ys, z[:, i], bc_type="natural")(ys, 1) for i in range(4)]).T
        zxy = np.array([CubicSpline(xs, zy[j], bc_type="natural")(xs, 1) for j in range(4)])
        i = int(np.clip(np.searchsorted(xs, x2, side="right") - 1, 0, 2))
        j = int(np.clip(np.searchsorted(ys, y2, side="right") - 1, 0, 2))
        hx, hy = xs[i + 1] - xs[i], ys[j + 1] - ys[j]
        t, u = (x2 - xs[i]) / hx, (y2 - ys[j]) / hy
        bx = [(h00(t), h10(t) * hx), (h01(t), h11(t) * hx)]     # (value, derivative) weights at the low and high x node
        by = [(h00(u), h10(u) * hy), (h01(u), h11(u) * hy)]
        want = 0.0
        for a in (0, 1):
            for b in (0, 1):
                want += (z[j + b, i + a] * bx[a][0] * by[b][0] + zx[j + b, i + a] * bx[a][1] * by[b][0] +
                         zy[j + b, i + a] * bx[a][0] * by[b][1] + zxy[j + b, i + a] * bx[a][1] * by[b][1])
        got = o.interpolate_collapse_time(l1, l2, l3)
        assert abs(got - want) <= 1e-10 * max(1.0, abs(want)), (d, xx, yy, got, want)
    l = lam(dv[40], 7 * bin_x, 9 * bin_x)
    assert abs(o.interpolate_collapse_time(*l) - tab[9, 7, 40]) <= 1e-9 * max(1.0, tab[9, 7, 40])
    # the three flavours agree to the table's own resolution where it is smooth, and the sweep runs with each of them
    dk = synth.make_density(16, seed=12)
    radii = np.array([2.0, 0.0])
    o.set_density(dk)
    o.set_ct_interpolation(0)
    tv = o.compute_fmax(radii, do_lpt=False)
    o.set_tabulated_ct(tv)
    res = []
    for fl in (0, 1, 2):
        o.set_ct_interpolation(fl)
        o.compute_fmax(radii, do_lpt=False)
        res.append(o.products()["Fmax"].copy())
    both = (res[0] >= 1.0) & (res[1] >= 1.0) & (res[2] >= 1.0)
    assert both.mean() > 0.05
    for fl in (1, 2):
        assert np.median(np.abs(res[fl][both] - res[0][both]) / res[0][both]) < 5e-3 and not np.array_equal(res[fl], res[0])


def test_double_precision_products_build_of_the_oracle():
    """-DDOUBLE_PRECISION_PRODUCTS (PRODFLOAT double, src/pinocchio.h:219-225): the same restatement with the one typedef
    changed.  Displacements: the float build's values are these rounded once; Fmax: without the fp32 rounding of the running
    maximum (quirk Q2 has nothing left to act on) -- so Rmax may move where two radii give F within an fp32 ulp."""
    n = 16
    dk = synth.make_density(n, seed=33)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([2.0, 1.0, 0.0])
    res = []
    for dp in (False, True):
        o = oracle_lib.Oracle(n, 2, double_products=dp)
        o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
        tv = o.compute_fmax(radii, do_lpt=True)
        res.append((tv, o.products(), o.fmax_pdf()))
    (tv4, p4, h4), (tv8, p8, h8) = res
    assert p4.dtype.itemsize == 56 and p8.dtype.itemsize == 112
    assert np.array_equal(tv4, tv8)
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert p8[name].dtype == np.float64 and np.array_equal(p8[name].astype(np.float32), p4[name]), name
    assert np.array_equal(p8["Fmax"].astype(np.float32), p4["Fmax"]) or np.mean(p8["Fmax"].astype(np.float32) != p4["Fmax"]) < 1e-3
    assert np.mean(p8["Rmax"] != p4["Rmax"]) < 1e-3 and np.abs(h4.astype(np.int64) - h8.astype(np.int64)).sum() <= 4
    assert not np.array_equal(p8["Fmax"].astype(np.float32).astype(np.float64), p8["Fmax"])


def _sng_rhs(t, y, cosmo):
    O0, OL, Or, Ok, fr0, hoc, size = cosmo
    z = 1.0 / t - 1.0
    E2 = (Or * (1 + z) ** 4 + O0 * (1 + z) ** 3 + Ok * (1 + z) ** 2 + OL) / (Or + O0 + Ok + OL)
    om, ol = O0 * (1 + z) ** 3 / E2, OL / E2
    la, lv, ld = y[0:3], y[3:6], y[6:9]
    delta = ld.sum()
    f = np.zeros(9)
    for i in range(3):
        s = 0.0
        for j in range(3):
            if i == j or la[i] == la[j]:
                continue
            s += (ld[j] - ld[i]) * ((1 - la[i]) ** 2 * (1 + lv[i]) - (1 - la[j]) ** 2 * (1 + lv[j])) / ((1 - la[i]) ** 2 - (1 - la[j]) ** 2)
        f[i] = lv[i] * (la[i] - 1) / t
        fm = 0.0
        if fr0:                                     # ForceModification, src/collapse_times.c:295-312
            ff = 4.0 * OL / O0
            th = fr0 / O0 / (hoc * size) ** 2 * t ** 7 * (1 + delta) ** (-1 / 3) * (((1 + ff) / (1 + ff * t ** 3)) ** 2 - ((1 + ff) / (1 + delta + ff * t ** 3)) ** 2)
            f3 = max(th * (3 + th * (-3 + th)), 0.0)
            fm = f3 / 3 if f3 < 1 else 1 / 3
        f[i + 3] = 0.5 * (lv[i] * (om - 2 * ol - 2) - 3 * om * ld[i] * (1 + fm) - 2 * lv[i] ** 2) / t
        f[i + 6] = ((5 / 6 + ld[i]) * ((3 + lv.sum()) - (1 + delta) / (2.5 + delta) * lv.sum()) - (2.5 + delta) * (1 + lv[i]) + s) / t
    return f


def test_ell_sng_restatement_vs_scipy():
    """ELL_SNG (row f-4; pinned end to end by the reference's f(R) run in test_hmf256_kat.py): the restated system + GSL-style
    RKF45 loop against scipy's DOP853 at tight tolerance, and the textbook spherical-collapse threshold"""
    from scipy.integrate import solve_ivp
    L = oracle_lib.lib()
    dp = C.POINTER(C.c_double)
    eds, lcdm = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]), np.array([0.25, 0.75, 0.0, 0.0, 0.0, 0.0, 0.0])
    fofr = np.array([0.25, 0.75, 0.0, 0.0, 1e-5, 100.0 / 299792.458, 2.0])     # MOD_GRAV_FR, |f_R0| = 1e-5, R = 2 Mpc
    # a sphere in Einstein-de Sitter collapses when the linear overdensity reaches 1.686
    a = L.orc_ell_sng(1.0, 1.0, 1.0, 1e-5, eds.ctypes.data_as(dp))
    assert abs(3.0 * a - 1.686) < 2e-3
    ev = lambda t, y, c: y[0] - 0.99999
    ev.terminal, ev.direction = True, 1
    # the fifth force can only speed the collapse up
    gr = L.orc_ell_sng(1.2, 0.6, 0.3, 1.28e-5, lcdm.ctypes.data_as(dp))
    fr = L.orc_ell_sng(1.2, 0.6, 0.3, 1.28e-5, fofr.ctypes.data_as(dp))
    assert 0.0 < fr < gr and (gr - fr) / gr > 1e-3
    for cosmo, din in ((eds, 1e-5), (lcdm, 1.28e-5), (fofr, 1.28e-5)):
        for l in ([2.0, 1.0, 0.5], [1.5, 0.2, -0.4], [0.8, 0.7, 0.1], [3.0, -0.5, -1.0], [-0.2, -0.3, -0.5]):
            y0 = np.array([x * din for x in l] + [x * din / (x * din - 1) for x in l] + [x * din for x in l])
            sol = solve_ivp(_sng_rhs, (1e-5, 5.0), y0, method="DOP853", rtol=1e-10, atol=1e-12, args=(cosmo,), events=ev)
            want = sol.t_events[0][0] if len(sol.t_events[0]) else 0.0
            got = L.orc_ell_sng(l[0], l[1], l[2], din, cosmo.ctypes.data_as(dp))
            # the reference returns the end of the first accepted step past the threshold, scaled by 1/lambda_a (its
            # interpolation runs from the initial point): within one step, i.e. ~1e-4, of the true crossing
            assert abs(got - want) <= 3e-4 * max(want, 1e-30) if want > 0 else got == 0.0, (cosmo, l, got, want)
            f = L.orc_ell_sng_F(l[0], l[1], l[2], din, cosmo.ctypes.data_as(dp))
            assert f == (1.0 / got if got > 0 else 0.0)


@pytest.mark.parametrize("n", [12, 20])
def test_grid_sizes_that_are_not_a_power_of_two(n):
    """the oracle on 2^a 3^b 5^c sizes (plain O(n^2) transforms) against numpy/pocketfft: Hessians, sweep, displacements"""
    rng = np.random.default_rng(n)
    dk = np.fft.rfftn(rng.standard_normal((n, n, n)), axes=(0, 1, 2))
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    for a, b in zip(o.second_derivatives(1.3), npr.hessian(dk, 1.3)):
        assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b))
    radii = np.array([1.5, 0.0])
    tv = o.compute_fmax(radii, do_lpt=True)
    p = o.products()
    f, r, tvn, hes = npr.sweep(dk, radii, npr.Spline(x, y))
    assert np.allclose(tv, tvn, rtol=1e-12)
    ulp = np.spacing(np.maximum(np.abs(f), 1.0).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(p["Fmax"].astype(np.float64) - f) <= 2 * ulp) and np.mean(p["Rmax"] != r) < 1e-3
    d = npr.lpt(dk, hes, g)
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p[name].astype(np.float64) - d[name])) <= 4e-7 * np.max(np.abs(d[name]))
    with pytest.raises(ValueError):
        oracle_lib.Oracle(15, 1)


@pytest.mark.parametrize("n", [16, 32, 24])
def test_sampled_plane_oracle_is_the_whole_oracle_on_its_planes(n):
    """orc_plane_derivatives / orc_plane_collapse_times (the form the 1024^3 test of the GPU suite uses: the whole oracle
    would need ~450 GB there) against the whole-box oracle: second derivatives of every radius to rounding, plain
    transform, Fmax / Rmax after the sweep, on planes that include the first and the last"""
    dk = synth.make_density(n, seed=11 + n)
    radii = synth.radii_ladder(5) / (64.0 / n)
    radii[-1] = 0.0
    x, y = synth.invgrow_table("lcdm")
    planes = [0, 1, n // 3, n // 2, n - 1]
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk)
    o.set_invgrow(x, y)
    po = oracle_lib.PlaneOracle(n, planes, 2)
    po.set_invgrow(x, y)
    for rs in radii:
        want = o.second_derivatives(rs)
        got = po.derivatives(dk, rs, po.HESSIAN)
        scale = max(np.max(np.abs(w)) for w in want)
        for i in range(6):
            assert np.max(np.abs(got[i] - want[i][planes])) <= 1e-13 * scale, (rs, i)
    plain = po.derivatives(dk, 0.0, [(-1, -1)])[0]
    want = o.c2r(dk) / n ** 3
    assert np.max(np.abs(plain - want[planes])) <= 1e-13 * np.max(np.abs(want))
    o.compute_fmax(radii, do_lpt=False)
    p = o.products()
    fmax, rmax = po.sweep(dk, radii)
    wf, wr = p["Fmax"][planes], p["Rmax"][planes]
    ulp = np.spacing(np.maximum(np.abs(wf), 1.0).astype(np.float32)).astype(np.float64)
    diff = np.abs(fmax.astype(np.float64) - wf.astype(np.float64))
    assert np.mean(diff > 2 * ulp) <= 1e-3 and np.mean(diff > 0) < 5e-3   # two correct transforms differ by rounding
    assert np.mean(rmax != wr) < 1e-3
    assert fmax.max() > 1.0


@pytest.mark.parametrize("n", [16, 24])
def test_streamed_plane_oracle_is_the_plane_oracle_to_the_bit(n):
    """orc_plane_acc_* (the form the 2048^3 test of BASELINE config 5 uses: its spectrum is 69 GB in fp64 and arrives in pieces of
    consecutive kx, three radii at once) gives orc_plane_derivatives' planes bit for bit, whatever the pieces"""
    dk = synth.make_density(n, seed=3 + n)
    radii = np.array([3.0, 1.1, 0.0])
    planes = [0, n // 3, n - 1]
    po = oracle_lib.PlaneOracle(n, planes, 2)
    st = po.stream(radii, po.HESSIAN)
    kx = 0
    for piece in (1, 5, 2, n):
        m = min(piece, n - kx)
        if m:
            st.add(np.ascontiguousarray(dk[kx:kx + m]), kx)
        kx += m
    with pytest.raises(AssertionError):
        st.add(np.ascontiguousarray(dk[:1]), 0)         # every row was added already
    o = oracle_lib.Oracle(n, 2)
    o.set_density(dk)
    x, y = synth.invgrow_table("lcdm")
    o.set_invgrow(x, y)
    tv = o.compute_fmax(radii, do_lpt=False)
    for i, rs in enumerate(radii):
        assert abs(st.power(i) / float(n) ** 6 - tv[i]) <= 1e-12 * tv[i], (rs, st.power(i) / float(n) ** 6, tv[i])   # Parseval against TrueVariance
        assert np.array_equal(st.finish(i), po.derivatives(dk, rs, po.HESSIAN)), rs
    st.close()
