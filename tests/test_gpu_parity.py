"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle
on identical inputs, plus size-independent properties at large grids.

Tolerances (fp64 field path), as stated in DESIGN.md:
  * Hessian fields, LPT spectra : max abs diff <= 1e-12 x field amplitude
  * TrueVariance                : rel <= 1e-12
  * Fmax (stored fp32)          : |diff| <= 2 ulp_fp32(max(|F|,1)) and > 0 on < 1e-3 of cells, EXCEPT on
                                  <= 2e-5 of the cells, where the reference's own cubic is ill-conditioned
                                  (den = det/126 + 5 l1 d (d-l1)/84 cancels when d = l1+l2+l3 ~ 0,
                                  src/collapse_times.c:133): there |diff| <= 2e-3, and the oracle fed with
                                  the GPU's Hessian (equal to its own to ~1e-15) reproduces the GPU value
  * per-cell solver, equal input: identical sentinels and zeros; PF_EXACT_LIBM=1 (the reference's own libm calls):
                                  <= 1e-12 relative on > 99.95 % of random Hessians, and every Fmax outlier of the
                                  full path is reproduced bit for bit by the oracle fed with the GPU's Hessian;
                                  default (sincos/cbrt/exp10 forms, ~1 ulp per call): <= 1e-10 on > 99.9 %, and every
                                  outlier lies within 8x the spread the oracle itself shows under 2-ulp input noise
  * Rmax                        : identical on >= 99.9 % of cells
  * displacements (stored fp32) : |diff| <= 4e-7 x amplitude (fp32 rounding of equal fp64 values)
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
from pinocchio_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def api():
    from pinocchio_amd import api as _api
    return _api


def _fmax_close(got, want, max_abs=2e-3):
    """returns the indices of the ill-conditioned outlier cells (normally none below 64^3)"""
    ulp = np.spacing(np.maximum(np.abs(want), 1.0).astype(np.float32)).astype(np.float64)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bad = d > 2 * ulp
    assert bad.sum() <= max(1, int(2e-5 * d.size)), (int(bad.sum()), d.max(), np.argwhere(bad)[:3])  # never less than one cell
    if max_abs is not None:
        assert d.max() <= max_abs, d.max()
    assert np.mean(d > 0) < 1e-3, np.mean(d > 0)
    return np.argwhere(bad)


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256])
def test_transforms_vs_pocketfft(api, n):
    rng = np.random.default_rng(n)
    real = rng.standard_normal((n, n, n))
    with api.Fmax(n) as f:
        spec = f.forward_transform(real)
        want = np.fft.rfftn(real, axes=(0, 1, 2))
        assert np.max(np.abs(spec - want)) < 1e-13 * np.max(np.abs(want)) * np.log2(n)
        junk = rng.standard_normal((n, n, n // 2 + 1)) + 1j * rng.standard_normal((n, n, n // 2 + 1))
        back = f.reverse_transform(junk)  # c2r then 1/N^3 (src/fmax-pfft.c:203-228)
        want = np.fft.irfftn(junk, s=(n, n, n), axes=(0, 1, 2))
        assert np.max(np.abs(back - want)) < 1e-13 * np.max(np.abs(want)) * np.log2(n)
        # round trip
        again = f.reverse_transform(f.forward_transform(real))
        assert np.max(np.abs(again - real)) < 1e-13 * np.log2(n)


@pytest.mark.parametrize("libm", ["fast", "exact"])
def test_collapse_cells_kat_and_random(api, libm, monkeypatch):
    if libm == "exact":
        monkeypatch.setenv("PF_EXACT_LIBM", "1")
    else:
        monkeypatch.delenv("PF_EXACT_LIBM", raising=False)
    with open(os.path.join(GOLD, "collapse_kat.json")) as fh:
        kat = json.load(fh)
    with api.Fmax(64) as f:
        x, y = synth.invgrow_table("eds")
        f.set_invgrow(x, y)
        d = np.array([c["d"] for c in kat["inverse_collapse_time"]])
        F = f.collapse_cells(d)
        for i, case in enumerate(kat["inverse_collapse_time"]):
            tol = 1e-7 if "degenerate" in case["branch"] else 1e-12
            assert F[i] == pytest.approx(case["F"], rel=tol, abs=1e-14), case
        # random Hessians incl. diagonal / zero / q==0 cases against the oracle (LCDM spline)
        x, y = synth.invgrow_table("lcdm")
        f.set_invgrow(x, y)
        rng = np.random.default_rng(5)
        m = 40000
        d = rng.standard_normal((m, 6)) * np.array([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
        d[:50, 3:] = 0.0
        d[50:60] = 0.0
        d[60:70, :3] = 0.4
        d[60:70, 3:] = 0.0
        F = f.collapse_cells(d)
    o = oracle_lib.Oracle(8, 1)
    o.set_invgrow(x, y)
    want = np.array([o.inverse_collapse_time(row)[0] for row in d])
    both_nan = np.isnan(F) & np.isnan(want)
    # no FMA contraction in the device solver: same IEEE operations as the CPU, only libm differs
    assert np.array_equal(F == -10.0, want == -10.0)
    assert np.array_equal(F == 0.0, want == 0.0)
    rel = np.abs(F - want) / np.maximum(1.0, np.abs(want))
    rel[both_nan] = 0.0
    # libm differences (<= 1-2 ulp in acos/cos/pow/exp/log10) are amplified where the cubic is ill-conditioned
    if libm == "exact":
        assert np.mean((F == want) | both_nan) > 0.5  # the rest: 1-ulp libm differences
        assert np.mean(rel <= 1e-12) > 0.9995, np.mean(rel <= 1e-12)
        assert rel.max() <= 1e-6
    else:
        assert np.median(rel) < 5e-15
        assert np.mean(rel <= 1e-10) > 0.999, np.mean(rel <= 1e-10)
        assert rel.max() <= 1e-5
    assert np.mean(F.astype(np.float32) != want.astype(np.float32)) < 2e-4


@pytest.mark.parametrize("n,rs", [(16, 0.0), (32, 1.5), (64, 2.8), (64, 0.0)])
def test_second_derivatives_vs_oracle(api, n, rs):
    dk = synth.make_density(n, seed=11 + n)
    # put something in the DC mode: it must pass the filter untouched (k^2 = 0, src/fmax-pfft.c:368)
    dk[0, 0, 0] = 0.37 * n ** 3
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    want = o.second_derivatives(rs)
    with api.Fmax(n) as f:
        f.set_density(dk)
        f.compute_second_derivatives(rs)
        got = [f.second_derivative(i) for i in range(6)]
    amp = max(np.max(np.abs(w)) for w in want)
    for i in range(6):
        assert np.max(np.abs(got[i] - want[i])) < 1e-12 * amp, i


def _explain_outliers(api, n, radii, kind, outliers, p, exact):
    """Outliers come from the ~1e-15 difference of the Hessians in cells where the reference's cubic is
    ill-conditioned.  exact libm: the oracle's solver on the GPU's Hessian gives the GPU's Fmax bit for bit.
    default libm: the GPU value lies within 8x the spread of the oracle under 2-ulp noise on its input."""
    dk = synth.make_density(n, seed=synth.SEED)
    x, y = synth.invgrow_table(kind)
    o = oracle_lib.Oracle(8, 1)
    o.set_invgrow(x, y)
    rng = np.random.default_rng(1)
    with api.Fmax(n) as f:
        f.set_density(dk)
        for ir in sorted(set(int(p["Rmax"][tuple(c)]) for c in outliers)):
            f.compute_second_derivatives(radii[ir])
            hg = [f.second_derivative(i) for i in range(6)]
            for c in outliers:
                c = tuple(c)
                if p["Rmax"][c] != ir:
                    continue
                h = np.array([hh[c] for hh in hg])
                fo = o.inverse_collapse_time(h)[0]
                if exact:
                    assert np.float32(fo) == p["Fmax"][c], (c, fo, p["Fmax"][c])
                else:
                    spread = max(abs(o.inverse_collapse_time(h * (1.0 + rng.uniform(-4.4e-16, 4.4e-16, 6)))[0] - fo)
                                 for _ in range(32))
                    ulp = float(np.spacing(np.float32(max(abs(fo), 1.0))))
                    assert abs(float(p["Fmax"][c]) - fo) <= max(8.0 * spread, 2.0 * ulp), (c, fo, p["Fmax"][c], spread)


def _run_both(api, n, radii, kind="lcdm", seed=synth.SEED, field_bytes=8, do_lpt=True):
    dk = synth.make_density(n, seed=seed)
    x, y = synth.invgrow_table(kind)
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    o.set_invgrow(x, y)
    o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=do_lpt)
    with api.Fmax(n, field_bytes=field_bytes) as f:
        f.set_density(dk)
        f.set_invgrow(x, y)
        f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=do_lpt)
        p = f.products()
        pdf = f.Fmax_PDF()
        kv = [f.kvector(w) for w in range(3)] if do_lpt else None
    return (tv, p, pdf, kv), (tv_o, o.products(), o.fmax_pdf(), [o.kvector(w) for w in range(3)] if do_lpt else None)


@pytest.mark.parametrize("n,ns,kind", [(16, 4, "eds"), (32, 6, "lcdm"), (64, 12, "lcdm"), (128, 5, "lcdm")])
def test_full_path_vs_oracle(api, n, ns, kind):
    radii = synth.radii_ladder(12) * (n / 256.0) if ns == 12 else synth.radii_ladder(ns) * (n / 128.0)
    radii[-1] = 0.0
    (tv, p, pdf, kv), (tv_o, po, pdf_o, kv_o) = _run_both(api, n, radii, kind)
    assert np.allclose(tv, tv_o, rtol=1e-12), (tv, tv_o)
    outliers = _fmax_close(p["Fmax"], po["Fmax"])
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    assert (po["Fmax"] >= 1).sum() > 0
    if len(outliers):
        _explain_outliers(api, n, radii, kind, outliers, p, exact=False)
    for w in range(3):
        assert np.max(np.abs(kv[w] - kv_o[w])) < 1e-12 * np.max(np.abs(kv_o[w])) * np.log2(n), w
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        a, b = p[name].astype(np.float64), po[name].astype(np.float64)
        amp = np.max(np.abs(b))
        assert amp > 0
        assert np.max(np.abs(a - b)) <= 4e-7 * amp, name
        assert np.mean(a != b) < 0.02, name  # fp32 rounding of fp64 values equal to ~1e-15
    # Fmax PDF (src/fmax.c:509-550): a cell 1 ulp across a bin edge may move one count
    assert int(pdf.sum()) == n ** 3
    assert np.abs(pdf.astype(np.int64) - pdf_o.astype(np.int64)).sum() <= max(2, int(2e-4 * n ** 3))


def test_full_path_exact_libm_outliers_are_bit_explained(api, monkeypatch):
    """PF_EXACT_LIBM=1: the solver makes the reference's own libm calls; every Fmax outlier at 128^3 is then
    reproduced bit for bit by the oracle's solver on the GPU's Hessian."""
    monkeypatch.setenv("PF_EXACT_LIBM", "1")
    n = 128
    radii = synth.radii_ladder(5) * (n / 128.0)
    radii[-1] = 0.0
    (tv, p, pdf, _), (tv_o, po, pdf_o, _) = _run_both(api, n, radii, "lcdm", do_lpt=False)
    assert np.allclose(tv, tv_o, rtol=1e-12)
    outliers = _fmax_close(p["Fmax"], po["Fmax"])
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    _explain_outliers(api, n, radii, "lcdm", outliers, p, exact=True)


def test_golden_fixture(api):
    """committed vectors (tests/golden/make_golden.py wrote them from the oracle)"""
    z = np.load(os.path.join(GOLD, "sweep_n16.npz"))
    n = int(z["n"])
    with api.Fmax(n) as f:
        f.set_density(z["dk"])
        f.set_invgrow(z["spline_x"], z["spline_y"])
        f.set_growth(z["growth"])
        tv = f.compute_fmax(z["radii"], do_lpt=True)
        p = f.products()
        pdf = f.Fmax_PDF()
    assert np.allclose(tv, z["true_variance"], rtol=1e-12)
    _fmax_close(p["Fmax"], z["Fmax"])
    assert np.mean(p["Rmax"] != z["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        amp = np.max(np.abs(z[name]))
        assert np.max(np.abs(p[name].astype(np.float64) - z[name].astype(np.float64))) <= 4e-7 * amp
    assert np.abs(pdf.astype(np.int64) - z["pdf"].astype(np.int64)).sum() <= 2


def test_fmax_only_skips_lpt_and_reentry(api):
    """config 'Fmax-only': no displacement build -> Vel* stay 0 (src/collapse_times.c:472-489);
    compute_displacements(0,0,z) re-entry reuses the resident sources (src/fragment.c:398-410)"""
    n = 32
    radii = np.array([2.0, 1.0, 0.0])
    (tv, p, _, _), (tv_o, po, _, _) = _run_both(api, n, radii, do_lpt=False)
    assert np.allclose(tv, tv_o, rtol=1e-12)
    _fmax_close(p["Fmax"], po["Fmax"])
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert not p[name].any()
    dk = synth.make_density(n, seed=3)
    x, y = synth.invgrow_table("lcdm")
    with api.Fmax(n) as f:
        f.set_density(dk)
        f.set_invgrow(x, y)
        with pytest.raises(api.PinfmaxError):
            f.compute_displacements(0, 0)  # sources not resident yet
        f.set_growth(synth.growth_multipliers())
        f.compute_fmax(radii, do_lpt=True)
        p1 = f.products()
        g2 = synth.growth_multipliers() * np.array([0.5, 0.25, 0.125, 0.125])
        f.set_growth(g2)
        f.compute_displacements(0, 0)
        p2 = f.products()
    assert np.array_equal(p1["Fmax"], p2["Fmax"])
    assert np.allclose(p2["Vel"], 0.5 * p1["Vel"], rtol=2e-7, atol=0)
    assert np.allclose(p2["Vel_2LPT"], 0.25 * p1["Vel_2LPT"], rtol=2e-7, atol=0)
    assert np.allclose(p2["Vel_3LPT_2"], 0.125 * p1["Vel_3LPT_2"], rtol=2e-7, atol=0)


@pytest.mark.parametrize("n", [64, 256])
def test_fp32_field_path(api, n):
    """config 5: fp32 density/derivative fields, fp64 collapse solve.  The contract of DESIGN.md section 4, set from the measured
    distributions of profiles/r06_fp32_contract.json (256^3 against the oracle: 99.9 % of the cells with F >= 0.5 within 2.4e-6, 99.99 %
    within 2.9e-6; 1024^3 against fp64 fields: 3.8e-6 / 4.9e-6; the rest are the ill-conditioned cells of the cubic, where fp64 fields
    differ from the oracle as much): the 99.9 % quantile of |dFmax| <= 1e-5 (twice the measured one, rounded up) and all but 1e-5 of
    the cells within the survey's 1e-4."""
    radii = np.array([4.0, 2.0, 1.0, 0.0]) * (n / 64.0) ** 0.5
    (tv, p, pdf, _), (tv_o, po, pdf_o, _) = _run_both(api, n, radii, field_bytes=4)
    assert np.allclose(tv, tv_o, rtol=1e-5)
    sel = po["Fmax"] >= 0.5
    d = np.abs(p["Fmax"][sel].astype(np.float64) - po["Fmax"][sel].astype(np.float64))
    assert np.quantile(d, 0.999) <= 1e-5 and np.sum(d > 1e-4) <= max(2, int(1e-5 * d.size)), (float(np.quantile(d, 0.999)), int(np.sum(d > 1e-4)), d.size)
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        a, b = p[name].astype(np.float64), po[name].astype(np.float64)
        assert np.sqrt(np.mean((a - b) ** 2)) <= 2e-5 * np.sqrt(np.mean(b ** 2)), name
    assert np.abs(pdf.astype(np.int64) - pdf_o.astype(np.int64)).sum() <= 2e-3 * n ** 3


def test_fp32_fields_on_the_largest_box_one_gpu_holds(api):
    """BASELINE config 5's arithmetic (fp32 density / derivative fields, fp64 collapse solve, the six-component kernels that a
    2048^3 run takes) end to end at the largest size whose fp32 path fits one GPU, 1024^3 -- against the fp64-field run of the
    same modes: variances to fp32 accuracy, Fmax within the stated tolerance of the fp32 path, plus the properties that need no
    second run.  (The 2048-point kernels themselves: tests/test_gpu_lines.py.)"""
    n = 1024
    x, y = synth.invgrow_table("lcdm")
    radii = synth.radii_ladder(12)[[2, 8, 11]]          # one band-limited radius, one full, R = 0
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        tv64 = f.sweep(radii)
        fm64 = f.block("FMAX")
    with api.Fmax(n, field_bytes=4) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        tv = f.sweep(radii)
        pdf = f.Fmax_PDF()
        fm = f.block("FMAX")
        rm = f.block("RMAX")
        assert f.device_bytes < 165e9  # (161 GB since round 6: the fp64 invariant rows of the sweep live in the fields the LPT part needs anyway; 170 before)
    assert np.allclose(tv, tv64, rtol=2e-5) and tv[0] < tv[1] < tv[2]
    assert np.sqrt(tv[-1]) == pytest.approx(2.5, rel=1e-5)
    assert int(pdf.sum()) == n ** 3 and np.isfinite(fm).all()
    assert rm.min() >= -1 and rm.max() == 2
    sel = fm64 >= 0.5
    d = np.abs(fm[sel].astype(np.float64) - fm64[sel].astype(np.float64))
    nfar = int(np.sum(d > 1e-4))   # (the quantile of 6.8e8 numbers by a partition of a subsample: every 16th cell)
    assert np.quantile(d[::16], 0.999) <= 1e-5 and nfar <= 1e-5 * d.size, (float(np.quantile(d[::16], 0.999)), nfar, d.size)
    assert abs(float((fm >= 1.0).mean()) - float((fm64 >= 1.0).mean())) < 1e-4   # collapsed fraction


def test_synth_density_matches_numpy_mirror(api):
    n = 32
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        got = f.density()
    want = synth.philox_density(n, synth.SEED, 2.5, -2.0)
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))
    h = n // 2
    assert not got[h].any() and not got[:, h].any() and not got[:, :, h].any() and got[0, 0, 0] == 0


@pytest.mark.parametrize("n", [256, 512])
def test_properties_at_scale(api, n):
    """size-independent checks where the oracle is too slow"""
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([8.0, 2.0, 0.0])
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        f.set_growth(synth.growth_multipliers())
        tv = f.sweep(radii)
        # sigma(R=0) was normalised to 2.5 by Parseval in k-space; here it is measured in real space
        assert np.sqrt(tv[-1]) == pytest.approx(2.5, rel=1e-10)
        assert tv[0] < tv[1] < tv[2]
        pdf = f.Fmax_PDF()
        assert int(pdf.sum()) == n ** 3
        # Laplacian identity at R=0: H11+H22+H33 = delta, so its mean square is TrueVariance
        f.compute_second_derivatives(0.0)
        tr = f.second_derivative(0) + f.second_derivative(1) + f.second_derivative(2)
        assert np.mean(tr ** 2) == pytest.approx(tv[-1], rel=1e-11)
        assert abs(tr.mean()) < 1e-12
        # displacement divergence: Zel'dovich displacement is -grad phi, so sum_a d_a Vel_a = -delta;
        # checked in k-space through the transforms of the library itself
        f.compute_displacements(1, 0)
        p = f.products()
        assert np.isfinite(p["Vel"]).all() and np.isfinite(p["Vel_3LPT_2"]).all()
        kx, ky, kz = synth.kgrid(n)
        div = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
        for a, k in enumerate((kx[:, None, None], ky[None, :, None], kz[None, None, :])):
            div += 1j * k * f.forward_transform(p["Vel"][..., a].astype(np.float64))
        lap = f.reverse_transform(div)
        # Nyquist planes break the identity (k = +pi has no -pi partner); the input has none
        assert np.sqrt(np.mean((lap + tr) ** 2)) < 1e-5 * np.sqrt(tv[-1])  # fp32 storage of Vel
        rmax = p["Rmax"]
        assert rmax.min() >= 0 and rmax.max() <= 2


def test_properties_on_the_box_of_the_metric(api):
    """1024^3 fp64 (BASELINE.json's box, 226 GB on the device): what can be checked without the oracle and without the
    60 GB of host products -- normalisation, monotone variances, the histogram, determinism of a repeated sweep, the
    running maximum over the radii cell by cell (single FMAX / RMAX blocks, 4.3 GB each), and linearity"""
    n = 1024
    x, y = synth.invgrow_table("lcdm")
    radii = synth.radii_ladder(12)[[2, 8, 11]]          # one band-limited radius, one full, R = 0
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        tv1 = f.sweep(radii[:1])
        fm1 = f.block("FMAX")
        tv = f.sweep(radii)
        assert tv[0] == tv1[0]
        assert np.sqrt(tv[-1]) == pytest.approx(2.5, rel=1e-10) and tv[0] < tv[1] < tv[2]
        pdf = f.Fmax_PDF()
        assert int(pdf.sum()) == n ** 3
        fm3 = f.block("FMAX")
        rm3 = f.block("RMAX")
        assert np.isfinite(fm3).all() and (fm3 >= fm1).all()       # running maximum
        moved = rm3 > 0
        assert rm3.min() >= -1 and rm3.max() == 2 and int((rm3 < 0).sum()) <= 100    # -1: the eigen-solver sentinel, rare
        assert fm3[rm3 >= 0].min() >= 0.0
        # a cell moves when the new F exceeds the stored float as doubles (quirk Q2): the float it stores may be the same one
        assert np.array_equal(fm3[~moved], fm1[~moved]) and float(np.mean(fm3[moved] > fm1[moved])) > 0.999
        del fm1, moved
        tv_again = f.sweep(radii)
        assert np.array_equal(tv_again, tv) and np.array_equal(f.block("FMAX"), fm3) and np.array_equal(f.block("RMAX"), rm3)
        collapsed = int((fm3 >= 1.0).sum())
        assert 0.3 * n ** 3 < collapsed < 0.7 * n ** 3   # sigma = 2.5: about half of the cells collapse by z = 0
        del fm3, rm3
        f.synth_density(synth.SEED, 5.0, -2.0)           # the same modes, twice the amplitude
        assert f.sweep(radii[1:]) == pytest.approx(4.0 * tv[1:], rel=1e-12)


@pytest.mark.parametrize("n", [16, 64, 256, 24, 40, 200])   # (24, 40, 200: k_mixed_c2r_invariants)
def test_invariant_zpass_equals_six_component_path_fp32_fields(api, n, monkeypatch):
    """the same with fp32 fields: the invariants are formed in fp64 from the fp32 components the transforms produce, exactly
    as the six-component solve forms them, and kept as fp64 rows of their own; the 3LPT(b) contraction inside the z-pass
    likewise (PF_LPT_FUSE).  Every product bit for bit."""
    dk = synth.make_density(n, seed=5 + n)
    dk[0, 0, 0] = -0.11 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([n / 16.0, n / 40.0, 1.5, 0.6, 0.0])
    out = {}
    on = "1" if n & (n - 1) == 0 else "2"   # (sizes that are not a power of two take six fp32 components by default: "2" asks for the invariants)
    for mode in ("0", on):
        monkeypatch.setenv("PF_INVARIANTS", mode)
        monkeypatch.setenv("PF_LPT_FUSE", mode)
        with api.Fmax(n, field_bytes=4, timing=True) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.set_growth(synth.growth_multipliers())
            f.reset_kernel_stats()
            tv = f.compute_fmax(radii, do_lpt=True)
            classes = {k["name"] for k in f.kernel_stats()}
            assert ("zpass_c2r_hess_6to3inv" in classes) == (mode != "0") and ("zpass_c2r_hess_6_lpt3b" in classes) == (mode != "0"), classes
            out["0" if mode == "0" else "1"] = (tv, f.products(), f.Fmax_PDF())
    assert np.array_equal(out["0"][0], out["1"][0])
    for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(out["0"][1][name], out["1"][1][name]), name
    assert np.array_equal(out["0"][2], out["1"][2])


@pytest.mark.parametrize("n", [16, 64, 128, 256, 24, 96, 200])   # (24, 96, 200: k_mixed_c2r_invariants)
def test_invariant_zpass_equals_six_component_path(api, n, monkeypatch):
    """Default sweep: for every radius but the last the z-pass stores the three invariants of the tensor (k_c2r_invariants)
    and the solve starts from them; PF_INVARIANTS=0 keeps six components throughout.  The component values and the
    invariants are formed by the same operations in both, so TrueVariance, Fmax, Rmax, the histogram and (from the last
    radius's Hessian, which both keep) the displacements agree bit for bit.  Band-limited radii included."""
    dk = synth.make_density(n, seed=3 + n)
    dk[0, 0, 0] = 0.37 * n ** 3        # a DC mode: added to every component after the transform
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([n / 16.0, n / 40.0, 1.5, 0.6, 0.0])
    out = {}
    for mode in ("0", "2"):   # ("2": the invariant z-pass wherever it exists -- the run-time plans of 24 and 96 points would not take it by themselves)
        monkeypatch.setenv("PF_INVARIANTS", mode)
        monkeypatch.setenv("PF_LPT_FUSE", mode)
        with api.Fmax(n, timing=True) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.set_growth(synth.growth_multipliers())
            f.reset_kernel_stats()
            tv = f.compute_fmax(radii, do_lpt=True)
            assert ("zpass_c2r_hess_6to3inv" in {k["name"] for k in f.kernel_stats()}) == (mode != "0")
            out["0" if mode == "0" else "1"] = (tv, f.products(), f.Fmax_PDF())
    assert np.array_equal(out["0"][0], out["1"][0])
    for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(out["0"][1][name], out["1"][1][name]), name
    assert np.array_equal(out["0"][2], out["1"][2])
    assert (out["1"][1]["Rmax"] > 0).any() and (out["1"][1]["Rmax"] < 4).any()


@pytest.mark.parametrize("n,fb", [(16, 8), (64, 8), (256, 8), (64, 4), (256, 4)])
def test_solve_beside_the_next_zpass_equals_the_in_line_order(api, n, fb, monkeypatch):
    """Default sweep: the collapse solve of radius i runs on its own stream beside the z-pass of radius i + 1, the passes of
    consecutive radii alternating between two field sets (fp32 fields: two sets of invariant rows); PF_SOLVE_BESIDE_Z=0 runs every
    kernel in line on one stream.  The same kernels on the same grids in the same order of the running maximum: TrueVariance,
    Fmax, Rmax, the histogram and the displacements bit for bit, over two consecutive steps of one context (the second step
    reuses both field sets), with two, three and six radii (zero, one and four solves that have a z-pass to run beside)."""
    dk = synth.make_density(n, seed=29 + n)
    dk[0, 0, 0] = 0.21 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    ladders = [np.array([1.5, 0.0]), np.array([n / 40.0, 1.5, 0.0]), np.array([n / 16.0, n / 40.0, 1.5, 0.9, 0.6, 0.0])]
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PF_SOLVE_BESIDE_Z", mode)
        res = []
        with api.Fmax(n, field_bytes=fb) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.set_growth(synth.growth_multipliers())
            for radii in ladders + ladders[-1:]:
                tv = f.compute_fmax(radii, do_lpt=True)
                beside = int(f.L.pf_solve_ran_beside_zpass(f.h))
                assert beside == (1 if mode == "1" and len(radii) > 2 else 0), (mode, len(radii), beside)
                res.append((tv, f.products(), f.Fmax_PDF()))
        out[mode] = res
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a[0], b[0])
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.array_equal(a[1][name], b[1][name]), name
        assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("n", [16, 64, 256])
def test_fused_3lpt_source_equals_separate_kernels(api, n, monkeypatch):
    """Default: the z-pass of the 2LPT potential's Hessian contracts its six components with the first-order Hessian into the
    3LPT(b) source on the fly (k_c2r_invariants, MODE 1) and stores none of them; PF_LPT_FUSE=0 stores six fields and runs
    k_lpt_accum.  One per-cell function serves both: every product column bit for bit."""
    dk = synth.make_density(n, seed=11 + n)
    dk[0, 0, 0] = -0.2 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([2.0, 0.0])
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PF_LPT_FUSE", mode)
        with api.Fmax(n) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.set_growth(synth.growth_multipliers())
            f.compute_fmax(radii, do_lpt=True)
            out[mode] = f.products()
    for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(out["0"][name], out["1"][name]), name
    assert np.abs(out["1"]["Vel_3LPT_2"]).max() > 0


@pytest.mark.parametrize("fb", [8, 4])
@pytest.mark.parametrize("n", [16, 64, 256])
def test_sources_formed_by_the_last_solve_equal_the_separate_kernel(api, n, fb):
    """compute_fmax (sweep + compute_displacements(1, 0) back to back, src/fmax.c:150-163): the collapse pass of the last radius
    also writes the three LPT sources of src/LPT.c:64-93 from the six components it holds (k_collapse_src) and
    k_lpt_sources is not run; sweep() and compute_displacements(1, 0) called apart run the separate kernel.  One per-cell
    function, the same grid and walk for the sum of S2: the LPT spectra and every product column bit for bit.  Afterwards the
    separate kernel must still serve a second compute_displacements(1, 0) and one after new second derivatives."""
    dk = synth.make_density(n, seed=21 + n)
    dk[0, 0, 0] = 0.1 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([2.0, 0.7, 0.0])
    out = {}
    for mode in ("apart", "together"):
        with api.Fmax(n, field_bytes=fb, timing=True) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.set_growth(synth.growth_multipliers())
            f.reset_kernel_stats()
            if mode == "apart":
                tv = f.sweep(radii)
                f.compute_displacements(1, 0)
            else:
                tv = f.compute_fmax(radii, do_lpt=True)
            classes = {k["name"] for k in f.kernel_stats()}
            assert ("collapse_lpt_sources" in classes) == (mode == "together") and ("lpt_sources" in classes) == (mode == "apart"), classes
            out[mode] = (tv, f.products(), [f.kvector(w) for w in (0, 1, 2)])
            if mode == "together":
                f.compute_displacements(1, 0)              # again: the sources are gone (transformed in place), the kernel runs
                again = f.products()
                f.compute_second_derivatives(0.0)
                f.compute_displacements(1, 0)
                after = f.products()
    assert np.array_equal(out["apart"][0], out["together"][0])
    for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(out["apart"][1][name], out["together"][1][name]), name
        assert np.array_equal(out["apart"][1][name], again[name]) and np.array_equal(out["apart"][1][name], after[name]), name
    for a, b in zip(out["apart"][2], out["together"][2]):
        assert np.array_equal(a, b)
    assert np.abs(out["together"][1]["Vel_2LPT"]).max() > 0


def test_lower_lpt_orders_of_a_build_without_three_lpt_or_two_lpt(api):
    """The reference picks the order of the displacements at compile time (-DTWO_LPT, -DTHREE_LPT: src/fmax.c:300-336,
    src/LPT.c:30, 78, 113, 214).  pf_set_lpt_order(2): the 2LPT source and its displacement only -- Vel and Vel_2LPT bit for
    bit those of the full run, no Hessian of the 2LPT potential, no 3LPT passes; (1): Zel'dovich only, no second
    derivatives on re-entry.  The columns of the orders left out are zero.  Fmax / Rmax do not depend on it."""
    n = 32
    dk = synth.make_density(n, seed=5)
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([1.5, 0.0])

    def run(order, entry):
        with api.Fmax(n, timing=True) as f:
            f.set_density(dk)
            return body(f, order, entry)

    def body(f, order, entry):
        f.set_invgrow(x, y)
        f.set_growth(synth.growth_multipliers())
        f.set_lpt_order(order)
        f.reset_kernel_stats()
        if entry == "compute_fmax":
            f.compute_fmax(radii, do_lpt=True)
        else:                                           # pinocchio.x parameterfile 3: compute_displacements(1, 1, z)
            f.compute_displacements(1, 1)
        return f.products(), {k["name"] for k in f.kernel_stats()}

    for entry in ("compute_fmax", "snapshot"):
        full, _ = run(3, entry)
        two, cls2 = run(2, entry)
        one, cls1 = run(1, entry)
        for name in ("Fmax", "Rmax", "Vel"):
            assert np.array_equal(full[name], two[name]) and np.array_equal(full[name], one[name]), (entry, name)
        assert np.array_equal(full["Vel_2LPT"], two["Vel_2LPT"]) and np.abs(two["Vel_2LPT"]).max() > 0
        assert not one["Vel_2LPT"].any()
        for name in ("Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.abs(full[name]).max() > 0 and not two[name].any() and not one[name].any(), (entry, name)
        assert "zpass_c2r_hess_6_lpt3b" not in cls2 and "lpt_accum" not in cls2
        assert not (cls1 & {"lpt_sources", "collapse_lpt_sources", "zpass_r2c", "xpass_fwd"})
        if entry == "snapshot":
            assert "xpass_hess_1to3" in cls2 and "xpass_hess_1to3" not in cls1


def test_pruned_transform_equals_full_transform(api, monkeypatch):
    """Smoothed radii use a pruned FFT: modes whose Gaussian weight is < 2^-60 are not transformed.
    Against the full transform (PF_PRUNE_EPS=0) the Hessian changes by less than its own rounding."""
    n = 128
    dk = synth.make_density(n, seed=99)
    out = {}
    for eps in ("0", None):
        if eps is None:
            monkeypatch.delenv("PF_PRUNE_EPS", raising=False)
        else:
            monkeypatch.setenv("PF_PRUNE_EPS", eps)
        with api.Fmax(n) as f:
            f.set_density(dk)
            res = []
            for rs in (16.0, 8.0, 4.0, 2.0):   # bands 12, 24, 47, off
                f.compute_second_derivatives(rs)
                res.append([f.second_derivative(i) for i in range(6)])
            out[eps] = res
    for full, pruned in zip(out["0"], out[None]):
        amp = max(np.max(np.abs(h)) for h in full)
        for a, b in zip(full, pruned):
            assert np.max(np.abs(a - b)) <= 4e-16 * amp


def test_hmf_validation_run_on_gpu(api):
    """The reference's own committed validation run (HMF_Validation: 128^3, seed 486604, 9 radii) through the HIP
    path: same per-radius sigma (4 logged decimals), collapsed-cell count and Fmax histogram as the reference's
    output files.  The density comes from the restated IC generator (tests/ic_oracle.py, pinned by the same data)."""
    import ic_oracle
    with open(os.path.join(GOLD, "hmf_validation_kat.json")) as fh:
        kat = json.load(fh)
    p = kat["params"]
    n = p["GridSize"]
    box = p["BoxSize_h100"] / p["Hubble100"]
    dk = ic_oracle.genic(n, box, p["RandomSeed"], kat["PkNorm"], p)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    radii_cells = np.array(kat["radii_Mpc"]) / (box / n)
    with api.Fmax(n) as f:
        f.set_density(dk)
        f.set_invgrow(x, y)
        tv = f.sweep(radii_cells)
        pdf = f.Fmax_PDF().astype(np.int64)
    assert np.all(np.abs(np.sqrt(tv) - np.array(kat["computed_sigma"])) <= 6e-5)
    want = np.array(kat["FmaxPDF"], dtype=np.int64)
    assert abs(int(pdf[10:].sum()) - kat["collapsed"]) <= 5
    assert np.abs(pdf - want).sum() <= 200 and np.max(np.abs(pdf - want)) <= 20


def test_genic_on_device_vs_oracle(api):
    """pf_genic_density (GenIC_large on the device) against the restated CPU generator, same seed and cosmology"""
    import ic_oracle
    with open(os.path.join(GOLD, "hmf_validation_kat.json")) as fh:
        p = json.load(fh)["params"]
    for n, seed in ((32, 7), (64, p["RandomSeed"])):
        box = float(n) / p["Hubble100"]
        pkn = 1.7e7
        want = ic_oracle.genic(n, box, seed, pkn, p)
        with api.Fmax(n) as f:
            f.genic_density(seed, box, p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"], pknorm=pkn)
            got = f.density()
        assert np.count_nonzero(want) > n ** 3 // 8
        assert np.array_equal(got == 0, want == 0)                       # same blind points and Nyquist planes
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))  # device libm vs glibc in log/cos/sin/pow


@pytest.mark.parametrize("spectrum,wdm", [("Efstathiou", 0.0), ("PowerLaw", 0.0), ("EH", 2.5), ("Efstathiou", 1.0)])
def test_genic_other_spectra_on_device_vs_oracle(api, spectrum, wdm):
    """the other forms of PowerSpectrum() (src/cosmo.c:953-1007): the Efstathiou fit, a power law, and the warm-dark-matter
    cut-off that multiplies any of them -- device generator against the restated CPU generator, and the sigma8
    normalisation (pf_pk_norm) against a scipy quadrature of the same integrand"""
    import ic_oracle
    with open(os.path.join(GOLD, "hmf_validation_kat.json")) as fh:
        p = dict(json.load(fh)["params"])
    if spectrum == "PowerLaw":
        p["PrimordialIndex"] = -1.5
    n, seed = 32, 11
    box = float(n) / p["Hubble100"]
    with api.Fmax(n) as f:
        pkn = f.genic_density(seed, box, p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"], sigma8=0.8,
                              spectrum=spectrum, wdm_mass_kev=wdm)
        got = f.density()
    want = ic_oracle.genic(n, box, seed, pkn, p, spectrum=spectrum, wdm_mass_kev=wdm)
    assert np.count_nonzero(want) > n ** 3 // 8
    assert np.array_equal(got == 0, want == 0)
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    # the normalisation: sigma8^2 over the top-hat variance at 8/h Mpc of the un-normalised spectrum
    from scipy.integrate import quad
    import ctypes as C
    L = ic_oracle._lib()
    L.orc_power_spectrum_form.restype = C.c_double
    L.orc_power_spectrum_form.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    cos = ic_oracle.Cosmo(p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"])
    R = 8.0 / p["Hubble100"]

    def integrand(logk):
        k = np.exp(logk)
        kr = k * R
        w = 1.0 if kr < 1e-5 else 3.0 * (np.sin(kr) / kr ** 3 - np.cos(kr) / kr ** 2)
        pw = L.orc_power_spectrum_form(k, ic_oracle.SPECTRUM_CODES[spectrum], C.cast(C.byref(cos), C.c_void_p), 0, None, None, None, wdm, 3.085678e24)
        return pw * w * w * k ** 3 / (2.0 * np.pi ** 2)

    val, _ = quad(integrand, -10.0, np.log(500.0 / R), epsabs=0, epsrel=1e-10, limit=2000)
    assert pkn == pytest.approx(0.8 ** 2 / val, rel=1e-6)


def test_hmf_validation_run_entirely_on_device(api):
    """seed + cosmology -> pf_genic_density -> pf_sweep -> Fmax PDF, nothing but parameters from the host: the
    reference's committed validation numbers again (sigma per radius, collapsed count, histogram)"""
    import ic_oracle
    with open(os.path.join(GOLD, "hmf_validation_kat.json")) as fh:
        kat = json.load(fh)
    p = kat["params"]
    n = p["GridSize"]
    box = p["BoxSize_h100"] / p["Hubble100"]
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    radii_cells = np.array(kat["radii_Mpc"]) / (box / n)
    with api.Fmax(n) as f:
        f.genic_density(p["RandomSeed"], box, p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"],
                        pknorm=kat["PkNorm"])
        f.set_invgrow(x, y)
        tv = f.sweep(radii_cells)
        pdf = f.Fmax_PDF().astype(np.int64)
        # the library's own normalisation (Gauss-Legendre to 1e-10; the reference's QAGS runs at 1e-4)
        assert f.genic_density(p["RandomSeed"], box, p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"],
                               sigma8=p["Sigma8"]) == pytest.approx(kat["PkNorm"], rel=1e-5)
    assert np.all(np.abs(np.sqrt(tv) - np.array(kat["computed_sigma"])) <= 6e-5)
    want = np.array(kat["FmaxPDF"], dtype=np.int64)
    assert abs(int(pdf[10:].sum()) - kat["collapsed"]) <= 5
    assert np.abs(pdf - want).sum() <= 200


def test_scale_dependent_build_vs_oracle(api):
    """rows f-3: what a -DSCALE_DEPENDENT build changes on the path -- one inverse-growth spline per smoothing radius
    (SPLINE_INVGROW[ismooth], src/cosmo.c:1828) and growth multipliers interpolated per mode in the 10 k-bins of
    InterpolateGrowth (src/cosmo.c:1728-1755) with |k| in rad/cell (src/fmax-pfft.c:315-359)"""
    n = 32
    radii = np.array([3.0, 1.5, 0.0])
    dk = synth.make_density(n, seed=21)
    splines = [synth.invgrow_table("lcdm", omega0=om) for om in (0.25, 0.30, 0.35)]
    g = synth.growth_multipliers()
    j = np.arange(10)
    # grid |k| spans 0.2 .. 5.4 rad/cell: bins 4..8 of 10^(-3 + 0.5 j); every order gets its own k-dependence
    tabs = [np.log10(abs(g[o]) * (1.0 + 0.04 * (o + 1) * j)) for o in range(4)]
    signs = [1.0, 1.0, -1.0, 1.0]
    with api.Fmax(n) as f:
        f.set_density(dk)
        for i, (x, y) in enumerate(splines):
            f.set_invgrow(x, y, ismooth=i)
        for o in range(4):
            f.set_growth_table(o + 1, tabs[o], sign=signs[o])
        tv = f.compute_fmax(radii, do_lpt=True)
        p = f.products()
        f.set_growth_table(1, [])                       # back to the scalar for the Zel'dovich order
        f.set_growth(g)
        f.compute_displacements(0, 0)
        p_scalar = f.products()
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    o.set_invgrow(*splines[0])
    for i, (x, y) in enumerate(splines):
        o.set_invgrow_radius(i, x, y)
    for k in range(4):
        o.set_growth_table(k + 1, tabs[k], sign=signs[k])
    tv_o = o.compute_fmax(radii, do_lpt=True)
    po = o.products()
    assert np.allclose(tv, tv_o, rtol=1e-12)
    _fmax_close(p["Fmax"], po["Fmax"])
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * np.max(np.abs(po[name])), name
    # the per-radius splines and the k-dependence really were in play
    o1 = oracle_lib.Oracle(n, 0)
    o1.set_density(dk); o1.set_invgrow(*splines[0]); o1.set_growth(g)
    o1.compute_fmax(radii, do_lpt=True)
    p1 = o1.products()
    assert np.mean(p1["Fmax"] != po["Fmax"]) > 0.1
    assert np.max(np.abs(p1["Vel"] - po["Vel"])) > 1e-2 * np.max(np.abs(po["Vel"]))
    assert np.max(np.abs(p_scalar["Vel"].astype(np.float64) - p1["Vel"])) <= 4e-7 * np.max(np.abs(p1["Vel"]))
    assert np.array_equal(p_scalar["Vel_2LPT"], p["Vel_2LPT"])


def test_update_products_merges_into_recompute_records(api):
    """pf_update_products writes only the named columns: a 104-byte RECOMPUTE_DISPLACEMENTS record keeps its *_prev
    copies, Fmax and Rmax (src/pinocchio.h:233-259; shift_all_displacements src/fragment.c:832-850)"""
    from pinocchio_amd import _lib
    n = 32
    rec = np.dtype([("Rmax", "<i4"), ("Fmax", "<f4"), ("Vel", "<f4", 3), ("Vel_2LPT", "<f4", 3), ("Vel_3LPT_1", "<f4", 3),
                    ("Vel_3LPT_2", "<f4", 3), ("Vel_prev", "<f4", 3), ("Vel_2LPT_prev", "<f4", 3),
                    ("Vel_3LPT_1_prev", "<f4", 3), ("Vel_3LPT_2_prev", "<f4", 3)])
    assert rec.itemsize == 104
    dk = synth.make_density(n, seed=8)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
        f.compute_fmax(np.array([2.0, 0.0]), do_lpt=True)
        p0 = f.products()
        full = _lib.ProductLayout(104, 0, 4, 8, 20, 32, 44)
        recs = np.zeros((n, n, n), dtype=rec)
        f._chk(f.L.pf_get_products(f.h, recs.ctypes.data_as(C.c_void_p), C.byref(full)))
        for name in ("Rmax", "Fmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.array_equal(recs[name], p0[name])
        # the host shifts (fragment.c), the device recomputes at another redshift, only Vel* come back
        for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            recs[name + "_prev"] = recs[name]
        recs["Fmax"] += 100.0
        f.set_growth(g * 0.5)
        f.compute_displacements(0, 0)
        f.update_products(recs, _lib.ProductLayout(104, -1, -1, 8, 20, 32, 44))
        p1 = f.products()
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(recs[name], p1[name]) and np.array_equal(recs[name + "_prev"], p0[name])
        assert np.allclose(p1[name], 0.5 * p0[name], rtol=2e-7, atol=0)
    assert np.array_equal(recs["Fmax"], p0["Fmax"] + np.float32(100.0)) and np.array_equal(recs["Rmax"], p0["Rmax"])


def test_preflight_refuses_a_box_the_device_cannot_hold(api, capfd):
    """pf_create compares the memory plan with hipMemGetInfo BEFORE it allocates anything and fails in the reference's format (2048^3 with
    fp64 fields on one rank: 1.8 TB); PF_PREFLIGHT=0 lets the allocations find out themselves"""
    with pytest.raises(api.PinfmaxError) as e:
        api.Fmax(2048)
    assert "needs" in str(e.value) and "GB of device memory" in str(e.value)
    assert "ERROR on task 0: pf_create: 2048^3 on 1 ranks" in capfd.readouterr().out
    with api.Fmax(64) as f:      # nothing of the refused context stayed behind
        assert f.device_bytes > 0


@pytest.mark.parametrize("env", [{"PF_HANDOFF_THREADS": "3", "PF_HANDOFF_CHUNK_MB": "1"}, {"PF_HOST_REGISTER": "1"}, {"PF_HANDOFF_THREADS": "1"}])
def test_handoff_variants_move_the_same_bytes(api, env, monkeypatch):
    """the host side of the boundary (PfHandoff, csrc/pf_api.hip) with other piece sizes and thread counts, and with the caller's array
    registered with the driver: products, a re-entrant update into 104-byte records, blocks and the density come out as by default"""
    from pinocchio_amd import _lib
    n = 64
    dk = synth.make_density(n, seed=21)
    x, y = synth.invgrow_table("lcdm")
    rec = np.dtype([("Rmax", "<i4"), ("Fmax", "<f4"), ("Vel", "<f4", 3), ("Vel_2LPT", "<f4", 3), ("Vel_3LPT_1", "<f4", 3),
                    ("Vel_3LPT_2", "<f4", 3), ("prev", "<f4", 12)])

    def run():
        with api.Fmax(n) as f:
            f.set_density(dk); f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
            f.compute_fmax(np.array([2.0, 0.0]), do_lpt=True)
            p = f.products()
            recs = np.full((n, n, n), 7, dtype=np.uint8).repeat(104, axis=2).view(rec).reshape(n, n, n)   # every byte of the records set
            f.update_products(recs, _lib.ProductLayout(104, -1, 4, 8, 20, 32, 44))                           # (Fmax named too, Rmax not)
            return p, recs.copy(), f.block("2LPT"), f.block("ID  ", 8), f.density()

    base = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = run()
    for a, b in zip(base, got):
        assert a.tobytes() == b.tobytes()
    p, recs = got[0], got[1]
    assert np.array_equal(recs["Fmax"], p["Fmax"]) and np.array_equal(recs["Vel_3LPT_2"], p["Vel_3LPT_2"])
    assert np.all(np.ascontiguousarray(recs["Rmax"]).view(np.uint8) == 7) and np.all(np.ascontiguousarray(recs["prev"]).view(np.uint8) == 7)   # what was not named kept its bytes


def test_fft_module_seam_single_components(api):
    """pf_derivative = compute_derivative(ThisGrid, a, b) (src/fmax-pfft.c:255-441) for every (a, b) the reference can be
    called with: second derivatives, first derivatives (re/im swap), the potential (-1/k^2), with smoothing and growth"""
    import np_restatement as npr
    n = 32
    dk = synth.make_density(n, seed=77)
    dk[0, 0, 0] = 0.37 * n ** 3            # the k = 0 mode is left untouched by the filter
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    hes_o = o.second_derivatives(1.3)
    with api.Fmax(n) as f:
        f.set_growth(g)
        for (a, b, rs, order) in [(1, 1, 1.3, 0), (2, 2, 1.3, 0), (3, 3, 1.3, 0), (1, 2, 1.3, 0), (1, 3, 1.3, 0), (2, 3, 1.3, 0),
                                  (2, 1, 0.0, 0), (1, 0, 0.0, 1), (2, 0, 0.0, 2), (3, 0, 0.7, 3), (0, 3, 0.0, 4), (0, 0, 0.0, 0),
                                  (0, 0, 2.0, 1)]:
            got = f.compute_derivative(dk, a, b, rs, order)
            growth = g[order - 1] if order else 1.0
            want = npr.derivative(dk, rs, a, b, growth)
            if a == 0 and b == 0:            # greens_function returns -1/k^2 for (0,0) (src/fmax-pfft.c:449-450)
                want = 2.0 * dk[0, 0, 0].real / n ** 3 - want
            assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want)), (a, b, rs, order)
            if order == 0 and rs == 1.3:
                ider = a if a == b else a + b + 1
                assert np.max(np.abs(got - hes_o[ider - 1])) <= 1e-12 * np.max(np.abs(hes_o[ider - 1]))
        with pytest.raises(api.PinfmaxError):
            f.compute_derivative(dk, 4, 0)


def test_select_sorted_and_snapshot_blocks(api):
    """row f-2: what the consumers of `products` read -- the fragmentation order (Fmax >= Flast by descending Fmax,
    src/distribute.c:695, src/fragment.c:484-503) and the blocks of the timeless snapshot (src/write_snapshot.c:207-342)
    -- produced on the device; checked on the device's own products (exact) and against the oracle's ordering"""
    n = 64
    radii = np.array([4.0, 2.0, 1.0, 0.0])
    dk = synth.make_density(n, seed=3)
    x, y = synth.invgrow_table("lcdm")
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(synth.growth_multipliers())
    o.compute_fmax(radii, do_lpt=True)
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
        f.compute_fmax(radii, do_lpt=True)
        p = f.products()
        F = p["Fmax"].ravel()
        for flast in (1.0, 2.5, -20.0, 1e9):
            idx, fs = f.select_sorted(flast)
            sel = np.flatnonzero(F >= np.float32(flast))
            order = sel[np.lexsort((sel, -F[sel].astype(np.float64)))]
            assert np.array_equal(idx, order.astype(np.uint32)) and np.array_equal(fs, F[order]), flast
        idx, fs = f.select_sorted(1.0)
        blocks = {k: f.block(k) for k in ("ID  ", "FMAX", "RMAX", "ZEL ", "2LPT", "31PT", "32PT")}
        id8 = f.block("ID  ", id_bytes=8)
        with pytest.raises(api.PinfmaxError):
            f.block("XXXX")
    # same selection as the reference path's: the oracle's order differs only where its Fmax differs (fp32 ulps)
    io, fo = o.select_sorted(1.0)
    assert abs(len(io) - len(idx)) <= 2 and np.all(np.diff(fs) <= 0)
    m = min(len(io), len(idx))
    assert np.mean(io[:m] == idx[:m]) > 0.99
    assert np.array_equal(blocks["ID  "], 1 + np.arange(n ** 3, dtype=np.uint32)) and np.array_equal(id8, blocks["ID  "].astype(np.uint64))
    assert np.array_equal(blocks["FMAX"], F) and np.array_equal(blocks["RMAX"], p["Rmax"].ravel())
    for name, col in (("ZEL ", "Vel"), ("2LPT", "Vel_2LPT"), ("31PT", "Vel_3LPT_1"), ("32PT", "Vel_3LPT_2")):
        assert np.array_equal(blocks[name], p[col].reshape(-1, 3))


def test_tabulated_ct_build_vs_oracle(api):
    """row f-4, TABULATED_CT build (src/collapse_times.c:780-1231): the table of ell() made on the device, the node
    splines and the interpolating collapse-time pass against the oracle's restatement (there is no reference output
    for this build option; the restatement is checked against scipy in tests/test_oracle.py)"""
    n = 64
    dk = synth.make_density(n, seed=41)
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([3.0, 1.5, 0.0])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y)
    var = o.compute_fmax(radii, do_lpt=False) * 1.1          # stands in for Smoothing.Variance (the expected variance)
    direct = o.products()
    o.set_tabulated_ct(var)
    tv_o = o.compute_fmax(radii, do_lpt=False)
    po = o.products()
    tab_o, dv = o.ct_build(1, var[1])
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y)
        tab = f.ct_build(1, var[1])
        # table entries: the device's libm (and the sincos / cbrt / exp10 forms) against glibc
        nz = (tab_o != 0) & (tab != 0)
        assert np.mean((tab_o != 0) != (tab != 0)) < 1e-4 and nz.mean() > 0.3
        err = np.abs(tab[nz] - tab_o[nz]) / np.maximum(1.0, tab_o[nz])
        # the regular grid puts ~0.1 % of its nodes next to the surface den = 0, the ill-conditioned branch of the
        # cubic, where one ulp in the libm calls moves ell() by up to 1e-5
        assert np.mean(err > 1e-11) < 3e-3 and np.mean(err > 1e-7) < 3e-4 and err.max() < 2e-3, (np.mean(err > 1e-11), np.mean(err > 1e-7), err.max())
        f.set_tabulated_ct(var)
        tv = f.sweep(radii)
        p = f.products()
        assert np.allclose(tv, tv_o, rtol=1e-12)
        _fmax_close(p["Fmax"], po["Fmax"])
        assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
        assert np.mean(po["Fmax"] != direct["Fmax"]) > 0.5   # the table really replaced the direct solve
        # the reference's own order of calls with a table read from a file (params.CTtableFile):
        # compute_second_derivatives; initialize_collapse_times; compute_collapse_times
        f.set_tabulated_ct([])
        fm = None
        for i, r in enumerate(radii):
            f.compute_second_derivatives(r)
            t_i, _ = o.ct_build(i, var[i])
            f.ct_load(i, var[i], t_i)
            f.compute_collapse_times(i)
        p2 = f.products()
    # same table bit for bit on both sides: only the Hessian's rounding is left
    ulp = np.spacing(np.maximum(np.abs(po["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.mean(np.abs(p2["Fmax"].astype(np.float64) - po["Fmax"]) > 2 * ulp) < 2e-5
    assert np.mean(p2["Rmax"] != po["Rmax"]) < 1e-3


@pytest.mark.parametrize("n", [32, 64])
def test_double_precision_products_build_vs_oracle(api, n):
    """-DDOUBLE_PRECISION_PRODUCTS (src/Makefile:68; PRODFLOAT double, src/pinocchio.h:219-225): Fmax is kept and compared in
    fp64 -- the running maximum is no longer rounded to fp32 between radii (src/collapse_times.c:587-590) -- and the Vel*
    fields take the doubles the transforms produce.  PF_FLAG_DOUBLE_PRODUCTS against the oracle built with the same flag
    (112-byte records); the sweep with invariants, the six-component route, the sources formed by the last solve."""
    dk = synth.make_density(n, seed=61 + n)
    dk[0, 0, 0] = 0.05 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([n / 16.0, 1.5, 0.7, 0.0])
    o = oracle_lib.Oracle(n, 0, double_products=True)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=True)
    po = o.products()
    assert po.dtype.itemsize == 112 and po["Fmax"].dtype == np.float64
    o32 = oracle_lib.Oracle(n, 0)
    o32.set_density(dk); o32.set_invgrow(x, y); o32.set_growth(g)
    o32.compute_fmax(radii, do_lpt=True)
    p32 = o32.products()
    with api.Fmax(n, double_products=True) as f:
        f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        p = f.products()
        pdf = f.Fmax_PDF()
        with pytest.raises(api.PinfmaxError):
            f.select_sorted(1.0)                       # fp32 by definition
    assert p.dtype.itemsize == 112 and np.allclose(tv, tv_o, rtol=1e-12)
    # Fmax in fp64: the default libm flavour against glibc, as in the fp32 build but without the fp32 rounding on top
    rel = np.abs(p["Fmax"] - po["Fmax"]) / np.maximum(1.0, np.abs(po["Fmax"]))
    assert np.mean(rel > 1e-10) < 1e-3 and np.quantile(rel, 0.99) < 1e-12, (np.mean(rel > 1e-10), np.quantile(rel, 0.99))
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    assert np.abs(pdf.astype(np.int64) - o.fmax_pdf().astype(np.int64)).sum() <= 4
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        amp = np.max(np.abs(po[name]))
        assert amp > 0 and np.max(np.abs(p[name] - po[name])) <= 1e-12 * amp, name      # fp64 all the way: no 6e-8 floor
    # and they are the fp32 build's numbers before rounding
    assert np.max(np.abs(p["Vel"].astype(np.float32) - p32["Vel"])) <= np.spacing(np.float32(np.abs(p32["Vel"]).max()))
    assert not np.array_equal(p["Fmax"].astype(np.float32).astype(np.float64), p["Fmax"])


@pytest.mark.parametrize("flavour", [1, 2])
def test_trilinear_and_all_spline_table_interpolation_vs_oracle(api, flavour):
    """a build with -DTRILINEAR (1) or -DALL_SPLINE (2) in OPTIONS (src/collapse_times.c:1153-1216, the three choices of
    tests/Readme_Pinocchio_tests_V5_1.txt): the same table, read with eight entries or with sixteen node splines and
    gsl_spline2d's bicubic.  The oracle's table is loaded on both sides, so only the Hessian's rounding is left."""
    n = 32
    dk = synth.make_density(n, seed=45)
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([2.0, 0.0])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y)
    var = o.compute_fmax(radii, do_lpt=False)
    o.set_tabulated_ct(var)
    o.compute_fmax(radii, do_lpt=False)
    p0 = o.products()                                   # BILINEAR_SPLINE
    o.set_ct_interpolation(flavour)
    tv_o = o.compute_fmax(radii, do_lpt=False)
    po = o.products()
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y)
        f.set_ct_interpolation(flavour)
        for i, r in enumerate(radii):
            f.compute_second_derivatives(r)
            t_i, _ = o.ct_build(i, var[i])
            f.ct_load(i, var[i], t_i)
            f.compute_collapse_times(i)
        p = f.products()
        f.set_tabulated_ct(var)                         # and the sweep building its own tables
        tv = f.sweep(radii)
        ps = f.products()
    ulp = np.spacing(np.maximum(np.abs(po["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.mean(np.abs(p["Fmax"].astype(np.float64) - po["Fmax"]) > 2 * ulp) < 1e-4
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    assert np.allclose(tv, tv_o, rtol=1e-12)
    _fmax_close(ps["Fmax"], po["Fmax"])
    assert np.mean(po["Fmax"] != p0["Fmax"]) > 0.3       # not the default flavour's numbers


def test_ell_sng_table_vs_oracle(api):
    """row f-4, ELL_SNG: per cell (a build without TABULATED_CT, src/collapse_times.c:416-426), then the collapse-time
    table filled by 250 000 adaptive RKF45 integrations on the device against the oracle's restatement
    (oracle/pf_sng.c), then the interpolating sweep.  The integrator's accept/reject decisions
    depend on pow() to the last bit, so a small share of the nodes may take a different step sequence: those agree
    to the integrator's own tolerance (1e-6 per step), the rest to rounding."""
    n = 32
    dk = synth.make_density(n, seed=43)
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([2.0, 0.0])
    cosmo = np.array([0.25, 0.75, 0.0, 0.0])
    d_in = np.array([1.28e-5, 1.28e-5])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y)
    var = o.compute_fmax(radii, do_lpt=False)
    o.set_collapse_model(1, cosmo, d_in)
    tv_c = o.compute_fmax(radii, do_lpt=False)          # ELL_SNG without TABULATED_CT: one integration per cell
    pc = o.products()
    o.set_tabulated_ct(var)
    tv_o = o.compute_fmax(radii, do_lpt=False)
    po = o.products()
    tab_o, _ = o.ct_build(1, var[1])
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y)
        f.set_collapse_model(1, cosmo, d_in)
        tv = f.sweep(radii)                             # k_collapse_sng on the six components
        p = f.products()
        assert np.allclose(tv, tv_c, rtol=1e-12)
        d = np.abs(p["Fmax"].astype(np.float64) - pc["Fmax"])
        ulp = np.spacing(np.maximum(np.abs(pc["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
        assert np.mean(d <= 2 * ulp) > 0.95 and np.mean(d > 1e-4 * np.maximum(1.0, pc["Fmax"])) < 1e-3, (np.mean(d <= 2 * ulp), d.max())
        assert np.mean(p["Rmax"] != pc["Rmax"]) < 5e-3 and (pc["Fmax"] >= 1.0).mean() > 0.05
        tab = f.ct_build(1, var[1])
        f.set_tabulated_ct(var)
        tv = f.sweep(radii)
        p = f.products()
    assert np.array_equal(tab == 0, tab_o == 0) or np.mean((tab == 0) != (tab_o == 0)) < 1e-4
    nz = (tab != 0) & (tab_o != 0)
    err = np.abs(tab[nz] - tab_o[nz]) / tab_o[nz]
    assert nz.mean() > 0.3 and np.mean(err > 1e-9) < 0.05 and err.max() < 1e-4, (np.mean(err > 1e-9), err.max())
    assert np.allclose(tv, tv_o, rtol=1e-12)
    d = np.abs(p["Fmax"].astype(np.float64) - po["Fmax"])
    assert np.mean(d > 1e-4 * np.maximum(1.0, po["Fmax"])) < 1e-3 and np.mean(p["Rmax"] != po["Rmax"]) < 5e-3
    assert (po["Fmax"] >= 1.0).mean() > 0.05
    # MOD_GRAV_FR on top: the table of one radius with the f(R) force modification switched on
    o.set_modified_gravity(1e-5, size=[2.0, 2.0])
    tab_fr_o, _ = o.ct_build(1, var[1])
    with api.Fmax(n) as f:
        f.set_invgrow(x, y)
        f.set_collapse_model(1, cosmo, d_in)
        f.set_modified_gravity(1e-5, size=[2.0, 2.0])
        tab_fr = f.ct_build(1, var[1])
    nz = (tab_fr != 0) & (tab_fr_o != 0)
    err = np.abs(tab_fr[nz] - tab_fr_o[nz]) / tab_fr_o[nz]
    assert np.mean(err > 1e-9) < 0.05 and err.max() < 1e-4
    both = nz & (tab_o != 0)
    assert np.mean(tab_fr_o[both] >= tab_o[both]) > 0.99 and np.mean(tab_fr_o[both] > tab_o[both] * (1 + 1e-4)) > 0.2   # earlier collapse


@pytest.mark.parametrize("case", ["zero", "dc_only", "single_mode", "one_radius", "huge_amplitude", "tiny_amplitude", "underflow_amplitude"])
def test_edge_case_fields_vs_oracle(api, case, monkeypatch):
    """degenerate inputs the solver's branch ladder exists for (src/collapse_times.c:114-221, 679-776): an empty field
    (q == 0, |l1| < 1e-20), a field that is only its k = 0 mode (untouched by the filter, src/fmax-pfft.c:368: all six
    components equal, eigenvalues (3h, 0, 0)), a single plane wave (two zero eigenvalues) -- both exactly on
    q^3 == r^2, where the -10 sentinel, a NaN from acos(1 + eps) or a value comes out depending on the last bit --,
    one smoothing radius, amplitudes far outside the inverse-growth table (linear extrapolation of my_spline_eval,
    src/cosmo.c:2016-2027), and an amplitude whose squares underflow: q == 0 with a tensor that is not isotropic, the
    reference's "already diagonal" branch (src/collapse_times.c:722-727), which the invariant z-pass cannot serve and
    answers by repeating the sweep with six components"""
    n = 16
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([2.0, 1.0, 0.0])
    dk = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    if case == "dc_only":
        dk[0, 0, 0] = 0.9 * n ** 3
    elif case == "single_mode":
        dk[1, 0, 0] = 0.4 * n ** 3
        dk[n - 1, 0, 0] = 0.4 * n ** 3            # Hermitian partner in the kz = 0 plane
    elif case == "one_radius":
        dk = synth.make_density(n, seed=2)
        radii = np.array([0.0])
    elif case == "huge_amplitude":
        dk = synth.make_density(n, seed=2) * 1e6
    elif case == "tiny_amplitude":
        dk = synth.make_density(n, seed=2) * 1e-12
    elif case == "underflow_amplitude":
        dk = synth.make_density(n, seed=2) * 1e-165
    o = oracle_lib.Oracle(n, 1)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=True)
    po = o.products()
    degenerate = case in ("dc_only", "single_mode")
    for flavour in (("fast", "exact") if degenerate else ("fast",)):
        monkeypatch.setenv("PF_EXACT_LIBM", "1" if flavour == "exact" else "0")
        with api.Fmax(n) as f:
            f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
            tv = f.compute_fmax(radii, do_lpt=True)
            p = f.products()
            pdf = f.Fmax_PDF()
            reruns = f.L.pf_debug_invariant_reruns(f.h)
        assert reruns == (1 if case == "underflow_amplitude" else 0), (case, reruns)
        if case == "underflow_amplitude":
            assert np.array_equal(p["Fmax"], po["Fmax"]) and np.array_equal(p["Rmax"], po["Rmax"])
        assert np.allclose(tv, tv_o, rtol=1e-12, atol=1e-300)
        assert int(pdf.sum()) == n ** 3
        fo, fg = po["Fmax"].astype(np.float64), p["Fmax"].astype(np.float64)
        ulp = np.spacing(np.maximum(np.abs(fo), 1.0).astype(np.float32)).astype(np.float64)
        close = np.abs(fg - fo) <= 2 * ulp * np.maximum(1.0, np.abs(fo) * 1e-6)
        if case == "dc_only" and flavour == "exact":
            # the Hessian is exact on both sides (a constant), the operations are the reference's own: same value
            assert np.array_equal(p["Fmax"], po["Fmax"]) and np.array_equal(p["Rmax"], po["Rmax"])
        elif case == "dc_only":
            # the value (to the sqrt(eps) sensitivity of acos at its end point), or the sentinel side of the edge
            assert np.all((np.abs(fg - fo) <= 1e-5 * np.abs(fo)) | (fg == -10.0)), (fg.ravel()[:4], fo.ravel()[:4])
        elif degenerate:
            # every cell of a plane wave sits on the edge at every radius, with a Hessian that carries the transform's
            # rounding: which of {sentinel, NaN-skipped update, value} wins per radius is not reproducible between two
            # correct implementations; what must hold is that nothing else comes out
            assert np.all((fg == -10.0) | ((fg >= 0.0) & np.isfinite(fg))) and np.all((fo == -10.0) | ((fo >= 0.0) & np.isfinite(fo)))
            assert np.max(fg) <= 1.5 * max(np.max(fo), 1.0) + 1.0
        else:
            assert np.mean(~close) < 2e-3, (fg.ravel()[:4], fo.ravel()[:4])
            assert np.mean(p["Rmax"] != po["Rmax"]) < 2e-3
        for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            amp = np.max(np.abs(po[name]))
            assert np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * amp + 1e-300, name
        if case == "zero":
            assert not p["Vel"].any() and np.all(p["Rmax"] == po["Rmax"]) and np.array_equal(p["Fmax"], po["Fmax"])


@pytest.mark.parametrize("n,general,path", [(24, "0", 1), (40, "0", 1), (48, "0", 1), (24, "1", 2), (40, "1", 2), (20, "0", 2), (36, "0", 2),
                                            (14, "0", 2), (22, "0", 2), (26, "0", 2), (28, "0", 2), (44, "0", 2)])
def test_general_grid_sizes_vs_oracle(api, n, general, path, monkeypatch):
    """grid sizes that are not a power of two (the reference takes any GridSize; 200^3 in INSTALLATION:101) against the oracle
    (plain O(n^2) transforms at these sizes), full path + taps: the hand-written passes with run-time stage plans where they
    apply (n = 8 m with m = 2^a 3^b 5^c: path 1); for the rest -- multiples of 2 and 4 only, prime factors 7, 11, 13 -- and under
    PF_GENERAL=1 the chirp-z transforms of csrc/pf_gfft.hip on the power-of-two stages (path 2).  No library transform anywhere."""
    import np_restatement as npr
    monkeypatch.setenv("PF_GENERAL", general)
    dk = synth.make_density(n, seed=n)
    dk[0, 0, 0] = 0.11 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([2.0, 1.0, 0.0])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=True)
    po = o.products()
    hes_o = o.second_derivatives(1.3)
    with api.Fmax(n) as f:
        assert f.L.pf_transform_path(f.h) == path
        f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        p = f.products()
        pdf = f.Fmax_PDF()
        f.compute_second_derivatives(1.3)
        hes = [f.second_derivative(i) for i in range(6)]
        rng = np.random.default_rng(1)
        real = rng.standard_normal((n, n, n))
        spec = f.forward_transform(real)
        back = f.reverse_transform(dk)
        d30 = f.compute_derivative(dk, 3, 0, 0.7, 2)
        assert np.array_equal(f.density(), dk)
    assert np.allclose(tv, tv_o, rtol=1e-12) and int(pdf.sum()) == n ** 3
    for a, b in zip(hes, hes_o):
        assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
    _fmax_close(p["Fmax"], po["Fmax"])
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * np.max(np.abs(po[name])), name
    want = np.fft.rfftn(real, axes=(0, 1, 2))
    assert np.max(np.abs(spec - want)) < 1e-12 * np.max(np.abs(want))
    want = np.fft.irfftn(dk, s=(n, n, n), axes=(0, 1, 2))
    assert np.max(np.abs(back - want)) < 1e-12 * np.max(np.abs(want))
    want = npr.derivative(dk, 0.7, 3, 0, g[1])
    assert np.max(np.abs(d30 - want)) < 1e-12 * np.max(np.abs(want))


def test_general_path_equals_fused_path_on_a_power_of_two(api, monkeypatch):
    """PF_GENERAL=1 forces the chirp-z transform path (one 3-D transform per component) on a size the shared passes also cover: same products"""
    n = 64
    dk = synth.make_density(n, seed=6)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([4.0, 1.0, 0.0])
    out = []
    for general in ("0", "1"):
        monkeypatch.setenv("PF_GENERAL", general)
        with api.Fmax(n) as f:
            f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
            tv = f.compute_fmax(radii, do_lpt=True)
            out.append((tv, f.products()))
    (tv0, p0), (tv1, p1) = out
    assert np.allclose(tv0, tv1, rtol=1e-12)
    _fmax_close(p1["Fmax"], p0["Fmax"])
    assert np.mean(p1["Rmax"] != p0["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p1[name].astype(np.float64) - p0[name])) <= 4e-7 * np.max(np.abs(p0[name])), name


@pytest.mark.parametrize("general", ["0", "1"])
def test_reference_example_size_200(api, general, monkeypatch):
    """BASELINE config 1 (200^3, INSTALLATION:101-102): size-independent properties and the device IC generator at that size,
    through the shared passes (200 = 8.5.5: run-time stage plan) and through the chirp-z transform path; the two
    against each other in test_grid_200_mixed_passes_vs_chirp_z_transforms_and_oracle"""
    n = 200
    monkeypatch.setenv("PF_GENERAL", general)
    x, y = synth.invgrow_table("lcdm")
    with api.Fmax(n) as f:
        assert f.L.pf_transform_path(f.h) == (2 if general == "1" else 1)
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
        tv = f.compute_fmax(np.array([8.0, 2.0, 0.0]), do_lpt=True)
        assert abs(np.sqrt(tv[-1]) - 2.5) < 1e-9                     # sigma(R=0) = the k-space normalisation
        assert tv[0] < tv[1] < tv[2]
        pdf = f.Fmax_PDF()
        assert int(pdf.sum()) == n ** 3
        f.compute_second_derivatives(0.0)
        h = [f.second_derivative(i) for i in range(3)]
        back = f.reverse_transform(f.density())
        assert np.max(np.abs(h[0] + h[1] + h[2] - back)) < 1e-10 * np.max(np.abs(back))   # Laplacian identity
        p = f.products()
        assert (p["Fmax"] >= 1.0).mean() > 0.2 and np.isfinite(p["Vel"]).all() and p["Vel_2LPT"].any()


def test_grid_200_mixed_passes_vs_chirp_z_transforms_and_oracle(api, monkeypatch):
    """200^3 (the reference's example size): the shared-pass path on the mixed-radix kernels against the chirp-z transform path
    (products of a three-radius sweep with displacements), its Hessian at one radius against the oracle, and the whole path --
    TrueVariance, Fmax, Rmax, the four displacement fields -- against the oracle"""
    n = 200
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([6.0, 1.5, 0.0])
    dk = synth.philox_density(n, synth.SEED, 2.5, -2.0)
    out = []
    for general in ("0", "1"):
        monkeypatch.setenv("PF_GENERAL", general)
        with api.Fmax(n) as f:
            f.set_density(dk); f.set_invgrow(x, y); f.set_growth(g)
            tv = f.compute_fmax(radii, do_lpt=True)
            p = f.products()
            f.compute_second_derivatives(1.5)
            out.append((tv, p, [f.second_derivative(i) for i in range(6)]))
    (tv0, p0, h0), (tv1, p1, h1) = out
    assert np.allclose(tv0, tv1, rtol=1e-12)
    for a, b in zip(h0, h1):
        assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
    _fmax_close(p0["Fmax"], p1["Fmax"], max_abs=None)
    assert np.mean(p0["Rmax"] != p1["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p0[name].astype(np.float64) - p1[name])) <= 4e-7 * np.max(np.abs(p1[name])), name
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    for a, b in zip(h0, o.second_derivatives(1.5)):
        assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
    # ... and the whole path against the oracle at this size: the sweep over the three radii and the 2LPT / 3LPT displacements
    # (the mixed-radix kernels with the plan of 200 points built in, the invariant z-pass and the fused 3LPT(b) contraction included)
    o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=True)
    po = o.products()
    assert np.allclose(tv0, tv_o, rtol=1e-12)
    _fmax_close(p0["Fmax"], po["Fmax"])
    assert np.mean(p0["Rmax"] != po["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p0[name].astype(np.float64) - po[name])) <= 4e-7 * np.max(np.abs(po[name])), name


def test_contexts_release_their_memory(api):
    """create / run / destroy in a loop, both transform paths, with every lazily allocated piece in use: the device
    memory in use returns to where it started (no leak across the reference's finalize_fft / compute_fft_plans cycles)"""
    x, y = synth.invgrow_table("lcdm")
    hip = C.CDLL("libamdhip64.so")            # the runtime the library itself is linked to (already loaded)
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        fr, tot = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
        return fr.value

    def cycle(n):
        with api.Fmax(n) as f:
            f.synth_density(synth.SEED, 2.5, -2.0)
            f.set_invgrow(x, y)
            f.set_tabulated_ct([1.0, 2.0])
            f.compute_fmax(np.array([1.0, 0.0]), do_lpt=True)
            f.select_sorted(1.0)
            f.products()

    cycle(64); cycle(48)
    free0 = free_bytes()
    for _ in range(10):
        cycle(64); cycle(48)
    free1 = free_bytes()
    assert abs(free1 - free0) < 64 * 2 ** 20, (free0, free1)


@pytest.mark.parametrize("n,ns,lpt", [(256, 12, False), (512, 3, True)])
def test_baseline_config_sizes_vs_oracle(api, n, ns, lpt):
    """BASELINE configs 2 and 3 as they are stated: 256^3 with all twelve smoothing radii, Fmax only; 512^3 with Fmax and
    the 2LPT / 3LPT displacements -- cell by cell against the oracle run on the host cores of the GPU box.  (Not at 1024^3:
    the oracle plus both product arrays need several hundred GB of host memory there; the metric's own size is covered by
    the size-independent properties.)"""
    radii = synth.radii_ladder(12) * (n / 1024.0) if ns == 12 else synth.radii_ladder(12)[[0, 4, 8, 10, 11]][-ns:] * (n / 1024.0)
    radii[-1] = 0.0
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=lpt)
        p = f.products()
        dk = f.density()
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii, do_lpt=lpt)
    po = o.products()
    assert np.allclose(tv, tv_o, rtol=1e-12)
    # with 10^7 - 10^8 cells a few sit orders of magnitude closer to the singular surface of the reference's cubic
    # (den -> 0) than anything at 128^3: there F moves by O(0.1 - 1) under a 1e-15 change of the Hessian.  The count
    # criterion stays; the far outliers must each be explained by the oracle's own solver on the GPU's Hessian.
    outliers = _fmax_close(p["Fmax"], po["Fmax"], max_abs=None)
    d = np.abs(p["Fmax"].astype(np.float64) - po["Fmax"])
    far = [tuple(c) for c in outliers if d[tuple(c)] > 2e-3]
    assert len(far) <= max(8, int(3e-7 * d.size)), len(far)   # (observed: 1 at 256^3, ~1e-7 of the cells at 512^3 and 1024^3)
    if far:
        o8 = oracle_lib.Oracle(8, 1)
        o8.set_invgrow(x, y)
        rng = np.random.default_rng(1)
        with api.Fmax(n) as f:
            f.set_density(dk)
            for ir in sorted(set(int(p["Rmax"][c]) for c in far)):
                f.compute_second_derivatives(radii[ir])
                hg = [f.second_derivative(i) for i in range(6)]
                for c in far:
                    if p["Rmax"][c] != ir:
                        continue
                    h = np.array([hh[c] for hh in hg])
                    fo = o8.inverse_collapse_time(h)[0]
                    spread = max(abs(o8.inverse_collapse_time(h * (1.0 + rng.uniform(-4.4e-16, 4.4e-16, 6)))[0] - fo) for _ in range(64))
                    ulp = float(np.spacing(np.float32(max(abs(fo), 1.0))))
                    assert abs(float(p["Fmax"][c]) - fo) <= max(8.0 * spread, 2.0 * ulp), (c, fo, p["Fmax"][c], spread)
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        if lpt:
            assert np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * np.max(np.abs(po[name])), name
        else:
            assert not p[name].any() and not po[name].any()


def test_fast_flavour_elementary_functions_on_the_device(api):
    """the solver's default arithmetic, function by function, on the device against correctly rounded values (numpy
    longdouble / mpmath): hardware-seeded division and square root, the cosine triple of the trigonometric root formula, series
    log10, exp and 10^, x^0.333333333333333 and x/9"""
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(17)
    ld = np.longdouble

    def run(f, which, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b if b is not None else np.ones_like(a), dtype=np.float64)
        out = np.empty_like(a)
        dp = C.POINTER(C.c_double)
        f._chk(f.L.pf_debug_math(f.h, which, a.ctypes.data_as(dp), b.ctypes.data_as(dp), len(a), out.ctypes.data_as(dp)))
        return out

    def ulps(got, want):
        want = np.asarray(want, dtype=np.float64)
        return np.abs(got - want) / np.spacing(np.maximum(np.abs(want), 1e-300))

    n = 200000
    with api.Fmax(64) as f:
        a = rng.standard_normal(n) * 10.0 ** rng.integers(-100, 100, n)
        b = rng.standard_normal(n) * 10.0 ** rng.integers(-100, 100, n)
        assert ulps(run(f, 0, a, b), (a.astype(ld) / b.astype(ld))).max() <= 1.0
        x = np.abs(a)
        assert ulps(run(f, 1, x), np.sqrt(x.astype(ld))).max() <= 1.0
        assert run(f, 1, np.array([0.0]))[0] == 0.0 and np.isnan(run(f, 1, np.array([-1.0]))[0])
        # the hardware seeds behind them: one refinement step is enough only while they are good to better than 2^-22
        assert np.max(np.abs(run(f, 9, x).astype(ld) * x.astype(ld) - 1)) < 2.0 ** -22
        assert np.max(np.abs(run(f, 10, x).astype(ld) ** 2 * x.astype(ld) - 1)) < 2.0 ** -21
        # the cosine triple cos((acos x + 2 pi k)/3) without acos and sincos: largest root within 2 ulp; the other two within
        # 2.5 eps plus what a 1-ulp change of x moves them (x -> +-1 is a degenerate pair of roots)
        xa = np.concatenate([rng.uniform(-1, 1, 30000), 1.0 - 10.0 ** rng.uniform(-16, 0, 5000), -1.0 + 10.0 ** rng.uniform(-16, 0, 5000), [1.0, -1.0, 0.5, -0.5, 0.0]])
        xa = np.clip(xa, -1.0, 1.0)
        tt = [mp.acos(mp.mpf(float(v))) for v in xa]
        t64 = np.arccos(xa)
        eps = np.finfo(float).eps
        for k in range(3):
            want = np.array([float(mp.cos((t + 2 * mp.pi * k) / 3)) for t in tt])
            got = run(f, 2, xa, np.full(len(xa), float(k)))
            if k == 0:
                assert ulps(got, want).max() <= 2.0
            with np.errstate(divide="ignore", invalid="ignore"):
                cond = np.abs(np.sin((t64 + 2 * np.pi * k) / 3)) / (3 * np.maximum(np.sin(t64), 1e-300)) * np.spacing(np.abs(xa))
            assert (np.abs(got - want) <= 2.5 * eps + 2.0 * np.where(np.isfinite(cond), cond, 0.0)).all(), k
        assert np.all(np.isnan(run(f, 2, np.array([1.0000001, -1.5]), np.zeros(2))))
        xl = np.concatenate([10.0 ** rng.uniform(-300, 300, 50000), 10.0 ** rng.uniform(-5, 2, 100000), 1.0 + rng.uniform(-1e-3, 1e-3, 20000)])
        assert ulps(run(f, 3, xl), np.log(xl.astype(ld)) / np.log(ld(10))).max() <= 2.0
        xp = 10.0 ** rng.uniform(-30, 30, 50000)
        assert ulps(run(f, 5, xp), xp.astype(ld) ** ld(0.333333333333333)).max() <= 2.0
        assert np.array_equal(run(f, 6, a), a / 9.0)
        xe = np.concatenate([rng.uniform(-40, 0.5, 20000), rng.uniform(-3.5, 3.5, 20000), rng.uniform(-300, 300, 5000), [0.0, 1.0, -1.0]])
        want = np.array([float(mp.exp(mp.mpf(float(v)))) for v in xe])
        want10 = np.array([float(mp.power(10, mp.mpf(float(v)))) for v in xe])
        ok, ok10 = (want > 1e-300) & np.isfinite(want), (want10 > 1e-300) & np.isfinite(want10)
        assert ulps(run(f, 7, xe)[ok], want[ok]).max() <= 1.0
        assert ulps(run(f, 8, xe)[ok10], want10[ok10]).max() <= 1.0
        edge = np.array([-746.0, -1e10, -np.inf, 710.0, np.inf])
        assert np.array_equal(run(f, 7, edge), [0.0, 0.0, 0.0, np.inf, np.inf])


@pytest.mark.parametrize("seed", [2024, 7, 99])
def test_state_machine_of_one_context_against_fresh_contexts(api, seed):
    """One long-lived context driven through a seeded random sequence of the entry points -- new density, growth factors, LPT order,
    sweep + displacements apart or together (sources formed by the last solve), second derivatives at another radius in between,
    re-entrant displacements -- must at every step hold exactly what a fresh context computes from scratch: the flags that say
    which fields are current (Hessian, real-space sources, resident source spectra) may not go stale."""
    n = 32
    rng = np.random.default_rng(seed)
    x, y = synth.invgrow_table("lcdm")
    state = {"seed": 1, "g": synth.growth_multipliers(), "order": 3, "radii": np.array([2.0, 1.0, 0.0])}

    def fresh(with_lpt=True, g=None):
        with api.Fmax(n) as f:
            f.set_density(synth.make_density(n, seed=state["seed"])); f.set_invgrow(x, y); f.set_growth(state["g"] if g is None else g)
            f.set_lpt_order(state["order"])
            tv = f.sweep(state["radii"])
            if with_lpt:
                f.compute_displacements(1, 0)
            return tv, f.products()

    def same(p, q, names):
        for name in names:
            assert np.array_equal(p[name], q[name]), name

    vel = ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2")
    with api.Fmax(n) as f:
        f.set_invgrow(x, y)
        f.set_density(synth.make_density(n, seed=state["seed"])); f.set_growth(state["g"])
        have_spectra = False
        for step in range(40):
            op = rng.integers(0, 7)
            if op == 0:                                            # new density
                state["seed"] += 1
                f.set_density(synth.make_density(n, seed=state["seed"]))
                have_spectra = False
            elif op == 1:                                          # other growth factors
                state["g"] = synth.growth_multipliers() * rng.uniform(0.5, 1.5, 4)
                f.set_growth(state["g"])
            elif op == 2:                                          # other LPT order
                state["order"] = int(rng.integers(1, 4))
                f.set_lpt_order(state["order"])
                have_spectra = False                               # (a lower order's spectra do not serve a higher one)
            elif op == 3:                                          # compute_fmax: sources formed by the last solve
                tv = f.compute_fmax(state["radii"], do_lpt=True)
                tv0, p0 = fresh()
                assert np.array_equal(tv, tv0)
                same(f.products(), p0, ("Fmax", "Rmax") + vel)
                have_spectra = True
            elif op == 4:                                          # sweep, something else in between, then the displacements
                tv = f.sweep(state["radii"])
                f.compute_second_derivatives(1.3)                  # overwrites the R = 0 Hessian
                f.compute_displacements(1, 1)                      # recompute_sd puts it back
                tv0, p0 = fresh()
                assert np.array_equal(tv, tv0)
                same(f.products(), p0, ("Fmax", "Rmax") + vel)
                have_spectra = True
            elif op == 5 and have_spectra:                         # re-entrant displacements at another redshift
                g2 = state["g"] * np.array([0.7, 0.5, 0.35, 0.35])
                f.set_growth(g2)
                f.compute_displacements(0, 0)
                _, p0 = fresh(g=g2)
                same(f.products(), p0, vel)
                f.set_growth(state["g"])
            elif op == 6:                                          # sweep with other radii, no displacements
                state["radii"] = np.array([float(rng.uniform(1.5, 4.0)), float(rng.uniform(0.5, 1.4)), 0.0])
                tv = f.sweep(state["radii"])
                tv0, p0 = fresh(with_lpt=False)
                assert np.array_equal(tv, tv0)
                same(f.products(), p0, ("Fmax", "Rmax"))
