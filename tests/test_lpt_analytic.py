"""The displacement half of the path against closed-form known answers (tests/golden/lpt_analytic.json: plane-wave
density fields, answers derived in exact arithmetic from the reference's formulas by tests/golden/make_lpt_analytic.py
-- src/LPT.c:64-93, 112-137, 181-228; src/fmax-pfft.c:366-384, 444-456 -- and cross-checked there by symbolic
differentiation).  The reference tree holds no `Vel*` output, so this is the pin of rows A11-A13 of SURVEY.md section 8:

  * CPU: oracle/pf_oracle.c is held to the vectors (Hessians and source spectra to 1e-13, the fp32 products to the
    correctly rounded value);
  * GPU (-m gpu): the HIP path is held to the same vectors, through the C ABI.
"""
import numpy as np
import pytest

import lpt_analytic as la
import oracle_lib
from pinocchio_amd import synth

CASES = ["three_waves", "two_waves_with_mean", "one_wave", "axis_waves"]


def test_float64_evaluation_matches_the_40_digit_samples():
    for name in CASES:
        c = la.case(name)
        e = la.expected(c, 16)
        assert la.check_sample(c, e) < 4e-15, name


def _check_fields(name, n, d, kv, prod, hess_tol=2e-14, spec_tol=2e-13, fp32_exact_frac=1e-3, x0=0, nx=None):
    c = la.case(name)
    e = la.expected(c, n, x0, nx)
    for i in range(6):
        assert np.max(np.abs(d[i] - e["d"][i])) <= hess_tol * max(1.0, np.max(np.abs(e["d"][i]))), (name, n, "d", i)
    if kv is not None:
        for w, key in enumerate(("s2", "s3a", "s3b")):
            want = la.spectrum_of(c[key], n)[x0:x0 + (n if nx is None else nx)]
            # (sources that vanish identically leave the rounding residue of products of the Hessian's amplitude)
            amp = max(np.max(np.abs(want)), float(n) ** 3 * max(np.max(np.abs(x)) for x in e["d"]) ** 3)
            assert np.max(np.abs(kv[w] - want)) <= spec_tol * amp, (name, n, key, np.max(np.abs(kv[w] - want)) / amp)
    for k in la.VEL_NAMES:
        amp = np.max(np.abs(e[k]))
        if amp == 0.0:  # sources that vanish identically (rank-one tensor): only rounding residue, far below fp32 resolution of Vel
            assert np.max(np.abs(prod[k])) <= 1e-12 * n, (name, n, k)
            continue
        # absolute slack: fp64 rounding of the transforms relative to the field's amplitude (cells where the field crosses zero)
        ulps, frac = la.fp32_close(prod[k], e[k], extra_abs=4e-15 * amp * np.log2(n))
        assert ulps <= 1.0, (name, n, k, ulps)
        assert frac <= fp32_exact_frac, (name, n, k, frac)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("n", [16, 32])
def test_oracle_vs_closed_form(name, n):
    c = la.case(name)
    o = oracle_lib.Oracle(n, 2)
    o.set_density(la.density_spectrum(c, n))
    x, y = synth.invgrow_table("eds")
    o.set_invgrow(x, y)
    o.set_growth(la.growth(c))
    d = o.second_derivatives(0.0)  # before the displacements: like the reference's globals, the oracle keeps ScaleDep.order of its last call
    o.compute_fmax(np.array([0.0]), do_lpt=True)
    kv = [o.kvector(w) for w in range(3)]
    _check_fields(name, n, d, kv, o.products())


def _check_scale_dependent(name, n, prod, x0=0, nx=None):
    c = la.case(name)
    e = la.scale_dependent_expected(c, n, x0, nx)
    for k in la.VEL_NAMES:
        amp = np.max(np.abs(e[k]))
        # the multiplier goes through log10 and pow(10., .) in the reference's own arithmetic: ~1e-15 relative per mode
        ulps, frac = la.fp32_close(prod[k], e[k], extra_abs=2e-14 * amp)
        assert ulps <= 1.0, (name, n, k, ulps)
        assert frac <= 1e-3, (name, n, k, frac)


@pytest.mark.parametrize("name", ["three_waves", "axis_waves"])
@pytest.mark.parametrize("n", [16, 32])
def test_oracle_scale_dependent_growth_vs_closed_form(name, n):
    """row f-3: k-dependent growth multipliers per mode (SCALE_DEPENDENT build), modes below kmin, inside the table and above kmax"""
    c = la.case(name)
    sd = la.scale_dependent_tables(c)
    o = oracle_lib.Oracle(n, 2)
    o.set_density(la.density_spectrum(c, n))
    x, y = synth.invgrow_table("eds")
    o.set_invgrow(x, y)
    for order in range(4):
        o.set_growth_table(order + 1, sd["T"][order], sd["logkmin"], sd["dlogk"], float(sd["sign"][order]))
    o.compute_fmax(np.array([0.0]), do_lpt=True)
    _check_scale_dependent(name, n, o.products())


# ---------------------------------------------------------------------------------------------------- GPU ----
@pytest.fixture(scope="module")
def api():
    from pinocchio_amd import api as a
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("n,fb", [(16, 8), (32, 8), (64, 8), (128, 8)])
def test_hip_path_vs_closed_form(api, name, n, fb):
    c = la.case(name)
    x, y = synth.invgrow_table("eds")
    with api.Fmax(n, field_bytes=fb) as f:
        f.set_density(la.density_spectrum(c, n))
        f.set_invgrow(x, y)
        f.set_growth(la.growth(c))
        f.compute_fmax(np.array([0.0]), do_lpt=True)
        d = [f.second_derivative(i) for i in range(6)]
        kv = [f.kvector(w) for w in range(3)]
        p = f.products()
    _check_fields(name, n, d, kv, p)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["three_waves", "two_waves_with_mean"])
def test_hip_path_vs_closed_form_unfused_and_six_component_kernels(api, name, monkeypatch):
    """the same answers from the kernels the default sweep does not take: 3LPT(b) accumulated by k_lpt_accum"""
    monkeypatch.setenv("PF_LPT_FUSE", "0")
    monkeypatch.setenv("PF_INVARIANTS", "0")
    c = la.case(name)
    n = 32
    x, y = synth.invgrow_table("eds")
    with api.Fmax(n) as f:
        f.set_density(la.density_spectrum(c, n))
        f.set_invgrow(x, y)
        f.set_growth(la.growth(c))
        f.compute_fmax(np.array([1.0, 0.0]), do_lpt=True)
        d = [f.second_derivative(i) for i in range(6)]
        kv = [f.kvector(w) for w in range(3)]
        p = f.products()
    _check_fields(name, n, d, kv, p)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["three_waves", "two_waves_with_mean"])
def test_hip_fp32_field_path_vs_closed_form(api, name):
    """BASELINE config 5 arithmetic (fp32 fields): the closed form to fp32 accuracy of the whole chain"""
    c = la.case(name)
    n = 64
    e = la.expected(c, n)
    x, y = synth.invgrow_table("eds")
    with api.Fmax(n, field_bytes=4) as f:
        f.set_density(la.density_spectrum(c, n))
        f.set_invgrow(x, y)
        f.set_growth(la.growth(c))
        f.compute_fmax(np.array([0.0]), do_lpt=True)
        d = [f.second_derivative(i) for i in range(6)]
        p = f.products()
    for i in range(6):
        assert np.max(np.abs(d[i] - e["d"][i])) <= 3e-6 * max(1.0, np.max(np.abs(e["d"][i]))), (name, i)
    for k in la.VEL_NAMES:
        amp = np.max(np.abs(e[k]))
        assert np.max(np.abs(p[k] - e[k])) <= 2e-5 * amp, (name, k, np.max(np.abs(p[k] - e[k])) / amp)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_hip_slabs_vs_closed_form(api, nranks):
    """the slab-decomposed path (virtual ranks on one GPU, in-process fabric) against the same closed form"""
    import test_gpu_multirank as mr
    c = la.case("three_waves")
    n = 32
    x, y = synth.invgrow_table("eds")
    dk = la.density_spectrum(c, n)
    nxl = n // nranks

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y)
        f.set_growth(la.growth(c))
        f.compute_fmax(np.array([0.0]), do_lpt=True)
        return [f.second_derivative(i) for i in range(6)], [f.kvector(w) for w in range(3)], f.products()

    res = mr.run_ranks(api, n, nranks, body)
    for r in range(nranks):
        _check_fields("three_waves", n, res[r][0], res[r][1], res[r][2], x0=r * nxl, nx=nxl)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [256, 512])
def test_lpt_identities_at_scale(api, n):
    """BASELINE sizes, no oracle: on a random field the library's sources must be the reference's cell-by-cell formulas of
    its own Hessian (src/LPT.c:64-93, 112-137; numpy, independent transforms), and every displacement field must be the
    irrotational field with div Psi = -g (S - <S>) (src/fmax-pfft.c:366-384 with the growth of src/LPT.c:181-228)"""
    x, y = synth.invgrow_table("lcdm")
    g = np.array([0.9, 0.31, -0.07, 0.052])
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        f.set_growth(g)
        f.sweep(np.array([0.0]))
        d = [f.second_derivative(i) for i in range(6)]
        f.compute_displacements(1, 0)
        kv = [f.kvector(w) for w in range(3)]
        dk = f.density()
        p = f.products()
        # phi2_ab through the library's own single-component transform at the larger size (checked against numpy elsewhere)
        phi2_lib = [f.compute_derivative(kv[0], a, b) for a, b in ((1, 1), (2, 2), (3, 3), (1, 2), (1, 3), (2, 3))] if n > 256 else None
    irfft = lambda s: np.fft.irfftn(s, s=(n, n, n), axes=(0, 1, 2))
    s2 = d[0] * d[1] + d[0] * d[2] + d[1] * d[2] - d[3] ** 2 - d[4] ** 2 - d[5] ** 2
    rms = lambda a: float(np.sqrt(np.mean(np.abs(a) ** 2)))
    assert rms(irfft(kv[0]) - s2) <= 1e-12 * rms(s2)
    s3a = 3.0 * (d[0] * (d[1] * d[2] - d[5] * d[5]) - d[3] * (d[3] * d[2] - d[4] * d[5]) + d[4] * (d[3] * d[5] - d[4] * d[1]))
    assert rms(irfft(kv[1]) - s3a) <= 1e-12 * rms(s3a)
    del s3a
    kx, ky, kz = synth.kgrid(n)
    kk = (kx[:, None, None], ky[None, :, None], kz[None, None, :])
    k2 = kk[0] ** 2 + kk[1] ** 2 + kk[2] ** 2
    k2[0, 0, 0] = 1.0
    s3b = 2.0 * (d[0] + d[1] + d[2]) * s2
    for i, (a, b) in enumerate(((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))):
        if phi2_lib is None:
            m = kk[a] * kk[b] / k2 * np.ones_like(k2)
            m[0, 0, 0] = 1.0  # the k = 0 mode is left untouched (src/fmax-pfft.c:368)
            phi2 = irfft(kv[0] * m)
        else:
            phi2 = phi2_lib[i]
        s3b -= 2.0 * (1.0 if i < 3 else 2.0) * phi2 * d[i]
    assert rms(irfft(kv[2]) - s3b) <= 1e-11 * rms(s3b)
    del s3b, s2, d, phi2_lib
    # displacements: fp32 columns; Nyquist planes excluded (k = +pi has no -pi partner, the synthetic field has none)
    h = n // 2
    ok = np.ones((n, n, h + 1), dtype=bool)
    ok[h, :, :] = False; ok[:, h, :] = False; ok[:, :, h] = False; ok[0, 0, 0] = False
    for name, spec, go in (("Vel", dk, g[0]), ("Vel_2LPT", kv[0], g[1]), ("Vel_3LPT_1", kv[1], g[2]), ("Vel_3LPT_2", kv[2], g[3])):
        v = [np.fft.rfftn(p[name][..., a].astype(np.float64), axes=(0, 1, 2)) for a in range(3)]
        div = 1j * (kk[0] * v[0] + kk[1] * v[1] + kk[2] * v[2])
        want = -go * spec
        assert rms((div - want)[ok]) <= 2e-6 * rms(want[ok]), name          # fp32 storage of the columns
        amp = rms(v[0][ok]) + rms(v[1][ok]) + rms(v[2][ok])
        for a, b in ((0, 1), (0, 2), (1, 2)):
            assert rms((kk[a] * v[b] - kk[b] * v[a])[ok]) <= 2e-6 * amp, (name, "curl", a, b)
        assert abs(v[0][0, 0, 0]) <= 1e-6 * amp * n ** 1.5                  # the mean displacement is zero


@pytest.mark.gpu
def test_full_bench_workload_on_the_box_of_the_metric(api):
    """1024^3 fp64, compute_fmax(radii_ladder(12), do_lpt=True): the very step bench.py times, checked without the oracle
    (which would need several hundred GB).  (1) TrueVariance of the last radius = sigma^2 of the synthetic field, the
    variances grow down the ladder, the PDF counts every cell.  (2) The 2LPT and 3LPT(a) sources are the reference's
    cell-by-cell formulas (src/LPT.c:64-93) of the library's own R = 0 Hessian, on sampled x-planes, with the source spectra
    transformed back by an independent FFT (scipy / pocketfft).  (3) Every one of the four displacement fields (twelve
    columns, read back one block at a time through pf_get_block) is the irrotational field with i k.Psi = -g S and zero
    mean (src/fmax-pfft.c:366-384 with the growths of src/LPT.c:181-228), to fp32 storage.  (4) Fmax and Rmax after all
    twelve radii, cell by cell against the oracle on five sampled x-planes (5.2 million cells), and (5) the 3LPT(b) source
    on those planes -- both through the sampled-plane form of the oracle (oracle_lib.PlaneOracle), which restates the path
    for the chosen planes only."""
    import os
    import scipy.fft as sfft
    n = 1024
    workers = oracle_lib.usable_cores()   # (not os.cpu_count(): the control group grants 16 of the host's 256 hardware threads)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = synth.radii_ladder(12)
    rms = lambda a: float(np.sqrt(np.mean(np.abs(a) ** 2)))
    kx, ky, kz = synth.kgrid(n)
    h = n // 2
    planes = [0, 1, 317, 512, 1023]
    with api.Fmax(n) as f:
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        assert np.sqrt(tv[-1]) == pytest.approx(2.5, rel=1e-10) and np.all(np.diff(tv) > 0)
        assert int(f.Fmax_PDF().sum()) == n ** 3
        # (2) sources against the Hessian that is still in place (the last radius of the ladder is R = 0)
        d = [f.second_derivative(i)[planes] for i in range(6)]
        kv = []
        for w in range(3):
            spec = f.kvector(w)
            kv.append(spec)
            if w < 2:
                real = sfft.irfftn(spec, s=(n, n, n), axes=(0, 1, 2), workers=workers)[planes]
                if w == 0:
                    want = d[0] * d[1] + d[0] * d[2] + d[1] * d[2] - d[3] ** 2 - d[4] ** 2 - d[5] ** 2
                else:
                    want = 3.0 * (d[0] * (d[1] * d[2] - d[5] * d[5]) - d[3] * (d[3] * d[2] - d[4] * d[5]) + d[4] * (d[3] * d[5] - d[4] * d[1]))
                assert rms(real - want) <= 1e-12 * rms(want), ("source", w)
                del real, want
        dk = f.density()
        # (4) Fmax and Rmax of the whole twelve-radius sweep, cell by cell against the oracle on the sampled planes
        # (orc_plane_*: the reference's filter per mode, the x-transform as the plain sum for these planes only, its
        # per-cell collapse pass with the running maximum -- the whole-box oracle needs ~450 GB of host memory here).
        # The usual contract: within 2 ulp(fp32) except on <= 2e-5 of the cells; the far ones must be cells whose cubic
        # is ill-conditioned in the oracle itself (it moves as much under 2-ulp noise on its own Hessian).
        po = oracle_lib.PlaneOracle(n, planes, 0)
        po.set_invgrow(x, y)
        st = po.stream(radii, po.HESSIAN)          # every radius in ONE pass over delta(k) (orc_plane_acc_*: 12 x 6 x 5 planes of accumulators, 3 GB)
        st.add(dk, 0)
        hess = []
        for ismooth in range(len(radii)):
            hess.append(st.finish(ismooth))
            po.collapse_times(ismooth, hess[-1])
        st.close()
        hamp = max(float(np.max(np.abs(hh))) for hh in hess[-1])
        for i in range(6):   # the R = 0 Hessian still in place on the device
            assert np.max(np.abs(d[i] - hess[-1][i])) <= 1e-12 * hamp, ("hessian", i)
        shape = (len(planes), n, n)
        wf, wr = po.fmax.reshape(shape), po.rmax.reshape(shape)
        gf = f.block("FMAX").reshape(n, n, n)[planes]
        gr = f.block("RMAX").view(np.int32).reshape(n, n, n)[planes]
        ulp = np.spacing(np.maximum(np.abs(wf), 1.0).astype(np.float32)).astype(np.float64)
        df = np.abs(gf.astype(np.float64) - wf.astype(np.float64))
        bad = np.argwhere(df > 2 * ulp)
        assert len(bad) <= max(1, int(2e-5 * df.size)), (len(bad), float(df.max()))
        assert np.mean(df > 0) < 1e-3 and np.mean(gr != wr) < 1e-3
        far = [tuple(c) for c in bad if df[tuple(c)] > 2e-3]
        assert len(far) <= 16, len(far)
        rng = np.random.default_rng(1)
        for c in far:
            ir = int(gr[c])
            hc = np.array([hess[ir][i][c] for i in range(6)])
            o8 = oracle_lib.Oracle(8, 1)
            o8.set_invgrow(x, y)
            fo = o8.inverse_collapse_time(hc)[0]
            spread = max(abs(o8.inverse_collapse_time(hc * (1.0 + rng.uniform(-4.4e-16, 4.4e-16, 6)))[0] - fo) for _ in range(64))
            assert abs(float(gf[c]) - fo) <= max(8.0 * spread, 2.0 * float(ulp[c])), (c, fo, float(gf[c]), spread)
        del hess, gf, gr
        # (5) the 3LPT(b) source (src/LPT.c:89-91, 134-137) on the same planes: 2 (d11 + d22 + d33) S2 minus the contraction of
        # the first-order Hessian with the Hessian of the 2LPT potential, the latter from the 2LPT source spectrum by the
        # plane oracle; the library's spectrum of it transformed back on the planes by the plane oracle as well
        s2 = d[0] * d[1] + d[0] * d[2] + d[1] * d[2] - d[3] ** 2 - d[4] ** 2 - d[5] ** 2
        s3b = 2.0 * (d[0] + d[1] + d[2]) * s2
        phi2 = po.derivatives(kv[0], 0.0, po.HESSIAN)
        for i in (0, 3, 4, 1, 5, 2):   # the reference's order 11,12,13,22,23,33
            s3b -= 2.0 * (1.0 if i < 3 else 2.0) * phi2[i] * d[i]
        real = po.derivatives(kv[2], 0.0, [(-1, -1)])[0]
        assert rms(real - s3b) <= 1e-11 * rms(s3b), ("source", 2, rms(real - s3b) / rms(s3b))
        del s2, s3b, phi2, real
        po.close()
        del d
        # (3) displacements, one block (three fp32 columns, 12.9 GB) at a time
        ok = np.ones((n, n, h + 1), dtype=bool)      # Nyquist planes excluded (k = +pi has no -pi partner), and k = 0
        ok[h, :, :] = False; ok[:, h, :] = False; ok[:, :, h] = False; ok[0, 0, 0] = False
        kk = (kx[:, None, None], ky[None, :, None], kz[None, None, :])
        for blk, spec, go in (("ZEL ", dk, g[0]), ("2LPT", kv[0], g[1]), ("31PT", kv[1], g[2]), ("32PT", kv[2], g[3])):
            vel = f.block(blk).reshape(n, n, n, 3)
            assert np.isfinite(vel).all()
            v = [sfft.rfftn(np.ascontiguousarray(vel[..., a], dtype=np.float64), axes=(0, 1, 2), workers=workers) for a in range(3)]
            del vel
            amp = rms(v[0][ok]) + rms(v[1][ok]) + rms(v[2][ok])
            assert abs(v[0][0, 0, 0]) <= 1e-6 * amp * n ** 1.5 and abs(v[1][0, 0, 0]) <= 1e-6 * amp * n ** 1.5   # zero mean
            div = 1j * (kk[0] * v[0])
            div += 1j * (kk[1] * v[1])
            div += 1j * (kk[2] * v[2])
            div += go * spec
            assert rms(div[ok]) <= 2e-6 * abs(go) * rms(spec[ok]), (blk, "div")                 # fp32 storage of the columns
            del div
            for a, b in ((0, 1), (0, 2), (1, 2)):
                curl = kk[a] * v[b]
                curl -= kk[b] * v[a]
                assert rms(curl[ok]) <= 2e-6 * amp, (blk, "curl", a, b)
                del curl
            del v


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["three_waves", "axis_waves"])
@pytest.mark.parametrize("n,nranks", [(16, 1), (32, 1), (64, 1), (32, 2)])
def test_hip_scale_dependent_growth_vs_closed_form(api, name, n, nranks):
    """row f-3 on the device: pf_set_growth_table + k_apply_growth against the per-mode closed form"""
    import test_gpu_multirank as mr
    c = la.case(name)
    sd = la.scale_dependent_tables(c)
    x, y = synth.invgrow_table("eds")
    dk = la.density_spectrum(c, n)
    nxl = n // nranks

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y)
        for order in range(4):
            f.set_growth_table(order + 1, sd["T"][order], sd["logkmin"], sd["dlogk"], float(sd["sign"][order]))
        f.compute_fmax(np.array([0.0]), do_lpt=True)
        return f.products()

    if nranks == 1:
        with api.Fmax(n) as f:
            _check_scale_dependent(name, n, body(f, 0))
    else:
        res = mr.run_ranks(api, n, nranks, body)
        for r in range(nranks):
            _check_scale_dependent(name, n, res[r], x0=r * nxl, nx=nxl)
