"""Second end-to-end pin on reference-held data: the 256^3 runs the reference commits under tests/only_HMF_tests/
(RECOMPUTE_DISPLACEMENTS_LCDM and SCALE_DEP_LCDM, V5.0: Omega0 = 0.3, no baryons in the E&H fit, fixed-amplitude initial
conditions, nine radii; identical logs and Fmax PDFs, the second from a -DSCALE_DEPENDENT build).  Data:
tests/golden/hmf256_kat.json (made by tests/golden/make_hmf256_kat.py).

  * CPU: the oracle (IC generator with params.FixedIC + hot path) reproduces the logged sigma of all nine radii to the four printed
    decimals, the collapsed-cell count and the 210-bin histogram -- with the one shared inverse-growth spline, and, the way the
    SCALE_DEPENDENT build runs, with one spline per radius made from the run's own growth table (scaledep.out);
  * GPU: the HIP path on the same density, and entirely on the device (GenIC with FixedIC, per-radius splines).
"""
import json
import os

import numpy as np
import pytest

import ic_oracle
import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLD, "hmf256_kat.json")) as f:
        return json.load(f)


def _box(p):
    return p["BoxSize_h100"] / p["Hubble100"]   # true Mpc


def _table_spline(kat):
    """SPLINE_INVGROW knots from the run's own growth table: x = log10 D(a), y = log10 a (src/initialization.c:1704-1708;
    in LCDM the smoothed-variance growth of every radius is the linear growing mode)"""
    t = np.array(kat["scaledep_a_D1"])
    return np.log10(t[:, 1]), np.log10(t[:, 0])


def _check(kat, tv, pdf, l1_max, count_max):
    sig = np.sqrt(tv)
    want = np.array(kat["computed_sigma"])
    assert np.all(np.abs(sig - want) <= 6e-5), (sig, want)          # logged with four decimals
    pdf = np.asarray(pdf).astype(np.int64)
    ref = np.array(kat["FmaxPDF"], dtype=np.int64)
    coll = int(pdf[10:].sum())
    l1 = int(np.abs(pdf - ref).sum())
    print("collapsed", coll, "reference", kat["collapsed"], "PDF L1", l1)
    assert abs(coll - kat["collapsed"]) <= count_max, (coll, kat["collapsed"])
    assert l1 <= l1_max, l1
    return coll, l1


def test_pk_normalisation_without_baryons(kat):
    got = ic_oracle.pk_norm(kat["params"], kat["params"]["Sigma8"])
    assert got == pytest.approx(kat["PkNorm"], rel=2e-5)            # 6.6972e+06 as logged (5 digits)


def test_growth_table_of_the_run(kat):
    x, y = ic_oracle.growth_table_lcdm(kat["params"]["Omega0"])
    t = np.array(kat["scaledep_a_D1"])
    sel = t[:, 0] <= 1.0
    mine = 10.0 ** np.interp(np.log10(t[sel, 0]), y, x)
    assert np.max(np.abs(mine / t[sel, 1] - 1.0)) < 2e-5            # the file has 6 significant digits


@pytest.fixture(scope="module")
def density(kat):
    p = kat["params"]
    return ic_oracle.genic(p["GridSize"], _box(p), p["RandomSeed"], kat["PkNorm"], p, fixed=bool(p["FixedIC"]))


def test_oracle_reproduces_the_256_runs(kat, density):
    p = kat["params"]
    n = p["GridSize"]
    radii_cells = np.array(kat["radii_Mpc"]) / (_box(p) / n)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(density)
    o.set_invgrow(x, y)
    tv = o.compute_fmax(radii_cells, do_lpt=False)
    _check(kat, tv, o.fmax_pdf(), l1_max=400, count_max=8)          # measured: 10 989 577 vs 10 989 578, L1 = 118 of 16 777 216
    # the SCALE_DEPENDENT build: SPLINE_INVGROW[ismooth], here from the six-digit table the run wrote
    xt, yt = _table_spline(kat)
    for i in range(len(radii_cells)):
        o.set_invgrow_radius(i, xt, yt)
    tv2 = o.compute_fmax(radii_cells, do_lpt=False)
    assert np.array_equal(tv2, tv)
    _check(kat, tv2, o.fmax_pdf(), l1_max=4000, count_max=400)      # the table's 6 digits move F by ~1e-6: cells next to a bin edge


@pytest.mark.gpu
def test_hip_path_reproduces_the_256_runs(kat, density):
    from pinocchio_amd import api
    p = kat["params"]
    n = p["GridSize"]
    radii_cells = np.array(kat["radii_Mpc"]) / (_box(p) / n)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    xt, yt = _table_spline(kat)
    with api.Fmax(n) as f:
        f.set_density(density)
        f.set_invgrow(x, y)
        tv = f.sweep(radii_cells)
        _check(kat, tv, f.Fmax_PDF(), l1_max=400, count_max=8)
        # entirely on the device: GenIC with FixedIC from seed and cosmology, one inverse-growth spline per radius
        f.genic_density(p["RandomSeed"], _box(p), p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"],
                        pknorm=kat["PkNorm"], fixed=True)
        for i in range(len(radii_cells)):
            f.set_invgrow(xt, yt, ismooth=i)
        tv2 = f.sweep(radii_cells)
        assert np.allclose(tv2, tv, rtol=1e-11)
        _check(kat, tv2, f.Fmax_PDF(), l1_max=4000, count_max=400)


@pytest.mark.gpu
def test_paired_initial_conditions_flip_the_field(kat):
    """params.PairedIC (src/GenIC.c:371): every phase shifted by pi, i.e. delta -> -delta, on the device generator"""
    from pinocchio_amd import api
    p = kat["params"]
    n = 64
    args = (p["RandomSeed"], 64.0 / 0.7, p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"])
    with api.Fmax(n) as f:
        f.genic_density(*args, pknorm=1.0e7)
        a = f.density()
        f.genic_density(*args, pknorm=1.0e7, paired=True)
        b = f.density()
        f.genic_density(*args, pknorm=1.0e7, fixed=True)
        c = f.density()
    amp = np.max(np.abs(a))
    assert np.max(np.abs(a + b)) <= 1e-12 * amp
    # fixed amplitudes: |delta(k)| depends on |k| only; same phases as the Rayleigh-sampled field
    sel = np.abs(a) > 0
    assert np.max(np.abs(np.angle(a[sel] * np.conj(c[sel])))) < 1e-9
    from pinocchio_amd import synth
    kx, ky, kz = synth.kgrid(n)
    k2 = np.round((kx[:, None, None] ** 2 + ky[None, :, None] ** 2 + kz[None, None, :] ** 2) * (n / (2 * np.pi)) ** 2).astype(int)
    for q in (1, 2, 3, 9, 50):
        v = np.abs(c[(k2 == q) & sel])
        assert len(v) and np.ptp(v) <= 1e-12 * np.max(v), q


# ---------------------------------------------------------------------------------------------------------------------
# TABULATED_CT + ELL_SNG + MOD_GRAV_FR: the reference's committed f(R) run, tests/golden/mg256_kat.json
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mg():
    with open(os.path.join(GOLD, "mg256_kat.json")) as f:
        return json.load(f)


def _mg_setup(mg):
    """what the reference hands the model (src/collapse_times.c:312-380, 295-312): GrowingMode at a = 1e-5 for every radius
    (below the table: linear extrapolation in log-log from its first two rows; every k-bin is the same there), the size the force
    modification screens with (the radius; the last one takes the one before), H0 / c, the density parameters"""
    # (the slope comes from the file's own d ln D / d ln a column, 1 at the first row: the six printed digits of two neighbouring
    #  rows would give it to 1e-5 only, which a decade of extrapolation turns into 1e-4 of D_in)
    a0, d0 = mg["growth_first_rows_a_D1"][0]
    d_in = d0 * (1e-5 / a0) ** mg["dlnD_dlna_first_row"]
    radii = np.array(mg["radii_Mpc"])
    size = radii.copy()
    size[-1] = size[-2]
    p = mg["params"]
    return np.full(len(radii), d_in), size, (p["Omega0"], p["OmegaLambda"], 0.0, 0.0), 100.0 / 299792.458


def test_oracle_reproduces_the_f_of_R_run_with_tabulated_sng_collapse(mg):
    """The whole table machinery against reference-held output: per radius 250 000 adaptive RKF45 integrations of the nine-equation
    ellipsoid system with the f(R) force modification, the node splines, the bilinear-of-splines lookup per cell, the running
    maximum.  Measured: sigma of all nine radii to the printed decimals, 10 935 586 collapsed cells against the reference's
    10 935 578, histogram L1 = 198 of 16 777 216.  (About two minutes on eight cores: it is the table integrations.)"""
    p = mg["params"]
    n = p["GridSize"]
    dk = ic_oracle.genic(n, _box(p), p["RandomSeed"], mg["PkNorm"], p, fixed=True)
    d_in, size, cosmo, h_over_c = _mg_setup(mg)
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    o.set_invgrow(x, y)                                     # not used by ELL_SNG (F = 1 / a_collapse), required by the interface
    o.set_collapse_model(1, cosmo=cosmo, d_in=d_in)
    o.set_modified_gravity(p["FR0"], h_over_c, size=size)
    o.set_tabulated_ct(np.array(mg["variance"]))
    tv = o.compute_fmax(np.array(mg["radii_Mpc"]) / (_box(p) / n), do_lpt=False)
    _check(mg, tv, o.fmax_pdf(), l1_max=800, count_max=40)


@pytest.mark.gpu
def test_hip_path_reproduces_the_f_of_R_run(mg):
    """the same on the device: pf_set_collapse_model / pf_set_modified_gravity / pf_set_tabulated_ct, tables built per radius by
    k_ct_table_sng (9 ms each), k_collapse_tab; density from the device's own GenIC with FixedIC"""
    from pinocchio_amd import api
    p = mg["params"]
    n = p["GridSize"]
    d_in, size, cosmo, h_over_c = _mg_setup(mg)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    with api.Fmax(n) as f:
        f.genic_density(p["RandomSeed"], _box(p), p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"],
                        pknorm=mg["PkNorm"], fixed=True)
        f.set_invgrow(x, y)
        f.set_collapse_model(1, cosmo=cosmo, d_in=d_in)
        f.set_modified_gravity(p["FR0"], h_over_c, size=size)
        f.set_tabulated_ct(np.array(mg["variance"]))
        tv = f.sweep(np.array(mg["radii_Mpc"]) / (_box(p) / n))
        _check(mg, tv, f.Fmax_PDF(), l1_max=800, count_max=40)


# ----------------------------------------------------------------------------------------------------------------------------
# READ_PK_TABLE + SCALE_DEPENDENT: the reference's committed run with a tabulated (CAMB) spectrum, tests/golden/readpk256_kat.json
# (tests/golden/make_readpk256_kat.py).  The z = 0 table the run read gives SPLINE[SP_PK]; its growth does not depend on scale
# (all ten k bins of scaledep.out agree), so SPLINE_INVGROW[ismooth] is one table for every radius.
@pytest.fixture(scope="module")
def rkat():
    with open(os.path.join(GOLD, "readpk256_kat.json")) as f:
        return json.load(f)


def _pk_table(rkat):
    """knots of SPLINE[SP_PK] (src/cosmo.c:1302-1309): log10 of k in true 1/Mpc, log10(k^3 P) with k in h/Mpc and P in (Mpc/h)^3"""
    t = np.array(rkat["camb_z0_k_hMpc_P"])
    return np.log10(t[:, 0] * rkat["params"]["Hubble100"]), np.log10(t[:, 0] ** 3 * t[:, 1])


@pytest.fixture(scope="module")
def rdensity(rkat):
    p = rkat["params"]
    return ic_oracle.genic(p["GridSize"], _box(p), p["RandomSeed"], 1.0, p, fixed=True, pk_table=_pk_table(rkat))


def test_sigma8_of_the_camb_table(rkat):
    """normalize_PowerSpectrum with Sigma8 = 0 (src/cosmo.c:1074-1079): the run logs the sigma8 of the table it was given"""
    from scipy.integrate import quad
    from scipy.interpolate import CubicSpline
    lk, lp = _pk_table(rkat)
    sp = CubicSpline(lk, lp, bc_type="natural")
    R = 8.0 / rkat["params"]["Hubble100"]

    def f(lnk):
        k = np.exp(lnk)
        x = np.log10(k)
        y = sp(x) if lk[0] <= x <= lk[-1] else (lp[0] + (x - lk[0]) * (lp[1] - lp[0]) / (lk[1] - lk[0]) if x < lk[0]
                                                else lp[-1] + (x - lk[-1]) * (lp[-1] - lp[-2]) / (lk[-1] - lk[-2]))
        kr = k * R
        w = 3.0 * (np.sin(kr) / kr ** 3 - np.cos(kr) / kr ** 2)
        return 10.0 ** y * w * w / (2.0 * np.pi ** 2)          # P k^3 / (2 pi^2) per ln k
    var = quad(f, np.log(1e-5), np.log(500.0 / R), limit=2000)[0]
    assert np.sqrt(var) == pytest.approx(rkat["Sigma8_of_the_table"], rel=2e-4)     # 1.057903 as logged


def test_oracle_reproduces_the_read_pk_table_run(rkat, rdensity):
    p = rkat["params"]
    n = p["GridSize"]
    radii_cells = np.array(rkat["radii_Mpc"]) / (_box(p) / n)
    xt, yt = _table_spline(rkat)
    o = oracle_lib.Oracle(n, 0)
    o.set_density(rdensity)
    for i in range(len(radii_cells)):
        o.set_invgrow_radius(i, xt, yt)
    tv = o.compute_fmax(radii_cells, do_lpt=False)
    _check(rkat, tv, o.fmax_pdf(), l1_max=2000, count_max=100)      # measured: 12 822 323 vs 12 822 323, L1 = 444 of 16 777 216


@pytest.mark.gpu
def test_hip_path_reproduces_the_read_pk_table_run(rkat, rdensity):
    from pinocchio_amd import api
    p = rkat["params"]
    n = p["GridSize"]
    radii_cells = np.array(rkat["radii_Mpc"]) / (_box(p) / n)
    xt, yt = _table_spline(rkat)
    with api.Fmax(n) as f:
        f.set_density(rdensity)
        for i in range(len(radii_cells)):
            f.set_invgrow(xt, yt, ismooth=i)
        tv = f.sweep(radii_cells)
        _check(rkat, tv, f.Fmax_PDF(), l1_max=2000, count_max=100)
        # entirely on the device: GenIC with the tabulated spectrum (PkNorm 1: the table is trusted), fixed amplitudes
        f.genic_density(p["RandomSeed"], _box(p), p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"],
                        pknorm=1.0, fixed=True, pk_table=_pk_table(rkat))
        dk = f.density()
        assert np.max(np.abs(dk - rdensity)) <= 1e-11 * np.max(np.abs(rdensity))
        tv2 = f.sweep(radii_cells)
        assert np.allclose(tv2, tv, rtol=1e-11)
        _check(rkat, tv2, f.Fmax_PDF(), l1_max=2000, count_max=100)


# ----------------------------------------------------------------------------------------------------------------------------
# The reference's example run (example/log: V5.1, default flags, 128^3 on four tasks, Eisenstein & Hu, Rayleigh-sampled amplitudes):
# sigma of the seven radii and the collapsed-cell count; tests/golden/example_kat.json (make_example_kat.py).  Its log holds no
# histogram (the FmaxPDF file beside it is from another run).
@pytest.fixture(scope="module")
def ekat():
    with open(os.path.join(GOLD, "example_kat.json")) as f:
        return json.load(f)


def _check_example(ekat, tv, pdf):
    sig = np.sqrt(tv)
    assert np.all(np.abs(sig - np.array(ekat["computed_sigma"])) <= 6e-5), sig
    coll = int(np.asarray(pdf).astype(np.int64)[10:].sum())
    print("collapsed", coll, "reference", ekat["collapsed"])
    assert abs(coll - ekat["collapsed"]) <= 8, (coll, ekat["collapsed"])          # measured: 687 252 vs 687 249 of 2 097 152


def test_oracle_reproduces_the_example_log(ekat):
    p = ekat["params"]
    n = p["GridSize"]
    assert ic_oracle.pk_norm(p, p["Sigma8"]) == pytest.approx(ekat["PkNorm"], rel=2e-5)
    dk = ic_oracle.genic(n, _box(p), p["RandomSeed"], ekat["PkNorm"], p)
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y)
    tv = o.compute_fmax(np.array(ekat["radii_Mpc"]) / (_box(p) / n), do_lpt=False)
    _check_example(ekat, tv, o.fmax_pdf())


@pytest.mark.gpu
def test_hip_path_reproduces_the_example_log_on_the_device(ekat):
    from pinocchio_amd import api
    p = ekat["params"]
    n = p["GridSize"]
    x, y = ic_oracle.growth_table_lcdm(p["Omega0"])
    with api.Fmax(n) as f:
        f.genic_density(p["RandomSeed"], _box(p), p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"], pknorm=ekat["PkNorm"])
        f.set_invgrow(x, y)
        tv = f.sweep(np.array(ekat["radii_Mpc"]) / (_box(p) / n))
        _check_example(ekat, tv, f.Fmax_PDF())
