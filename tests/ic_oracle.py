"""Initial-condition side of the CPU oracle (TEST INFRASTRUCTURE): seed plane along
the spiral (src/GenIC.c:840-990), E&H power spectrum normalised to sigma8
(src/cosmo.c:1058-1075, 1559-1585), GenIC_large through oracle/pf_genic.c."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle_lib


class Cosmo(C.Structure):
    _fields_ = [("Omega0", C.c_double), ("OmegaBaryon", C.c_double), ("Hubble100", C.c_double), ("PrimordialIndex", C.c_double)]


def _lib():
    L = oracle_lib.lib()
    L.orc_ranlxd1_nth.restype = C.c_ulong
    L.orc_ranlxd1_nth.argtypes = [C.c_ulong, C.c_int]
    L.orc_powerspec_EH.restype = C.c_double
    L.orc_powerspec_EH.argtypes = [C.c_double, C.POINTER(Cosmo)]
    L.orc_genic.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_uint), C.c_double, C.POINTER(Cosmo), C.POINTER(C.c_double)]
    L.orc_genic_ic.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_uint), C.c_double, C.POINTER(Cosmo), C.c_int, C.c_int, C.POINTER(C.c_double)]
    return L


def get_map(x, y):
    """ordinal (1-based) of the point (x, y) along the square spiral, src/GenIC.c:840-855"""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    l = 2 * np.maximum(np.abs(x), np.abs(y))
    c = (y > x).astype(np.int64) + ((x > 0) & (x == y)).astype(np.int64)
    d = np.where(c != 0, l * 3 + x + y, l - x - y)
    return (l - 1) * (l - 1) + d


def seed_table(n: int, seed: int) -> np.ndarray:
    """seed[jj, ii]: the get_map-th draw of MT19937(seed) at the spiral coordinates of grid point (ii, jj)
    (generate_seeds_subregion, src/GenIC.c:875-990; plane -> spiral: coordinate >= n/2 -> coordinate - n, :1029-1041)"""
    g = np.arange(n)
    s = np.where(g >= n // 2, g - n, g)
    m = get_map(s[None, :], s[:, None])          # [jj][ii]
    assert len(np.unique(m)) == n * n and m.min() == 1
    mt = np.random.RandomState(seed).randint(0, 2 ** 32, size=int(m.max()), dtype=np.uint32)  # gsl_rng_mt19937 draws
    return np.ascontiguousarray(mt[m - 1].astype(np.uint32))


def pk_norm(p, sigma8: float) -> float:
    """normalize_PowerSpectrum (src/cosmo.c:1058-1075): sigma8^2 / top-hat mass variance at 8/h Mpc"""
    from scipy.integrate import quad
    L = _lib()
    cos = Cosmo(p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"])
    R = 8.0 / p["Hubble100"]

    def integrand(logk):
        k = np.exp(logk)
        kr = k * R
        w = 1.0 if kr < 1e-5 else 3.0 * (np.sin(kr) / kr ** 3 - np.cos(kr) / kr ** 2)
        return L.orc_powerspec_EH(k, C.byref(cos)) * w * w * k ** 3 / (2.0 * np.pi ** 2)

    val, _ = quad(integrand, -10.0, np.log(500.0 / R), epsabs=0, epsrel=1e-10, limit=1000)
    return sigma8 ** 2 / val


SPECTRUM_CODES = {"EH": 1, "Efstathiou": 3, "PowerLaw": 4}   # WhichSpectrum (src/cosmo.c:1009-1046)


def genic(n: int, box_true_mpc: float, seed: int, pknorm: float, p, fixed: bool = False, paired: bool = False, pk_table=None,
          spectrum: str = "EH", wdm_mass_kev: float = 0.0) -> np.ndarray:
    """fixed / paired: params.FixedIC / params.PairedIC (src/GenIC.c:370-376); pk_table = (log10 k [1/Mpc], log10(k^3 P)): the knots of
    SPLINE[SP_PK] for a tabulated spectrum (PowerSpec_Tabulated, src/cosmo.c:1432-1435) instead of Eisenstein & Hu; spectrum
    "Efstathiou" / "PowerLaw" and wdm_mass_kev: the other forms of PowerSpectrum() and its warm-dark-matter cut-off (:953-1007)"""
    L = _lib()
    cos = Cosmo(p["Omega0"], p["OmegaBaryon"], p["Hubble100"], p["PrimordialIndex"])
    st = seed_table(n, seed)
    out = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    dp = C.POINTER(C.c_double)
    if pk_table is None:
        npk, lk, lp = 0, None, None
    else:
        lk = np.ascontiguousarray(pk_table[0], dtype=np.float64)
        lp = np.ascontiguousarray(pk_table[1], dtype=np.float64)
        npk = len(lk)
    L.orc_genic_form.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_uint), C.c_double, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp,
                                 C.c_int, C.c_int, dp]
    which = 2 if npk else SPECTRUM_CODES[spectrum]
    rc = L.orc_genic_form(n, box_true_mpc, st.ctypes.data_as(C.POINTER(C.c_uint)), pknorm, C.cast(C.byref(cos), C.c_void_p), which,
                          wdm_mass_kev, 3.085678e24, npk, lk.ctypes.data_as(dp) if npk else None, lp.ctypes.data_as(dp) if npk else None,
                          int(fixed), int(paired), out.view(np.float64).ctypes.data_as(dp))
    assert rc == 0
    return out


def growth_table_lcdm(omega0: float):
    """SPLINE[SP_INVGROW] knots (src/cosmo.c:101,229,298-401): log10 D vs log10 a, D(a=1) = 1, flat LCDM without
    radiation (growing mode = H int da/(aH)^3; the reference integrates the same ODE with rkf45 to 1e-8)"""
    from pinocchio_amd import synth
    return synth.invgrow_table("lcdm", omega0)
