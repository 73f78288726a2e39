"""Synthetic inputs of the hot path: density spectrum,
smoothing ladder, inverse-growth spline knots, growth multipliers.

Host/numpy side, used for the small parity cases (the same arrays are handed
to the oracle and to the HIP library).  Large benchmark inputs are generated
directly in HBM by the library (pf_synth_density, csrc/pf_synth.hip); the
numpy mirror of that generator is `philox_density` below.
"""
from __future__ import annotations

import numpy as np

SEED = 486604  # the reference's RandomSeed (HMF_Validation/parameter_file)
NBINS = 210    # src/pinocchio.h:65


def radii_ladder(ns: int = 12) -> np.ndarray:
    """Smoothing radii in CELL units, last one 0 (src/initialization.c:424)."""
    full = np.array([16.0, 11.3, 8.0, 5.7, 4.0, 2.8, 2.0, 1.4, 1.0, 0.7, 0.35, 0.0])
    if ns == len(full):
        return full.copy()
    if ns < 1:
        raise ValueError("ns >= 1")
    if ns == 1:
        return np.array([0.0])
    r = 16.0 * 2.0 ** (-0.5 * np.arange(ns - 1))
    return np.concatenate([r, [0.0]])


def kgrid(n: int):
    """Signed wavenumbers in rad/cell exactly as src/fmax-pfft.c:306-339
    (index > n/2 -> index - n; Nyquist maps to +pi)."""
    idx = np.arange(n)
    s = np.where(idx > n // 2, idx - n, idx).astype(np.float64)
    k1 = (2.0 * np.pi / n) * s
    kz = (2.0 * np.pi / n) * np.arange(n // 2 + 1, dtype=np.float64)
    return k1, k1, kz


def shape_spectrum(n: int, white_k: np.ndarray, sigma0: float = 2.5, slope: float = -2.0) -> np.ndarray:
    """white-noise half-spectrum -> delta(k) with P(k) ~ k^slope inside the
    Nyquist sphere, DC and Nyquist planes zero (src/GenIC.c:193-281), scaled so
    that sigma(R=0) = sigma0; pre-multiplied by N^3 like kdensity (GenIC.c:430)."""
    kx, ky, kz = kgrid(n)
    k2 = kx[:, None, None] ** 2 + ky[None, :, None] ** 2 + kz[None, None, :] ** 2
    with np.errstate(divide="ignore"):
        amp = np.where(k2 > 0, k2 ** (slope / 4.0), 0.0)
    amp[k2 >= np.pi ** 2] = 0.0
    h = n // 2
    amp[h, :, :] = 0.0
    amp[:, h, :] = 0.0
    amp[:, :, h] = 0.0
    dk = white_k * amp
    real = np.fft.irfftn(dk, s=(n, n, n), axes=(0, 1, 2))
    sig = np.sqrt(np.mean(real ** 2))
    dk *= sigma0 / sig
    return np.ascontiguousarray(dk)


def make_density(n: int, seed: int = SEED, sigma0: float = 2.5, slope: float = -2.0) -> np.ndarray:
    """complex128 [n][n][n/2+1] half-spectrum, numpy RNG (small parity cases)."""
    rng = np.random.default_rng(seed)
    white = rng.standard_normal((n, n, n))
    wk = np.fft.rfftn(white, axes=(0, 1, 2))
    return shape_spectrum(n, wk, sigma0, slope)


def invgrow_table(kind: str = "lcdm", omega0: float = 0.25):
    """210 knots of SPLINE[SP_INVGROW] (src/cosmo.c:101,229,401): x = log10 D(a),
    y = log10 a on log10 a = -4 + 0.02 i.  'eds': D = a.  'lcdm': flat LCDM
    growing mode D(a) = 2.5 Om H(a) int_0^a da'/(a' H(a'))^3, D(1) = 1."""
    loga = -4.0 + 0.02 * np.arange(NBINS)
    a = 10.0 ** loga
    if kind == "eds":
        return loga.copy(), loga.copy()
    if kind != "lcdm":
        raise ValueError(kind)
    ol = 1.0 - omega0

    def hub(x):
        return np.sqrt(omega0 / x ** 3 + ol)

    def growth(x):
        # Gauss-Legendre on t in [0,1], a' = x t^2 (removes the a'^{1/2} endpoint singularity)
        t, w = np.polynomial.legendre.leggauss(200)
        t = 0.5 * (t + 1.0)
        w = 0.5 * w
        ap = x * t ** 2
        integrand = 1.0 / (ap * hub(ap)) ** 3 * (2.0 * x * t)
        return 2.5 * omega0 * hub(x) * np.sum(w * integrand)

    d = np.array([growth(x) for x in a])
    d /= growth(1.0)
    return np.log10(d), loga.copy()


def growth_multipliers() -> np.ndarray:
    """GrowingMode, GrowingMode_2LPT, GrowingMode_3LPT_1 (with its minus sign,
    src/cosmo.c:1810), GrowingMode_3LPT_2 at z=0, EdS values with the
    reference's normalisations (src/cosmo.c:250-257)."""
    return np.array([1.0, 3.0 / 7.0, -1.0 / 9.0, 5.0 / 42.0])


# --- numpy mirror of the device generator (csrc/pf_synth.hip) ---------------

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al. 2011) on uint32 numpy arrays."""
    c0 = c0.astype(np.uint32); c1 = c1.astype(np.uint32)
    c2 = c2.astype(np.uint32); c3 = c3.astype(np.uint32)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    for _ in range(10):
        p0 = _M0 * c0.astype(np.uint64)
        p1 = _M1 * c2.astype(np.uint64)
        hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = (p0 & _MASK32).astype(np.uint32)
        hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = (p1 & _MASK32).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
        k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_white(n: int, seed: int = SEED, x0: int = 0, nx: int | None = None) -> np.ndarray:
    """Real-space unit Gaussian white noise for the x-slab [x0, x0+nx), one
    Philox call per PAIR of cells (global pair index as the counter), Box-Muller.
    Decomposition independent: any rank can generate its own slab."""
    nx = n if nx is None else nx
    rows = np.arange(x0 * n, (x0 + nx) * n, dtype=np.uint64).reshape(nx, n)
    gidx = rows[:, :, None] * np.uint64(n // 2) + np.arange(n // 2, dtype=np.uint64)[None, None, :]
    c0 = (gidx & _MASK32).astype(np.uint32)
    c1 = (gidx >> np.uint64(32)).astype(np.uint32)
    z = np.zeros_like(c0)
    r0, r1, r2, r3 = philox4x32(c0, c1, z, z, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = ((r0.astype(np.uint64) << np.uint64(21)) ^ (r1.astype(np.uint64) >> np.uint64(11))).astype(np.float64)
    u1 = (u1 + 0.5) * (1.0 / 9007199254740992.0)  # 2^-53, in (0,1)
    u2 = ((r2.astype(np.uint64) << np.uint64(21)) ^ (r3.astype(np.uint64) >> np.uint64(11))).astype(np.float64)
    u2 = (u2 + 0.5) * (1.0 / 9007199254740992.0)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    out = np.empty((nx, n, n), dtype=np.float64)
    out[:, :, 0::2] = rad * np.cos(ang)
    out[:, :, 1::2] = rad * np.sin(ang)
    return out


def philox_density(n: int, seed: int = SEED, sigma0: float = 2.5, slope: float = -2.0) -> np.ndarray:
    """numpy mirror of pf_synth_density (csrc/pf_synth.hip): Philox white noise in
    real space -> r2c -> P(k) shaping -> sigma(R=0) = sigma0."""
    wk = np.fft.rfftn(philox_white(n, seed), axes=(0, 1, 2))
    return shape_spectrum(n, wk, sigma0, slope)
