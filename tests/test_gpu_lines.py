"""Every instantiation of the hand-written 1-D passes, line by line against numpy's pocketfft (an independent transform):
the strided x/y pass (k_strided), the z-pass c2r (k_c2r / k_c2r_persistent), the forward z-pass (k_r2c) and the six-row
invariant z-pass (k_c2r_invariants), fp64 and fp32 fields, with and without band pruning and with the k-space factors of
compute_derivative (src/fmax-pfft.c:306-397).  N = 2048 is the row length of BASELINE config 5 (2048^3 on eight GPUs): its
box does not fit one GPU, its kernels do -- pf_debug_lines runs them on a batch of lines without a context."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [16, 64, 256, 1024, 2048]
# grid sizes that are not a power of two: the run-time stage plans of csrc/pf_mixed_kernels.hip (radices 8, 5, 4, 3, 2) --
# 24 = 8.3, 40 = 8.5, 96 = 8.4.3, 120 = 8.5.3, 200 = 8.5.5 (the reference's example size), 384 = 8.8.2.3, 768 = 8.8.4.3,
# 1000 = 8.5.5.5, 1536 = 8.8.8.3; their z-passes run on the half lengths 12 = 4.3, 20 = 4.5, 48, 60, 100 = 4.5.5, 192, ...
MIXED = [24, 40, 96, 120, 200, 384, 768, 1000, 1536, 400, 640, 800, 1280, 1600, 2000,   # (24, 40: run-time plans; from 96 on every plan is compiled in)
         144, 360, 720, 1080, 1440, 1800, 1944]                                         # (round 6: the sizes of the second and third translation unit)
ALL_SIZES = SIZES + MIXED


@pytest.fixture(scope="module")
def L():
    from pinocchio_amd import _lib
    return _lib.load()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def run(L, fb, n, pas, inp, out_shape, mul=0, band=1 << 30, nouter=1, ncols=1, pre=0, rs=0.0, growth=1.0, outer_offset=0):
    inp = np.ascontiguousarray(inp)
    out = np.zeros(out_shape, dtype=np.complex128 if pas in (0, 1, 3) else np.float64)
    rc = L.pf_debug_lines(fb, n, pas, mul, band, nouter, ncols, pre, rs, growth, outer_offset,
                          _dp(inp.view(np.float64)), _dp(out.view(np.float64)))
    assert rc == 0, L.pf_last_error()
    return out


def run_invariants(L, n, x, rows, fb=8):
    """pass 4: six spectra rows -> (mu1, mu2, mu3) rows (fp64 whatever the fields are), plus the q == 0 flag"""
    x = np.ascontiguousarray(x)
    out = np.zeros(3 * rows * n + 1)
    rc = L.pf_debug_lines(fb, n, 4, 0, 1 << 30, rows, 1, 0, 0.0, 1.0, 0, _dp(x.view(np.float64)), _dp(out))
    assert rc == 0, L.pf_last_error()
    return out[:-1].reshape(3, rows, n), out[-1]


def signed(n):
    e = np.arange(n)
    return np.where(e > n // 2, e - n, e).astype(np.float64)  # Nyquist stays +n/2 (src/fmax-pfft.c:306-339)


def kfactor(mul, k):
    return {0: np.ones_like(k) + 0j, 1: k + 0j, 2: k * k + 0j, 3: 1j * k}[mul]


def tol(fb, n):
    return (2e-15 if fb == 8 else 1e-6) * np.log2(n)


@pytest.mark.parametrize("n", ALL_SIZES)
@pytest.mark.parametrize("fb", [8, 4])
def test_strided_pass_lines(L, n, fb):
    rng = np.random.default_rng(n + fb)
    nouter, ncols = 3, 21  # not a multiple of the tile width: the last tile is ragged
    x = rng.standard_normal((nouter, n, ncols)) + 1j * rng.standard_normal((nouter, n, ncols))
    k = 2 * np.pi / n * signed(n)
    for mul in (0, 1, 2, 3):
        got = run(L, fb, n, 0, x, x.shape, mul=mul, nouter=nouter, ncols=ncols)
        want = np.fft.ifft(x * kfactor(mul, k)[None, :, None], axis=1) * n
        assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, mul)
    got = run(L, fb, n, 1, x, x.shape, nouter=nouter, ncols=ncols)
    want = np.fft.fft(x, axis=1)
    assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, "forward")
    # band-limited loads: wavenumbers beyond the band are exact zeros of the input
    band = n // 5
    xb = x * (np.abs(signed(n)) <= band)[None, :, None]
    got = run(L, fb, n, 0, x, x.shape, mul=2, band=band, nouter=nouter, ncols=ncols)
    want = np.fft.ifft(xb * kfactor(2, k)[None, :, None], axis=1) * n
    assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, "band")


@pytest.mark.parametrize("n", ALL_SIZES)
@pytest.mark.parametrize("fb", [8, 4])
def test_first_pass_filter_on_lines(L, n, fb):
    """the x-pass of the sweep: window, 1/k^2, growth and the factor of its own axis fused into the load"""
    rng = np.random.default_rng(3 * n + fb)
    nouter, ncols, off = 4, 9, n - 2  # outer rows n-2, n-1, 0, 1 (mod n) around k_outer = 0: the k = 0 mode sits in row 2
    x = rng.standard_normal((nouter, n, ncols)) + 1j * rng.standard_normal((nouter, n, ncols))
    ke = 2 * np.pi / n * signed(n)
    so = (np.arange(nouter) + off)
    so = np.where(so > n // 2, so - n, so).astype(np.float64)
    ko = 2 * np.pi / n * so
    kc = 2 * np.pi / n * np.arange(ncols)
    k2 = ko[:, None, None] ** 2 + ke[None, :, None] ** 2 + kc[None, None, :] ** 2
    for rs, g, mul in ((0.0, 1.0, 0), (2.5, 0.37, 2), (0.0, -0.111, 3)):
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.where(k2 > 0, np.exp(-0.5 * k2 * rs * rs) * g / k2, 0.0)
        got = run(L, fb, n, 0, x, x.shape, mul=mul, nouter=nouter, ncols=ncols, pre=1, rs=rs, growth=g, outer_offset=off)
        want = np.fft.ifft(x * f * kfactor(mul, ke)[None, :, None], axis=1) * n
        assert np.max(np.abs(got - want)) <= 4 * tol(fb, n) * np.max(np.abs(want)), (n, fb, rs, mul)


@pytest.mark.parametrize("n", ALL_SIZES)
@pytest.mark.parametrize("fb", [8, 4])
@pytest.mark.parametrize("persist", ["24", "0"])
def test_zpass_c2r_lines(L, n, fb, persist, monkeypatch):
    monkeypatch.setenv("PF_ZPASS_PERSIST", persist)  # 0: the one-shot kernel k_c2r (switches are read at every call of the tap)
    rng = np.random.default_rng(5 * n + fb)
    rows, h = 37, n // 2 + 1
    x = rng.standard_normal((rows, h)) + 1j * rng.standard_normal((rows, h))
    kz = 2 * np.pi / n * np.arange(h)
    for mul in (0, 1, 2, 3):
        got = run(L, fb, n, 2, x, (rows, n), mul=mul, nouter=rows)
        want = np.fft.irfft(x * kfactor(mul, kz)[None, :], n=n, axis=1) * n
        assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, mul)
    band = n // 7
    got = run(L, fb, n, 2, x, (rows, n), mul=1, band=band, nouter=rows)
    want = np.fft.irfft(x * (np.arange(h) <= band) * kz, n=n, axis=1) * n
    assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, "band")


@pytest.mark.parametrize("n", ALL_SIZES)
@pytest.mark.parametrize("fb", [8, 4])
def test_zpass_r2c_lines(L, n, fb):
    rng = np.random.default_rng(7 * n + fb)
    rows = 29
    x = rng.standard_normal((rows, n))
    got = run(L, fb, n, 3, x, (rows, n // 2 + 1), nouter=rows)
    want = np.fft.rfft(x, axis=1)
    assert np.max(np.abs(got - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb)


@pytest.mark.parametrize("n", [16, 64, 256, 512, 1024, 2048, 24, 200, 768, 1000, 2000, 96, 720, 1440, 1920])      # (512, 1024, 2048: the forms with waves that only reduce; 24 ...: k_mixed_c2r_invariants, from 96 on with reducing waves too)
def test_invariant_zpass_lines_fp32_fields(L, n):
    """the same pass on fp32 fields (BASELINE config 5's arithmetic; rows of up to 2048 points fit): transforms in fp32, the
    reduction in fp64 from the fp32 components, fp64 invariants out"""
    rng = np.random.default_rng(13 * n)
    rows, h = 11, n // 2 + 1
    x = (rng.standard_normal((6, rows, h)) + 1j * rng.standard_normal((6, rows, h))).astype(np.complex64).astype(np.complex128)
    kz = 2 * np.pi / n * np.arange(h)
    mul6 = (0, 0, 2, 0, 1, 1)
    d = [np.fft.irfft(x[j] * kfactor(mul6[j], kz)[None, :], n=n, axis=1) * n for j in range(6)]
    got, flag = run_invariants(L, n, x, rows, fb=4)
    assert flag == 0.0
    mu1 = d[0] + d[1] + d[2]
    mu2 = 0.5 * mu1 * mu1 - 0.5 * (d[0] ** 2 + d[1] ** 2 + d[2] ** 2) - (d[3] ** 2 + d[4] ** 2 + d[5] ** 2)
    mu3 = d[0] * d[1] * d[2] + 2 * d[3] * d[4] * d[5] - d[0] * d[5] ** 2 - d[1] * d[4] ** 2 - d[2] * d[3] ** 2
    amp = max(np.max(np.abs(v)) for v in d)
    for gi, wi, p in ((got[0], mu1, 1), (got[1], mu2, 2), (got[2], mu3, 3)):
        assert np.max(np.abs(gi - wi)) <= 8 * tol(4, n) * amp ** p, (n, p)


@pytest.mark.parametrize("n", [16, 64, 256, 512, 1024, 24, 40, 200, 768, 1000, 1536, 2000, 96, 120, 360, 720, 1080, 1440, 1728])   # (24 ...: k_mixed_c2r_invariants; from 96 on plans built in, reducing waves)
def test_invariant_zpass_lines(L, n):
    """six rows in, the three invariants of each cell's tensor out (what the solve of every radius but the last reads)"""
    rng = np.random.default_rng(11 * n)
    rows, h = 23, n // 2 + 1
    x = rng.standard_normal((6, rows, h)) + 1j * rng.standard_normal((6, rows, h))
    kz = 2 * np.pi / n * np.arange(h)
    mul6 = (0, 0, 2, 0, 1, 1)
    d = [np.fft.irfft(x[j] * kfactor(mul6[j], kz)[None, :], n=n, axis=1) * n for j in range(6)]
    got, flag = run_invariants(L, n, x, rows)
    assert flag == 0.0
    mu1 = d[0] + d[1] + d[2]
    mu2 = 0.5 * mu1 * mu1 - 0.5 * (d[0] ** 2 + d[1] ** 2 + d[2] ** 2) - (d[3] ** 2 + d[4] ** 2 + d[5] ** 2)
    mu3 = d[0] * d[1] * d[2] + 2 * d[3] * d[4] * d[5] - d[0] * d[5] ** 2 - d[1] * d[4] ** 2 - d[2] * d[3] ** 2
    amp = max(np.max(np.abs(v)) for v in d)
    for gi, wi, p in ((got[0], mu1, 1), (got[1], mu2, 2), (got[2], mu3, 3)):
        assert np.max(np.abs(gi - wi)) <= 8 * tol(8, n) * amp ** p, (n, p)


def test_invariant_zpass_flags_tensors_it_cannot_serve(L):
    """q == 0 in floating point: the reference takes the tensor's own diagonal (src/collapse_times.c:722-727).  Rows holding
    one kz = 1 mode make every cell of a row the same tensor d times cos(2 pi z / n)."""
    n, rows = 64, 4
    h = n // 2 + 1
    kz = 2 * np.pi / n
    fac = (1.0, 1.0, kz * kz, 1.0, kz, kz)  # the kz factors of the six components

    def flag_of(d):
        x = np.zeros((6, rows, h), dtype=np.complex128)
        for j in range(6):
            x[j, :, 1] = d[j] / (2.0 * fac[j])
        got, flag = run_invariants(L, n, x, rows)
        want = ((d[0] + d[1]) + d[2]) * np.cos(kz * np.arange(n))
        assert np.max(np.abs(got[0] - want[None, :])) <= 1e-14 * max(abs(v) for v in d) if any(d) else not got.any()
        return flag

    assert flag_of([0.0] * 6) == 0.0                                        # exactly isotropic (the empty field): mu1/3 IS the diagonal
    assert flag_of([1.0, 1.0 + 2.0 ** -30, 1.0, 0, 0, 0]) == 1.0             # anisotropy below sqrt(eps): q rounds to zero
    assert flag_of([1.0, 1.0, 1.0, 2.0 ** -30, 0, 0]) == 1.0                 # the same through an off-diagonal component
    assert flag_of([1e-170, 2e-170, -1e-170, 1e-171, 0, 0]) == 1.0           # squares underflow
    assert flag_of([1.0, 1.1, 0.9, 0.01, 0.0, 0.0]) == 0.0                   # an ordinary tensor


def test_packed_complex_algebra_of_the_sixteen_point_pass(L):
    """csrc/pf_fft16.h: a column's complex number is a (re, im) register pair and a +- i b, a * w are packed instructions whose
    operand halves are swapped / negated by modifiers in inline assembly -- each held against its definition"""
    rng = np.random.default_rng(16)
    cnt = 4096
    a = (rng.standard_normal(cnt) + 1j * rng.standard_normal(cnt)).astype(np.complex64)
    b = (rng.standard_normal(cnt) + 1j * rng.standard_normal(cnt)).astype(np.complex64)
    fp = C.POINTER(C.c_float)
    w16 = np.complex64(0.92387953251128675613 + 0.38268343236508977173j)
    want = {0: a + 1j * b, 1: a - 1j * b, 2: 1j * a, 3: -1j * a, 4: a * b, 5: a * np.conj(b), 6: a * b[0], 7: a * np.conj(b[0]),
            8: a * w16, 9: a * np.conj(w16), 10: a - 1j * b}
    for which, ref in want.items():
        out = np.zeros(cnt, dtype=np.complex64)
        rc = L.pf_debug_pk(which, a.view(np.float32).ctypes.data_as(fp), b.view(np.float32).ctypes.data_as(fp), out.view(np.float32).ctypes.data_as(fp), cnt)
        assert rc == 0
        assert np.max(np.abs(out - ref.astype(np.complex64))) <= 4e-7 * np.max(np.abs(ref)), which


@pytest.mark.parametrize("n", [6, 10, 14, 20, 22, 26, 36, 44, 50, 100])
def test_chirp_z_transforms_of_the_general_path(L, n):
    """csrc/pf_gfft.hip: the 3-D c2r / r2c of any even grid size -- prime factors 3, 5, 7, 11, 13 here -- as Bluestein convolutions on
    the power-of-two stages, against numpy's pocketfft (the reference plans any GridSize through FFTW / PFFT, src/fmax-pfft.c:139-188)"""
    rng = np.random.default_rng(n)
    real = rng.standard_normal((n, n, n))
    spec = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    assert L.pf_debug_gfft(n, -1, _dp(real), _dp(spec.view(np.float64))) == 0
    want = np.fft.rfftn(real)
    assert np.max(np.abs(spec - want)) <= 2e-14 * np.max(np.abs(want))
    s = rng.standard_normal(spec.shape) + 1j * rng.standard_normal(spec.shape)   # junk imaginary parts in the DC / Nyquist planes: ignored along z, like irfftn
    s = np.fft.rfftn(np.fft.irfftn(s, s=(n, n, n))) + 0j                           # ... but Hermitian in x and y, as every spectrum of a real field
    back = np.zeros((n, n, n))
    assert L.pf_debug_gfft(n, +1, _dp(np.ascontiguousarray(s).view(np.float64)), _dp(back)) == 0
    want = np.fft.irfftn(s, s=(n, n, n)) * n ** 3
    assert np.max(np.abs(back - want)) <= 2e-14 * np.max(np.abs(want))


@pytest.mark.parametrize("n", [300, 514, 700, 1022, 1026, 1500, 2046])
def test_chirp_z_lines_of_every_convolution_length(L, n):
    """one k_blue pass on a few lines: the convolution lengths M = 1024 (n = 300, 514), 2048 (700, 1022) and 4096 (1026, 1500, 2046 -- the
    only four-stage 8.8.8.8 plan of the library) that the whole-box tests above (M <= 512) never reach, in the three forms of the pass"""
    rng = np.random.default_rng(n)
    nl, h = 5, n // 2
    x = rng.standard_normal((n, nl)) + 1j * rng.standard_normal((n, nl))
    for d in (+1, -1):
        out = np.zeros((n, nl), dtype=np.complex128)
        assert L.pf_debug_gfft_lines(n, 0, d, nl, _dp(np.ascontiguousarray(x).view(np.float64)), _dp(out.view(np.float64))) == 0
        want = np.fft.ifft(x, axis=0) * n if d > 0 else np.fft.fft(x, axis=0)
        assert np.max(np.abs(out - want)) <= 4e-14 * np.max(np.abs(want)), (n, d)
    s = rng.standard_normal((nl, h + 1)) + 1j * rng.standard_normal((nl, h + 1))
    real = np.zeros((nl, n))
    assert L.pf_debug_gfft_lines(n, 1, +1, nl, _dp(np.ascontiguousarray(s).view(np.float64)), _dp(real)) == 0
    want = np.fft.irfft(s, n=n, axis=1) * n
    assert np.max(np.abs(real - want)) <= 4e-14 * np.max(np.abs(want)), n
    r = rng.standard_normal((nl, n))
    spec = np.zeros((nl, h + 1), dtype=np.complex128)
    assert L.pf_debug_gfft_lines(n, 2, -1, nl, _dp(r), _dp(spec.view(np.float64))) == 0
    want = np.fft.rfft(r, axis=1)
    assert np.max(np.abs(spec - want)) <= 4e-14 * np.max(np.abs(want)), n


@pytest.mark.parametrize("n,fb", [(2048, 4), (1024, 4), (1024, 8), (2048, 8), (256, 4), (768, 8), (768, 4), (200, 8), (200, 4), (120, 8)])
def test_strided_launch_with_several_jobs_per_tile(L, n, fb):
    """one launch, six jobs on three inputs as the y-pass of the sweep issues them (A2 -> 1, A1 -> 2, A0 -> 3 outputs with their
    k factors): tiles kept in registers over their jobs (k_strided, k_strided_pk8, k_mixed_strided with a compile-time plan) or
    read again per job (k_strided16, k_mixed_strided with a run-time plan)"""
    rng = np.random.default_rng(7 * n + fb)
    nouter, ncols, nin = 2, 37, 3
    x = rng.standard_normal((nin, nouter, n, ncols)) + 1j * rng.standard_normal((nin, nouter, n, ncols))
    in_of = np.array([0, 1, 1, 2, 2, 2], dtype=np.int32)
    mul = np.array([0, 1, 0, 2, 1, 3], dtype=np.int32)
    out = np.zeros((6, nouter, n, ncols), dtype=np.complex128)
    ip = C.POINTER(C.c_int)
    rc = L.pf_debug_strided_jobs(fb, n, 6, nin, in_of.ctypes.data_as(ip), mul.ctypes.data_as(ip), nouter, ncols,
                                 _dp(np.ascontiguousarray(x).view(np.float64)), _dp(out.view(np.float64)))
    assert rc == 0, L.pf_last_error()
    k = 2 * np.pi / n * signed(n)
    for j in range(6):
        want = np.fft.ifft(x[in_of[j]] * kfactor(int(mul[j]), k)[None, :, None], axis=1) * n
        assert np.max(np.abs(out[j] - want)) <= tol(fb, n) * np.max(np.abs(want)), (n, fb, j)
