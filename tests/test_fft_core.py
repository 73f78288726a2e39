"""Unit test of the device FFT header (pinocchio_amd/csrc/pf_fft_core.h) compiled
for the host: stage index algebra, butterflies, twiddles, real<->complex glue."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_emul", "fft_emul.cpp")
SO = os.path.join(HERE, "cpu_emul", "libfft_emul.so")


@pytest.fixture(scope="module")
def emul():
    hdr = os.path.join(HERE, "..", "pinocchio_amd", "csrc", "pf_fft_core.h")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", SO, SRC])
    L = C.CDLL(SO)
    dp = C.POINTER(C.c_double)
    for f in (L.emul_fft,):
        f.argtypes = [C.c_int, C.c_int, dp, dp]
    L.emul_c2r.argtypes = [C.c_int, dp, dp]
    L.emul_r2c.argtypes = [C.c_int, dp, dp]
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 256, 512, 1024, 2048])
@pytest.mark.parametrize("direction", [+1, -1])
def test_complex_line(emul, n, direction):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    out = np.empty(n, dtype=np.complex128)
    assert emul.emul_fft(n, direction, _dp(x.view(np.float64)), _dp(out.view(np.float64))) == 0
    want = np.fft.fft(x) if direction < 0 else np.fft.ifft(x) * n
    assert np.max(np.abs(out - want)) < 2e-15 * np.sqrt(n) * np.max(np.abs(want)) * 4


@pytest.mark.parametrize("n", [128, 1024])
@pytest.mark.parametrize("direction", [+1, -1])
def test_complex_line_paired_plan(emul, n, direction):
    """the paired plan of the strided passes (N = 16 * 8^k): radix 16 by thread pairs first, then radix 8 only"""
    emul.emul_fft_p16.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(n + 7)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    out = np.empty(n, dtype=np.complex128)
    assert emul.emul_fft_p16(n, direction, _dp(x.view(np.float64)), _dp(out.view(np.float64))) == 0
    want = np.fft.fft(x) if direction < 0 else np.fft.ifft(x) * n
    assert np.max(np.abs(out - want)) < 2e-15 * np.sqrt(n) * np.max(np.abs(want)) * 4


@pytest.mark.parametrize("n", [16, 32, 64, 256, 1024, 2048])
def test_real_lines(emul, n):
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n)
    spec = np.empty(n // 2 + 1, dtype=np.complex128)
    assert emul.emul_r2c(n, _dp(x), _dp(spec.view(np.float64))) == 0
    want = np.fft.rfft(x)
    assert np.max(np.abs(spec - want)) < 1e-13 * np.max(np.abs(want))
    # c2r of a spectrum with junk imaginary parts at DC/Nyquist: ignored, like irfft
    s = rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)
    back = np.empty(n)
    assert emul.emul_c2r(n, _dp(s.view(np.float64)), _dp(back)) == 0
    want = np.fft.irfft(s, n) * n
    assert np.max(np.abs(back - want)) < 1e-13 * np.max(np.abs(want))


@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("direction", [+1, -1])
def test_sixteen_points_per_thread_plan(alg, direction):
    """pf_fft16.h: 2048 = 16 x 16 x 8 with sixteen points per thread (the strided passes of 2048-point fp32 lines), its index
    algebra and butterflies on the host -- the double algebra (0) and the host form of the packed (re, im) fp32 algebra (1)"""
    src, so = os.path.join(HERE, "cpu_emul", "fft16_emul.cpp"), os.path.join(HERE, "cpu_emul", "libfft16_emul.so")
    hdr = os.path.join(HERE, "..", "pinocchio_amd", "csrc", "pf_fft16.h")
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.emul_fft16.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    n = 2048
    rng = np.random.default_rng(160 + alg)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    out = np.empty(n, dtype=np.complex128)
    assert L.emul_fft16(alg, direction, _dp(x.view(np.float64)), _dp(out.view(np.float64))) == 0
    want = np.fft.fft(x) if direction < 0 else np.fft.ifft(x) * n
    assert np.max(np.abs(out - want)) < (2e-15 if alg == 0 else 1e-6) * np.sqrt(n) * np.max(np.abs(want))
