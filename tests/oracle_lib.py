"""ctypes binding of the CPU oracle (oracle/libpf_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Builds the oracle with `make -C oracle` on first
use if the shared object is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libpf_oracle.so")
NBINS = 210

PRODUCT_DTYPE = np.dtype(
    [("Rmax", "<i4"), ("Fmax", "<f4"), ("Vel", "<f4", 3), ("Vel_2LPT", "<f4", 3),
     ("Vel_3LPT_1", "<f4", 3), ("Vel_3LPT_2", "<f4", 3)], align=False)
assert PRODUCT_DTYPE.itemsize == 56
# a -DDOUBLE_PRECISION_PRODUCTS build (PRODFLOAT double, src/pinocchio.h:219-225): natural alignment puts Fmax at byte 8
PRODUCT_DTYPE_DP = np.dtype(
    {"names": ["Rmax", "Fmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"],
     "formats": ["<i4", "<f8", ("<f8", 3), ("<f8", 3), ("<f8", 3), ("<f8", 3)],
     "offsets": [0, 8, 16, 40, 64, 88], "itemsize": 112})
SO_DP = os.path.join(ORACLE_DIR, "libpf_oracle_dp.so")

_lib = None
_lib_dp = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pf_oracle.c", "pf_oracle.h", "pf_genic.c", "pf_sng.c")]
    stale = any((not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs) for so in (SO, SO_DP))
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B"])
    return SO



def usable_cores() -> int:
    """the cores this process may really use: the hardware threads, cut by its CPU set and by the CPU-time quota of its control group
    (cpu.max: on the GPU boxes of this pool 16 of the host's 256 hardware threads -- 256 OpenMP threads on 16 cores' worth of quota only
    get throttled: the 2048^3 plane oracle took 310 s that way, round 6)"""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, cores)


def lib(double_products: bool = False):
    """the oracle, or its -DDOUBLE_PRECISION_PRODUCTS build"""
    global _lib, _lib_dp
    if (_lib_dp if double_products else _lib) is None:
        build()
        L = C.CDLL(SO_DP if double_products else SO)
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_density.argtypes = [C.c_void_p, dp]
        L.orc_set_invgrow.argtypes = [C.c_void_p, dp, dp, C.c_int]
        L.orc_set_growth.argtypes = [C.c_void_p, dp]
        L.orc_set_invgrow_radius.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int]
        L.orc_set_growth_table.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_compute_fmax.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, dp]
        L.orc_compute_second_derivatives.argtypes = [C.c_void_p, C.c_double]
        L.orc_compute_collapse_times.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_compute_displacements.argtypes = [C.c_void_p, C.c_int]
        L.orc_fmax_pdf.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        L.orc_products.restype = C.c_void_p
        L.orc_products.argtypes = [C.c_void_p]
        L.orc_second_derivative.restype = dp
        L.orc_second_derivative.argtypes = [C.c_void_p, C.c_int]
        L.orc_kvector.restype = dp
        L.orc_kvector.argtypes = [C.c_void_p, C.c_int]
        L.orc_c2r.argtypes = [C.c_void_p, dp, dp]
        L.orc_r2c.argtypes = [C.c_void_p, dp, dp]
        L.orc_ell_classic.restype = C.c_double
        L.orc_ell_classic.argtypes = [C.c_double] * 3
        L.orc_inverse_collapse_time.restype = C.c_double
        L.orc_inverse_collapse_time.argtypes = [C.c_void_p, dp, dp, dp, dp, C.POINTER(C.c_int)]
        L.orc_inverse_growing_mode.restype = C.c_double
        L.orc_inverse_growing_mode.argtypes = [C.c_void_p, C.c_double]
        L.orc_spline_eval.restype = C.c_double
        L.orc_spline_eval.argtypes = [C.c_void_p, C.c_double]
        L.orc_timers.argtypes = [C.c_void_p, dp]
        L.orc_set_tabulated_ct.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_ct_build.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_ct_table.restype = dp
        L.orc_ct_table.argtypes = [C.c_void_p]
        L.orc_ct_delta.restype = dp
        L.orc_ct_delta.argtypes = [C.c_void_p]
        L.orc_interpolate_collapse_time.restype = C.c_double
        L.orc_interpolate_collapse_time.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_ell_sng.restype = C.c_double
        L.orc_ell_sng.argtypes = [C.c_double] * 4 + [dp]
        L.orc_ell_sng_F.restype = C.c_double
        L.orc_ell_sng_F.argtypes = [C.c_double] * 4 + [dp]
        L.orc_set_collapse_model.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, dp]
        L.orc_set_modified_gravity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, dp]
        L.orc_select_sorted.restype = C.c_size_t
        L.orc_select_sorted.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_uint), C.POINTER(C.c_float)]
        ip = C.POINTER(C.c_int)
        L.orc_create_planes.restype = C.c_void_p
        L.orc_create_planes.argtypes = [C.c_int, C.c_int]
        L.orc_plane_derivatives.argtypes = [C.c_void_p, dp, C.c_double, C.c_int, ip, ip, C.c_int, ip, dp]
        L.orc_plane_collapse_times.argtypes = [C.c_void_p, C.c_int, C.c_size_t, dp, C.c_void_p, ip]
        L.orc_plane_acc_create.restype = C.c_void_p
        L.orc_plane_acc_create.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, ip, ip, C.c_int, ip]
        L.orc_plane_acc_add.argtypes = [C.c_void_p, dp, C.c_int, C.c_int]
        L.orc_plane_acc_finish.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_plane_acc_destroy.argtypes = [C.c_void_p]
        L.orc_plane_acc_power.restype = C.c_double
        L.orc_plane_acc_power.argtypes = [C.c_void_p, C.c_int]
        L.orc_plane_acc_destroy.restype = None
        if double_products:
            _lib_dp = L
        else:
            _lib = L
    return _lib_dp if double_products else _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One oracle context on an n^3 grid."""

    def __init__(self, n: int, nthreads: int = 0, double_products: bool = False):
        self.L = lib(double_products)
        self.product_dtype = PRODUCT_DTYPE_DP if double_products else PRODUCT_DTYPE
        self.n = n
        if nthreads <= 0:  # the slab loops have n iterations: more threads than n/4 only add fork/join cost (256-thread hosts)
            nthreads = max(1, min(usable_cores(), n // 4))
        self.h = self.L.orc_create(n, nthreads)
        if not self.h:
            raise ValueError("orc_create failed (n must be even and >= 4)")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_density(self, dk: np.ndarray):
        n = self.n
        dk = np.ascontiguousarray(dk, dtype=np.complex128)
        assert dk.shape == (n, n, n // 2 + 1)
        self.L.orc_set_density(self.h, _dp(dk.view(np.float64)))

    def set_invgrow(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        rc = self.L.orc_set_invgrow(self.h, _dp(x), _dp(y), len(x))
        assert rc == 0

    def set_growth(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        self.L.orc_set_growth(self.h, _dp(g))

    def set_invgrow_radius(self, ismooth, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert self.L.orc_set_invgrow_radius(self.h, int(ismooth), _dp(x), _dp(y), len(x)) == 0

    def set_growth_table(self, order, log10_growth, logkmin=-3.0, dlogk=0.5, sign=1.0):
        t = np.ascontiguousarray(log10_growth, dtype=np.float64)
        assert self.L.orc_set_growth_table(self.h, int(order), _dp(t), len(t), logkmin, dlogk, sign) == 0

    def compute_fmax(self, radii_cells, do_lpt=True):
        r = np.ascontiguousarray(radii_cells, dtype=np.float64)
        tv = np.zeros(len(r))
        rc = self.L.orc_compute_fmax(self.h, len(r), _dp(r), int(do_lpt), _dp(tv))
        if rc:
            raise RuntimeError("orc_compute_fmax failed")
        return tv

    def second_derivatives(self, rs_cells: float):
        rc = self.L.orc_compute_second_derivatives(self.h, float(rs_cells))
        assert rc == 0
        n = self.n
        return [np.ctypeslib.as_array(self.L.orc_second_derivative(self.h, i), shape=(n, n, n)).copy()
                for i in range(6)]

    def collapse_times(self, ismooth: int) -> float:
        tv = C.c_double(0)
        rc = self.L.orc_compute_collapse_times(self.h, ismooth, C.byref(tv))
        assert rc == 0
        return tv.value

    def displacements(self, compute_sources=True):
        rc = self.L.orc_compute_displacements(self.h, int(compute_sources))
        assert rc == 0

    def set_collapse_model(self, model, cosmo=None, d_in=None):
        """0: ELL_CLASSIC; 1: ELL_SNG with cosmo = (Omega0, OmegaLambda, OmegaRad, OmegaK), D_in per radius"""
        cosmo = np.ascontiguousarray(cosmo if cosmo is not None else np.zeros(4), dtype=np.float64)
        d_in = np.ascontiguousarray(d_in if d_in is not None else np.zeros(1), dtype=np.float64)
        assert self.L.orc_set_collapse_model(self.h, model, _dp(cosmo), len(d_in), _dp(d_in)) == 0

    def set_modified_gravity(self, fr0, h_over_c=100.0 / 299792.458, size=None):
        sz = np.ascontiguousarray(size if size is not None else np.zeros(1), dtype=np.float64)
        assert self.L.orc_set_modified_gravity(self.h, fr0, h_over_c, len(sz), _dp(sz)) == 0

    def set_tabulated_ct(self, variance):
        v = np.ascontiguousarray(variance, dtype=np.float64)
        assert self.L.orc_set_tabulated_ct(self.h, len(v), _dp(v) if len(v) else None) == 0

    def ct_build(self, ismooth: int, variance: float):
        """-> (CT_table[iy][ix][id], delta_vector[id]) of initialize_collapse_times (src/collapse_times.c:820)"""
        assert self.L.orc_ct_build(self.h, ismooth, variance) == 0
        t = np.ctypeslib.as_array(self.L.orc_ct_table(self.h), shape=(50, 50, 100)).copy()
        d = np.ctypeslib.as_array(self.L.orc_ct_delta(self.h), shape=(100,)).copy()
        return t, d

    def set_ct_interpolation(self, flavour: int):
        """0 BILINEAR_SPLINE (the source's define), 1 -DTRILINEAR, 2 -DALL_SPLINE"""
        self.L.orc_set_ct_interpolation.argtypes = [C.c_void_p, C.c_int]
        assert self.L.orc_set_ct_interpolation(self.h, int(flavour)) == 0

    def interpolate_collapse_time(self, l1, l2, l3):
        return self.L.orc_interpolate_collapse_time(self.h, l1, l2, l3)

    def select_sorted(self, flast: float):
        idx = np.empty(self.n ** 3, dtype=np.uint32)
        f = np.empty(self.n ** 3, dtype=np.float32)
        m = self.L.orc_select_sorted(self.h, flast, idx.ctypes.data_as(C.POINTER(C.c_uint)), f.ctypes.data_as(C.POINTER(C.c_float)))
        return idx[:m].copy(), f[:m].copy()

    def products(self) -> np.ndarray:
        n = self.n
        ptr = self.L.orc_products(self.h)
        buf = (C.c_char * (self.product_dtype.itemsize * n ** 3)).from_address(ptr)
        return np.frombuffer(buf, dtype=self.product_dtype).reshape(n, n, n).copy()

    def kvector(self, which: int) -> np.ndarray:
        n = self.n
        a = np.ctypeslib.as_array(self.L.orc_kvector(self.h, which), shape=(n, n, n // 2 + 1, 2)).copy()
        return a.view(np.complex128)[..., 0]

    def fmax_pdf(self) -> np.ndarray:
        h = (C.c_ulonglong * NBINS)()
        self.L.orc_fmax_pdf(self.h, h)
        return np.array(h[:], dtype=np.uint64)

    def c2r(self, spec: np.ndarray) -> np.ndarray:
        n = self.n
        spec = np.ascontiguousarray(spec, dtype=np.complex128)
        out = np.empty((n, n, n))
        self.L.orc_c2r(self.h, _dp(spec.view(np.float64)), _dp(out))
        return out

    def r2c(self, real: np.ndarray) -> np.ndarray:
        n = self.n
        real = np.ascontiguousarray(real, dtype=np.float64)
        out = np.empty((n, n, n // 2 + 1), dtype=np.complex128)
        self.L.orc_r2c(self.h, _dp(real), _dp(out.view(np.float64)))
        return out

    def inverse_collapse_time(self, d):
        d = np.ascontiguousarray(d, dtype=np.float64)
        x1, x2, x3 = C.c_double(), C.c_double(), C.c_double()
        fail = C.c_int()
        f = self.L.orc_inverse_collapse_time(self.h, _dp(d), C.byref(x1), C.byref(x2), C.byref(x3), C.byref(fail))
        return f, (x1.value, x2.value, x3.value), fail.value

    def inverse_growing_mode(self, D: float) -> float:
        return self.L.orc_inverse_growing_mode(self.h, float(D))

    def spline_eval(self, x: float) -> float:
        return self.L.orc_spline_eval(self.h, float(x))

    def timers(self):
        t = np.zeros(5)
        self.L.orc_timers(self.h, _dp(t))
        return dict(total=t[0], deriv=t[1], fft=t[2], coll=t[3], lpt=t[4])


class PlaneOracle:
    """The oracle on sampled x-planes of a box whose whole oracle would not fit the host (orc_plane_*, oracle/pf_oracle.c):
    the reference's filter per mode, its transforms with the x-transform written as the plain sum for the sampled planes,
    its per-cell collapse pass with the running maximum."""

    HESSIAN = ((1, 1), (2, 2), (3, 3), (1, 2), (1, 3), (2, 3))  # storage order 11,22,33,12,13,23 (src/LPT.c:36-44)

    def __init__(self, n: int, planes, nthreads: int = 0):
        self.L = lib()
        self.n = n
        self.planes = np.ascontiguousarray(planes, dtype=np.int32)
        if nthreads <= 0:
            nthreads = max(1, min(usable_cores(), n // 4))
        self.h = self.L.orc_create_planes(n, nthreads)
        if not self.h:
            raise ValueError("orc_create_planes failed")
        ncell = len(self.planes) * n * n
        self.fmax = np.empty(ncell, dtype=np.float32)
        self.rmax = np.empty(ncell, dtype=np.int32)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_invgrow(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert self.L.orc_set_invgrow(self.h, _dp(x), _dp(y), len(x)) == 0

    def derivatives(self, spec: np.ndarray, rs_cells: float, comps) -> np.ndarray:
        """-> [ncomp][nplanes][n][n]: compute_derivative(ia, ib) of `spec` at smoothing rs_cells on the sampled planes"""
        n = self.n
        assert spec.dtype == np.complex128 and spec.shape == (n, n, n // 2 + 1) and spec.flags.c_contiguous
        ia = np.ascontiguousarray([c[0] for c in comps], dtype=np.int32)
        ib = np.ascontiguousarray([c[1] for c in comps], dtype=np.int32)
        out = np.empty((len(comps), len(self.planes), n, n))
        ip = C.POINTER(C.c_int)
        rc = self.L.orc_plane_derivatives(self.h, _dp(spec.view(np.float64)), float(rs_cells), len(comps), ia.ctypes.data_as(ip),
                                          ib.ctypes.data_as(ip), len(self.planes), self.planes.ctypes.data_as(ip), _dp(out))
        assert rc == 0
        return out

    def stream(self, radii_cells, comps):
        """the streaming form of derivatives() for a spectrum that does not fit the host: -> an accumulator with
        add(rows, kx0) for consecutive pieces [nkx][n][n/2+1] of the spectrum (ascending kx, all of them) and finish(irad) ->
        [ncomp][nplanes][n][n] of radius irad (orc_plane_acc_*)"""
        return PlaneStream(self, radii_cells, comps)

    def collapse_times(self, ismooth: int, d6: np.ndarray):
        """the collapse pass of radius ismooth on the planes' cells; d6 = [6][nplanes][n][n]"""
        d6 = np.ascontiguousarray(d6, dtype=np.float64).reshape(6, -1)
        assert d6.shape[1] == len(self.fmax)
        rc = self.L.orc_plane_collapse_times(self.h, int(ismooth), d6.shape[1], _dp(d6), self.fmax.ctypes.data_as(C.c_void_p),
                                             self.rmax.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0

    def sweep(self, dk: np.ndarray, radii_cells):
        """compute_fmax's loop over the radii (src/fmax.c:81-157) on the sampled planes -> (Fmax, Rmax) [nplanes][n][n]"""
        for ismooth, rs in enumerate(radii_cells):
            self.collapse_times(ismooth, self.derivatives(dk, rs, self.HESSIAN))
        shape = (len(self.planes), self.n, self.n)
        return self.fmax.reshape(shape), self.rmax.reshape(shape)


class PlaneStream:
    def __init__(self, po: PlaneOracle, radii_cells, comps):
        self.po, self.L, self.n = po, po.L, po.n
        self.ncomp, self.nplanes = len(comps), len(po.planes)
        r = np.ascontiguousarray(radii_cells, dtype=np.float64)
        ia = np.ascontiguousarray([c[0] for c in comps], dtype=np.int32)
        ib = np.ascontiguousarray([c[1] for c in comps], dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.h = self.L.orc_plane_acc_create(po.h, len(r), _dp(r), len(comps), ia.ctypes.data_as(ip), ib.ctypes.data_as(ip),
                                             len(po.planes), po.planes.ctypes.data_as(ip))
        if not self.h:
            raise ValueError("orc_plane_acc_create failed")

    def add(self, rows: np.ndarray, kx0: int):
        n = self.n
        assert rows.dtype == np.complex128 and rows.shape[1:] == (n, n // 2 + 1) and rows.flags.c_contiguous
        assert self.L.orc_plane_acc_add(self.h, _dp(rows.view(np.float64)), int(kx0), rows.shape[0]) == 0

    def power(self, irad: int) -> float:
        """sum over the modes added so far of |spec|^2 W(k R)^2, both halves of the spectrum: n^6 times the variance of the smoothed field"""
        return float(self.L.orc_plane_acc_power(self.h, int(irad)))

    def finish(self, irad: int) -> np.ndarray:
        out = np.empty((self.ncomp, self.nplanes, self.n, self.n))
        assert self.L.orc_plane_acc_finish(self.h, int(irad), _dp(out)) == 0
        return out

    def close(self):
        if self.h:
            self.L.orc_plane_acc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def ell_classic(l1, l2, l3) -> float:
    return lib().orc_ell_classic(float(l1), float(l2), float(l3))
