"""Host-side mirror of the reference's Fmax interface on top of the C ABI.

Method names and argument meaning follow the reference (src/pinocchio.h:545-562,
637-647): compute_fmax, compute_second_derivatives, compute_collapse_times,
compute_displacements, Fmax_PDF.  All compute runs in libpinfmax_hip.so on the
GPU; numpy is only the host container of inputs and outputs.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

PRODUCT_DTYPE = np.dtype(
    [("Rmax", "<i4"), ("Fmax", "<f4"), ("Vel", "<f4", 3), ("Vel_2LPT", "<f4", 3),
     ("Vel_3LPT_1", "<f4", 3), ("Vel_3LPT_2", "<f4", 3)], align=False)  # src/pinocchio.h:233-259, 56 B
# the same record of a -DDOUBLE_PRECISION_PRODUCTS build (PRODFLOAT double, src/pinocchio.h:219-225): 112 B
PRODUCT_DTYPE_DP = np.dtype(
    {"names": ["Rmax", "Fmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"],
     "formats": ["<i4", "<f8", ("<f8", 3), ("<f8", 3), ("<f8", 3), ("<f8", 3)],
     "offsets": [0, 8, 16, 40, 64, 88], "itemsize": 112})


class PinfmaxError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def plan_bytes(n: int, nranks: int = 1, field_bytes: int = 8, double_products: bool = False):
    """(bytes pf_create allocates, bytes held at most) per rank of this configuration -- pf_plan_bytes; no device needed"""
    L = _lib.load()
    cfg = _lib.Config(n=n, rank=0, nranks=nranks, device=0, field_bytes=field_bytes, flags=_lib.FLAG_DOUBLE_PRODUCTS if double_products else 0)
    a, b = C.c_size_t(), C.c_size_t()
    if L.pf_plan_bytes(C.byref(cfg), C.byref(a), C.byref(b)):
        raise PinfmaxError("pf_plan_bytes: bad configuration")
    return a.value, b.value


class Fmax:
    """One rank's context: an x-slab of an n^3 grid on one MI355X."""

    def __init__(self, n: int, rank: int = 0, nranks: int = 1, device: int = 0, field_bytes: int = 8,
                 timing: bool = False, double_products: bool = False):
        self.L = _lib.load()
        self.double_products = bool(double_products)
        self.n, self.rank, self.nranks = int(n), int(rank), int(nranks)
        self.nxl = self.n // self.nranks
        cfg = _lib.Config(n=n, rank=rank, nranks=nranks, device=device, field_bytes=field_bytes,
                          flags=(_lib.FLAG_TIMING if timing else 0) | (_lib.FLAG_DOUBLE_PRODUCTS if double_products else 0))
        h = C.c_void_p()
        self._chk(self.L.pf_create(C.byref(h), C.byref(cfg)))
        self.h = h
        self._keep = []

    # -- plumbing ---------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise PinfmaxError(self.L.pf_last_error().decode() or f"error {rc}")

    def close(self):
        if getattr(self, "h", None):
            self.L.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        self._chk(self.L.pf_synchronize(self.h))

    @property
    def device_bytes(self) -> int:
        return int(self.L.pf_device_bytes(self.h))

    # -- inputs -----------------------------------------------------------
    def set_density(self, dk: np.ndarray):
        """kdensity[0]: this rank's x-slab [nxl][n][n/2+1] complex128."""
        dk = np.ascontiguousarray(dk, dtype=np.complex128)
        if dk.shape != (self.nxl, self.n, self.n // 2 + 1):
            raise ValueError(f"density slab shape {dk.shape}")
        self._chk(self.L.pf_set_density(self.h, _dp(dk.view(np.float64))))

    def synth_density(self, seed: int, sigma0: float = 2.5, slope: float = -2.0):
        self._chk(self.L.pf_synth_density(self.h, C.c_uint64(seed), sigma0, slope))

    def genic_density(self, seed: int, box_true_mpc: float, omega0: float, omega_baryon: float, hubble100: float,
                      primordial_index: float, sigma8: float = 0.0, pknorm: float = 0.0, fixed: bool = False, paired: bool = False,
                      pk_table=None, spectrum: str = "EH", wdm_mass_kev: float = 0.0) -> float:
        """GenIC_large (src/GenIC.c:73) on the device.  Give PkNorm, or sigma8 to have it computed
        (normalize_PowerSpectrum, src/cosmo.c:1058).  pk_table = (log10 k [1/Mpc], log10(k^3 P)): a tabulated spectrum
        (SPLINE[SP_PK]) instead of Eisenstein & Hu; PkNorm is then what the caller says (1 for a trusted table).
        spectrum "Efstathiou" / "PowerLaw": the other two analytic forms of PowerSpectrum(); wdm_mass_kev > 0: its warm-dark-matter
        cut-off (src/cosmo.c:953-1007).  Returns the PkNorm used."""
        p = _lib.GenicParams(omega0, omega_baryon, hubble100, primordial_index, box_true_mpc, pknorm, seed, int(fixed), int(paired))
        p.spectrum = {"EH": 0, "Efstathiou": 3, "PowerLaw": 4}[spectrum]   # FileWithInputSpectrum (src/cosmo.c:1009-1046)
        p.WDM_PartMass_in_kev = wdm_mass_kev
        if pk_table is not None:
            lk = np.ascontiguousarray(pk_table[0], dtype=np.float64)
            lp = np.ascontiguousarray(pk_table[1], dtype=np.float64)
            assert lk.shape == lp.shape and lk.ndim == 1
            p.pk_n, p.pk_logk, p.pk_logk3p = len(lk), _dp(lk), _dp(lp)
            if pknorm <= 0.0:
                p.PkNorm = 1.0
        elif pknorm <= 0.0:
            v = C.c_double()
            self._chk(self.L.pf_pk_norm(C.byref(p), sigma8, C.byref(v)))
            p.PkNorm = v.value
        self._chk(self.L.pf_genic_density(self.h, C.byref(p)))
        return p.PkNorm

    def set_invgrow(self, x, y, ismooth: int = -1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self._chk(self.L.pf_set_invgrow(self.h, ismooth, _dp(x), _dp(y), len(x)))

    def set_growth(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        assert g.shape == (4,)
        self._chk(self.L.pf_set_growth(self.h, _dp(g)))

    def set_growth_table(self, order: int, log10_growth, logkmin: float = -3.0, dlogk: float = 0.5, sign: float = 1.0):
        """k-binned growth of ScaleDep.order = order (SCALE_DEPENDENT build, src/cosmo.c:1728-1755); empty: scalar"""
        t = np.ascontiguousarray(log10_growth, dtype=np.float64)
        self._chk(self.L.pf_set_growth_table(self.h, int(order), _dp(t) if len(t) else None, len(t), logkmin, dlogk, sign))

    def set_collapse_model(self, model: int, cosmo=None, d_in=None):
        """0: ELL_CLASSIC; 1: ELL_SNG (table only) with cosmo = (Omega0, OmegaLambda, OmegaRad, OmegaK), D_in per radius"""
        cs = np.ascontiguousarray(cosmo, dtype=np.float64) if cosmo is not None else None
        di = np.ascontiguousarray(d_in, dtype=np.float64) if d_in is not None else None
        self._chk(self.L.pf_set_collapse_model(self.h, model, _dp(cs) if cs is not None else None, len(di) if di is not None else 0,
                                               _dp(di) if di is not None else None))

    def set_modified_gravity(self, fr0: float, h_over_c: float = 100.0 / 299792.458, size=None):
        """MOD_GRAV_FR force modification inside the ELL_SNG system (src/collapse_times.c:295-312); fr0 = 0: off"""
        sz = np.ascontiguousarray(size, dtype=np.float64) if size is not None else None
        self._chk(self.L.pf_set_modified_gravity(self.h, fr0, h_over_c, len(sz) if sz is not None else 0, _dp(sz) if sz is not None else None))

    def set_tabulated_ct(self, variance):
        """TABULATED_CT build: Smoothing.Variance[] per radius ([] = direct solve), src/collapse_times.c:780-1231"""
        v = np.ascontiguousarray(variance, dtype=np.float64)
        self._chk(self.L.pf_set_tabulated_ct(self.h, len(v), _dp(v) if len(v) else None))

    def ct_build(self, ismooth: int, variance: float) -> np.ndarray:
        """initialize_collapse_times(ismooth): -> CT_table[iy][ix][id]"""
        t = np.empty((50, 50, 100))
        self._chk(self.L.pf_ct_build(self.h, ismooth, variance, _dp(t)))
        return t

    def ct_load(self, ismooth: int, variance: float, table: np.ndarray):
        t = np.ascontiguousarray(table, dtype=np.float64)
        assert t.shape == (50, 50, 100)
        self._chk(self.L.pf_ct_load(self.h, ismooth, variance, _dp(t)))

    # -- the path (reference names) ----------------------------------------
    def sweep(self, radii_cells) -> np.ndarray:
        """radius loop of compute_fmax (src/fmax.c:66-150) -> TrueVariance[]"""
        r = np.ascontiguousarray(radii_cells, dtype=np.float64)
        tv = np.zeros(len(r))
        self._chk(self.L.pf_sweep(self.h, len(r), _dp(r), _dp(tv)))
        return tv

    def compute_fmax(self, radii_cells, do_lpt: bool = True) -> np.ndarray:
        """compute_fmax (src/fmax.c:36-190): sweep, then compute_displacements(1,0,z)"""
        self._chk(self.L.pf_set_sources_in_sweep(self.h, 1 if do_lpt else 0))
        try:
            tv = self.sweep(radii_cells)
        finally:
            self.L.pf_set_sources_in_sweep(self.h, 0)
        if do_lpt:
            self.compute_displacements(1, 0)
        return tv

    def compute_second_derivatives(self, radius_cells: float):
        self._chk(self.L.pf_second_derivatives(self.h, float(radius_cells)))

    def compute_collapse_times(self, ismooth: int) -> float:
        tv = C.c_double()
        self._chk(self.L.pf_collapse_times(self.h, ismooth, C.byref(tv)))
        return tv.value

    def set_transposed_spectra(self, on: bool):
        """spectra cross the interface as [ky_local][kx][kz] (params.use_transposed_fft on slabs)"""
        self._chk(self.L.pf_set_transposed_spectra(self.h, 1 if on else 0))

    def set_ct_interpolation(self, flavour: int):
        """0 BILINEAR_SPLINE (default), 1 -DTRILINEAR, 2 -DALL_SPLINE"""
        self._chk(self.L.pf_set_ct_interpolation(self.h, int(flavour)))

    def set_lpt_order(self, order: int):
        """3: -DTWO_LPT -DTHREE_LPT (default); 2: -DTWO_LPT only; 1: Zel'dovich displacements only"""
        self._chk(self.L.pf_set_lpt_order(self.h, int(order)))

    def compute_displacements(self, compute_sources: int = 1, recompute_sd: int = 0):
        self._chk(self.L.pf_displacements(self.h, int(compute_sources), int(recompute_sd)))

    def Fmax_PDF(self) -> np.ndarray:
        h = (C.c_ulonglong * _lib.NBINS)()
        self._chk(self.L.pf_fmax_pdf(self.h, h))
        return np.array(h[:], dtype=np.uint64)

    # -- outputs ----------------------------------------------------------
    def products(self) -> np.ndarray:
        lay = _lib.ProductLayout()
        self.L.pf_layout_3lpt(C.byref(lay))
        dtype = PRODUCT_DTYPE
        if self.double_products:      # PRODFLOAT double: the natural alignment of the reference's struct (Fmax at byte 8)
            dtype = PRODUCT_DTYPE_DP
            lay.stride, lay.off_Rmax, lay.off_Fmax = 112, 0, 8
            lay.off_Vel, lay.off_Vel_2LPT, lay.off_Vel_3LPT_1, lay.off_Vel_3LPT_2 = 16, 40, 64, 88
        out = np.zeros((self.nxl, self.n, self.n), dtype=dtype)
        self._chk(self.L.pf_get_products(self.h, out.ctypes.data_as(C.c_void_p), C.byref(lay)))
        return out

    def update_products(self, records: np.ndarray, layout) -> np.ndarray:
        """merge the device columns named by `layout` (a _lib.ProductLayout) into caller-held AoS records"""
        assert records.flags.c_contiguous and records.nbytes == self.nxl * self.n * self.n * layout.stride
        self._chk(self.L.pf_update_products(self.h, records.ctypes.data_as(C.c_void_p), C.byref(layout)))
        return records

    def select_sorted(self, flast: float):
        """cells with Fmax >= flast by descending Fmax (src/distribute.c:695, src/fragment.c:484-503) -> (index, Fmax)"""
        cnt = C.c_size_t()
        self._chk(self.L.pf_select_sorted(self.h, flast, 0, None, None, C.byref(cnt)))
        idx = np.empty(cnt.value, dtype=np.uint32)
        f = np.empty(cnt.value, dtype=np.float32)
        if cnt.value:
            self._chk(self.L.pf_select_sorted(self.h, flast, cnt.value, idx.ctypes.data_as(C.POINTER(C.c_uint)),
                                              f.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cnt)))
        return idx, f

    def block(self, name: str, id_bytes: int = 4) -> np.ndarray:
        """one block of the timeless snapshot (src/write_snapshot.c:207-342) for this rank's slab"""
        nc = self.nxl * self.n * self.n
        if name == "ID  ":
            out = np.empty(nc, dtype=np.uint64 if id_bytes == 8 else np.uint32)
        elif name == "RMAX":
            out = np.empty(nc, dtype=np.int32)
        elif name == "FMAX":
            out = np.empty(nc, dtype=np.float32)
        else:
            out = np.empty((nc, 3), dtype=np.float32)
        self._chk(self.L.pf_get_block(self.h, name.encode(), id_bytes, out.ctypes.data_as(C.c_void_p)))
        return out

    def second_derivative(self, i: int) -> np.ndarray:
        out = np.empty((self.nxl, self.n, self.n))
        self._chk(self.L.pf_get_second_derivative(self.h, i, _dp(out)))
        return out

    def kvector(self, which: int) -> np.ndarray:
        out = np.empty((self.nxl, self.n, self.n // 2 + 1), dtype=np.complex128)
        self._chk(self.L.pf_get_kvector(self.h, which, _dp(out.view(np.float64))))
        return out

    def density(self) -> np.ndarray:
        out = np.empty((self.nxl, self.n, self.n // 2 + 1), dtype=np.complex128)
        self._chk(self.L.pf_get_density(self.h, _dp(out.view(np.float64))))
        return out

    def replicated_rows(self, kx0: int, nkx: int) -> np.ndarray:
        """rows kx0 .. kx0 + nkx - 1 of the whole delta(k) a PF_REPLICATE_DK context transforms -> [nkx][n][n/2+1] complex128"""
        out = np.empty((nkx, self.n, self.n // 2 + 1), dtype=np.complex128)
        self._chk(self.L.pf_debug_replicated_rows(self.h, int(kx0), int(nkx), _dp(out.view(np.float64))))
        return out

    def forward_transform(self, real: np.ndarray) -> np.ndarray:
        real = np.ascontiguousarray(real, dtype=np.float64)
        assert real.shape == (self.nxl, self.n, self.n)
        out = np.empty((self.nxl, self.n, self.n // 2 + 1), dtype=np.complex128)
        self._chk(self.L.pf_forward_transform(self.h, _dp(real), _dp(out.view(np.float64))))
        return out

    def reverse_transform(self, spec: np.ndarray) -> np.ndarray:
        spec = np.ascontiguousarray(spec, dtype=np.complex128)
        assert spec.shape == (self.nxl, self.n, self.n // 2 + 1)
        out = np.empty((self.nxl, self.n, self.n))
        self._chk(self.L.pf_reverse_transform(self.h, _dp(spec.view(np.float64)), _dp(out)))
        return out

    def compute_derivative(self, spec: np.ndarray, first_derivative: int, second_derivative: int, rs_cells: float = 0.0,
                           order: int = 0) -> np.ndarray:
        """compute_derivative (src/fmax-pfft.c:255-441) of a caller-held spectrum slab -> real slab"""
        spec = np.ascontiguousarray(spec, dtype=np.complex128)
        assert spec.shape == (self.nxl, self.n, self.n // 2 + 1)
        out = np.empty((self.nxl, self.n, self.n))
        self._chk(self.L.pf_derivative(self.h, _dp(spec.view(np.float64)), first_derivative, second_derivative,
                                       float(rs_cells), int(order), _dp(out)))
        return out

    def collapse_cells(self, d: np.ndarray, ismooth: int = -1) -> np.ndarray:
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(-1, 6)
        out = np.empty(len(d))
        self._chk(self.L.pf_collapse_cells(self.h, ismooth, _dp(d), len(d), _dp(out)))
        return out

    # -- measurement ------------------------------------------------------
    def cputime(self) -> dict:
        t = _lib.CpuTime()
        self._chk(self.L.pf_get_cputime(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def reset_cputime(self):
        self._chk(self.L.pf_reset_cputime(self.h))

    def kernel_stats(self) -> list:
        arr = (_lib.KernelStat * 32)()
        n = C.c_int()
        self._chk(self.L.pf_kernel_stats(self.h, arr, 32, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=arr[i].total_ms,
                     alg_bytes=arr[i].alg_bytes) for i in range(n.value)]

    def reset_kernel_stats(self):
        self._chk(self.L.pf_reset_kernel_stats(self.h))
