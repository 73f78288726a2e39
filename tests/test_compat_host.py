"""The reference-named C host layer (pinocchio_amd/host/pf_compat.c): reads like
the reference's own driver -- fill the globals, set_one_grid, compute_fft_plans,
compute_fmax -- and is checked against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from pinocchio_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "pinocchio_amd", "libpf_compat.so")

GROWTH_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double)
HUBBLE_FN = C.CFUNCTYPE(C.c_double, C.c_double)


class Smoothing(C.Structure):
    _fields_ = [("Nsmooth", C.c_int), ("Radius", C.POINTER(C.c_double)), ("Variance", C.POINTER(C.c_double)),
                ("TrueVariance", C.POINTER(C.c_double))]


class Grid(C.Structure):
    _fields_ = [("total_local_size", C.c_uint), ("total_local_size_fft", C.c_uint), ("off", C.c_uint),
                ("ParticlesPerTask", C.c_uint), ("GSglobal", C.c_ssize_t * 3), ("GSlocal", C.c_ssize_t * 3),
                ("GSstart", C.c_ssize_t * 3), ("GSlocal_k", C.c_ssize_t * 3), ("GSstart_k", C.c_ssize_t * 3),
                ("lower_k_cutoff", C.c_double), ("upper_k_cutoff", C.c_double), ("norm", C.c_double),
                ("BoxSize", C.c_double), ("CellSize", C.c_double), ("Ntotal", C.c_ulonglong)]


class ScaleDep(C.Structure):
    _fields_ = [("nseg", C.c_int), ("myseg", C.c_int), ("no_interp", C.c_int), ("order", C.c_int),
                ("z", C.c_double * 100), ("D", C.c_double * 100), ("D2", C.c_double * 100), ("D31", C.c_double * 100),
                ("D32", C.c_double * 100), ("redshift", C.c_double)]


class Params(C.Structure):
    _fields_ = [("RunFlag", C.c_char * 100), ("DumpDir", C.c_char * 100), ("GridSize", C.c_int * 3), ("RandomSeed", C.c_int),
                ("Omega0", C.c_double), ("OmegaBaryon", C.c_double), ("Hubble100", C.c_double), ("Sigma8", C.c_double),
                ("PrimordialIndex", C.c_double), ("BoxSize_htrue", C.c_double), ("OmegaLambda", C.c_double),
                ("CTtableFile", C.c_char * 400), ("use_transposed_fft", C.c_int), ("FixedIC", C.c_int), ("PairedIC", C.c_int)]


def c_output(capfd):
    """what the C side printed since the last call (its stdio buffer is flushed first)"""
    C.CDLL(None).fflush(None)
    return capfd.readouterr().out


class Knots(C.Structure):
    _fields_ = [("size", C.c_size_t), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double))]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(SO) or not os.path.exists(os.path.join(ROOT, "pinocchio_amd", "libpinfmax_hip.so")):
        g.build()
    return C.CDLL(SO)


def test_reference_symbols_and_geometry(lib):
    for name in ("set_one_grid", "compute_fft_plans", "finalize_fft", "compute_fmax", "compute_displacements",
                 "compute_collapse_times", "compute_LPT_displacements", "Fmax_PDF", "dump_products", "read_dumps", "fdate"):
        assert hasattr(lib, name), name
    grid = Grid.in_dll(lib, "grid0") if hasattr(lib, "grid0") else C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    n = 64
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = 128.0
    assert lib.set_one_grid(0) == 0
    # src/fmax-pfft.c:80-134 for one rank: local = global, half-spectrum on z, norm = 1/N^3, CellSize = Box/N
    assert list(grid.GSlocal) == [n, n, n] and list(grid.GSlocal_k) == [n, n, n // 2 + 1]
    assert grid.total_local_size == n ** 3 and grid.total_local_size_fft == 2 * n * n * (n // 2 + 1)
    assert grid.norm == 1.0 / n ** 3 and grid.CellSize == 2.0
    lib.fdate.restype = C.c_char_p
    assert len(lib.fdate()) == 24


def test_fdate_is_the_references_rearranged_ctime(lib):
    """src/fmax.c:261-289: characters 0-9 of ctime(), then its year, then its time of day"""
    import time
    lib.fdate.restype = C.c_char_p
    libc = C.CDLL(None)
    libc.ctime.restype = C.c_char_p
    for _ in range(3):
        t0 = int(time.time())
        got = lib.fdate().decode()
        t1 = int(time.time())
        want = []
        for t in range(t0 - 1, t1 + 1):   # (C's time(NULL) reads a coarser clock: just after a second boundary it is one behind)
            s = libc.ctime(C.byref(C.c_long(t))).decode()          # "Www Mmm dd hh:mm:ss yyyy\n"
            want.append(s[:10] + s[19:24] + s[10:19])
        assert got in want, (got, want)


def test_geometry_refuses_what_the_path_does_not_do(lib, capfd):
    """ragged slabs and non-cubic grids fail loudly, in the reference's format; UseTransposedFFT (src/fmax-pfft.c:92, 271-281)
    is a geometry: the k-space extents become those of a ky-slab"""
    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    par = Params.in_dll(lib, "params")
    ntasks = C.c_int.in_dll(lib, "NTasks")
    n = 64
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = 128.0
    try:
        par.use_transposed_fft = 1
        ntasks.value = 4
        C.c_int.in_dll(lib, "ThisTask").value = 3
        assert lib.set_one_grid(0) == 0
        assert [grid.GSlocal_k[i] for i in range(3)] == [n, n // 4, n // 2 + 1] and [grid.GSstart_k[i] for i in range(3)] == [0, 3 * n // 4, 0]
        assert [grid.GSlocal[i] for i in range(3)] == [n // 4, n, n] and grid.total_local_size_fft == 2 * (n // 4) * n * (n // 2 + 1)
        C.c_int.in_dll(lib, "ThisTask").value = 0
        par.use_transposed_fft = 0
        ntasks.value = 3
        assert lib.set_one_grid(0) == 1
        assert "ERROR on task 0: NTasks=3 must divide GridSize=64" in c_output(capfd)
        ntasks.value = 1
        grid.GSglobal[1] = 32
        assert lib.set_one_grid(0) == 1
        assert "ERROR on task 0" in c_output(capfd)
    finally:
        par.use_transposed_fft = 0
        ntasks.value = 1
        C.c_int.in_dll(lib, "ThisTask").value = 0
        grid.GSglobal[1] = n
    assert lib.set_one_grid(0) == 0


def test_dump_and_read_products_round_trip_and_formats(lib, tmp_path, capfd):
    """dump_products / read_dumps (src/fmax.c:372-506) need no device: byte formats of the three files, the round trip, and
    the reference's messages when the dump belongs to another run"""
    n = 8
    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = 16.0
    assert lib.set_one_grid(0) == 0
    par = Params.in_dll(lib, "params")
    par.DumpDir = (str(tmp_path) + "/dump/").encode()
    par.RandomSeed = 486604
    for i in range(3):
        par.GridSize[i] = n
    rng = np.random.default_rng(1)
    raw = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = raw[(-raw.ctypes.data) % 32:][:n ** 3 * 56]
    prod[:] = rng.integers(0, 256, prod.size, dtype=np.uint8)
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.array([0.25, 1.5, 6.25])
    sm.Nsmooth = 3
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.dump_products() == 0
    d = tmp_path / "dump"
    assert (d / "summary").read_text() == "1   # NTasks\n486604   # random seed\n8   # grid size\n56   # length of product_data\n"
    assert (d / "TrueVariance").read_bytes() == tv.tobytes()
    assert (d / "Task.0").read_bytes() == prod.tobytes()
    want = prod.copy()
    prod[:] = 0
    tv[:] = 0
    assert lib.read_dumps() == 0
    assert np.array_equal(prod, want) and list(tv) == [0.25, 1.5, 6.25]
    c_output(capfd)
    par.RandomSeed = 7
    assert lib.read_dumps() == 1
    out = c_output(capfd)
    assert "ERROR: the random seed in %ssummary does not match - 486604 vs 7" % par.DumpDir.decode() in out
    par.RandomSeed = 486604
    # Fmax_PDF from host products only (no device context): the reference's own loop and file
    rec = np.frombuffer(prod, dtype=np.dtype([("Rmax", "<i4"), ("Fmax", "<f4"), ("rest", "V48")]))
    fm = rng.uniform(-11.0, 25.0, n ** 3).astype(np.float32)
    np.frombuffer(prod, dtype=np.uint8).reshape(-1, 56)[:, 4:8] = fm.view(np.uint8).reshape(-1, 4)
    par.RunFlag = b"hostpdf"
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.Fmax_PDF() == 0
        lines = open("pinocchio.hostpdf.FmaxPDF.out").read().splitlines()
    finally:
        os.chdir(cwd)
    assert lines[0] == "# Fmax PDF over %d particles" % n ** 3 and lines[1:4] == ["# 1-2: F interval", "# 3: number of particles in that interval", "#"]
    bins = np.clip((fm.astype(np.float32) * np.float64(10.0)).astype(np.int64), 0, 209)  # (int)(float * 10.) in double, truncation
    counts = np.bincount(bins, minlength=210)
    assert len(lines) == 4 + 210
    for b in (0, 1, 57, 208, 209):
        assert lines[4 + b] == " %6.1f   %6.1f  %d" % (b / 10.0, 999.0 if b == 209 else (b + 1) / 10.0, counts[b])
    assert rec.shape == (n ** 3,)


LPT3 = ["-DTWO_LPT", "-DTHREE_LPT"]          # the reference Makefile's defaults (src/Makefile:46-47)


@pytest.mark.parametrize("flags", [LPT3, LPT3 + ["-DSCALE_DEPENDENT"], LPT3 + ["-DTABULATED_CT"], LPT3 + ["-DELL_SNG"],
                                   LPT3 + ["-DTABULATED_CT", "-DELL_SNG"], LPT3 + ["-DTABULATED_CT", "-DTRILINEAR"],
                                   LPT3 + ["-DTABULATED_CT", "-DALL_SPLINE"],
                                   LPT3 + ["-DTABULATED_CT", "-DELL_SNG", "-DMOD_GRAV_FR", "-DFR0=1e-5"],
                                   LPT3 + ["-DRECOMPUTE_DISPLACEMENTS", "-DSCALE_DEPENDENT"],
                                   ["-DTWO_LPT"], ["-DTWO_LPT", "-DRECOMPUTE_DISPLACEMENTS"], [], ["-DTHREE_LPT"],
                                   LPT3 + ["-DDOUBLE_PRECISION_PRODUCTS", "-DRECOMPUTE_DISPLACEMENTS"]])
def test_in_tree_build_of_the_adapter_type_checks(flags):
    """INTEGRATION.md's recipe compiles pf_compat.c with -DPF_IN_PINOCCHIO_TREE against the reference's pinocchio.h; MPI, GSL
    and PFFT are not in this image, so the #ifdef branches are type-checked (-fsyntax-only) against declaration-only
    headers (tests/intree_decls/: test infrastructure, no definitions, pins nothing)"""
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-DPF_IN_PINOCCHIO_TREE", *flags,
           "-I" + os.path.join(ROOT, "tests", "intree_decls"), os.path.join(ROOT, "pinocchio_amd", "host", "pf_compat.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
def test_compute_fmax_like_the_reference_driver(lib, tmp_path):
    n = 32
    cell = 2.0                                     # Mpc per cell
    dk = np.ascontiguousarray(synth.make_density(n, seed=synth.SEED))
    radii_mpc = np.array([4.0, 2.0, 1.0, 0.0])     # Smoothing.Radius in Mpc; Rsmooth = R / CellSize
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()

    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = n * cell
    assert lib.set_one_grid(0) == 0

    # sizeof(product_data) = 56: the aligned(32) attribute sits on the typedef, it does not pad the record
    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = dk.ctypes.data
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.zeros(len(radii_mpc)); var = np.ones(len(radii_mpc))
    sm.Nsmooth = len(radii_mpc)
    sm.Radius = radii_mpc.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.nseg = 1
    sd.z[0] = 0.0
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v)) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"pftest"
    par.DumpDir = (str(tmp_path) + "/").encode()
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    par.RandomSeed = synth.SEED

    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0
        assert lib.compute_fmax() == 0
        assert lib.dump_products() == 0
    finally:
        os.chdir(cwd)

    p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)

    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii_mpc / cell, do_lpt=True)
    po = o.products()
    assert np.allclose(tv, tv_o, rtol=1e-12)
    ulp = np.spacing(np.maximum(np.abs(po["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(p["Fmax"].astype(np.float64) - po["Fmax"]) <= 2 * ulp)
    assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * np.max(np.abs(po[name]))
    # the FmaxPDF file has the reference's format (src/fmax.c:537-546)
    lines = open(tmp_path / "pinocchio.pftest.FmaxPDF.out").read().splitlines()
    assert lines[0] == "# Fmax PDF over %d particles" % n ** 3 and len(lines) == 4 + 210
    counts = np.array([int(l.split()[2]) for l in lines[4:]])
    assert counts.sum() == n ** 3 and np.abs(counts - o.fmax_pdf().astype(np.int64)).sum() <= 2
    # dump files (src/fmax.c:372-426) and their reload (:429-506)
    assert open(tmp_path / "summary").read().split("\n")[3].startswith("56 ")
    dumped = np.fromfile(tmp_path / "Task.0", dtype=np.uint8)
    assert dumped.size == n ** 3 * 56 and np.array_equal(dumped, prod)
    prod[:] = 0
    tv[:] = 0
    assert lib.read_dumps() == 0
    assert np.array_equal(dumped, prod) and np.allclose(tv, tv_o, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("transposed", [0, 1])
def test_lpt_snapshot_mode_goes_straight_to_the_displacements(lib, tmp_path, transposed):
    """"pinocchio.x parameterfile 3" (src/pinocchio.c:172-200, the N-body initial-condition mode of
    tests/Readme_Pinocchio_tests_V5_1.txt): after the initialisation (plans made) the driver calls
    compute_displacements(1, 1, outputs.z[0]) -- second derivatives at R = 0 recomputed, sources, twelve displacement fields --
    and hands the host `products` to write_LPT_snapshot; compute_fmax never runs.  Against the oracle's
    compute_second_derivatives(0) + compute_LPT_displacements + first derivatives.  transposed: UseTransposedFFT 1, the
    setting for initial conditions that reproduce an N-GenIC / 2LPTic run (DOCUMENTATION:377) -- kdensity is stored [y, x, z]."""
    n = 32
    dk = np.ascontiguousarray(synth.make_density(n, seed=77))
    kdens = np.ascontiguousarray(dk.transpose(1, 0, 2)) if transposed else dk
    g = np.array([0.61, -0.21, 0.017, -0.043])      # growth multipliers at the snapshot's redshift
    x, y = synth.invgrow_table("lcdm")
    assert lib.finalize_fft() == 0                   # whatever an earlier test left behind
    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = n * 2.0
    par = Params.in_dll(lib, "params")
    par.use_transposed_fft = transposed
    assert lib.set_one_grid(0) == 0
    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = kdens.ctypes.data
    sm = Smoothing.in_dll(lib, "Smoothing")
    radii = np.array([2.0, 0.0]); tv = np.zeros(2); var = np.ones(2)
    sm.Nsmooth = 2
    sm.Radius = radii.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v) if z == 49.0 else float("nan")) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"pfsnap"
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    C.c_int.in_dll(lib, "pf_compat_scale_dependent").value = 0
    C.c_int.in_dll(lib, "pf_compat_tabulated_ct").value = 0
    C.c_int.in_dll(lib, "pf_compat_ell_sng").value = 0
    lib.compute_displacements.argtypes = [C.c_int, C.c_int, C.c_double]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0          # initialization() -> set_grids -> compute_fft_plans
        assert lib.compute_displacements(1, 1, 49.0) == 0
        assert lib.finalize_fft() == 0
    finally:
        os.chdir(cwd)
        par.use_transposed_fft = 0
    p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    o.second_derivatives(0.0)
    o.displacements(compute_sources=True)
    po = o.products()
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        amp = np.max(np.abs(po[name]))
        assert amp > 0 and np.max(np.abs(p[name].astype(np.float64) - po[name])) <= 4e-7 * amp, name


@pytest.mark.gpu
def test_genic_on_device_through_the_reference_driver(lib, tmp_path):
    """GenIC_large replaced by pf_compat_genic: no kdensity[0] on the host at all; same sigma and Fmax as the oracle
    fed with the restated generator's field"""
    import json
    import ic_oracle
    with open(os.path.join(ROOT, "tests", "golden", "hmf_validation_kat.json")) as fh:
        kp = json.load(fh)["params"]
    n, seed, pkn = 32, 486604, 2.0e7
    box = n * 2.0 / kp["Hubble100"]                # true Mpc
    cell = box / n
    radii_mpc = np.array([3.0 * cell, 1.5 * cell, 0.0])
    x, y = ic_oracle.growth_table_lcdm(kp["Omega0"])
    g = synth.growth_multipliers()

    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = box
    assert lib.set_one_grid(0) == 0
    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = None                                   # would fault if the adapter touched it
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.zeros(len(radii_mpc)); var = np.ones(len(radii_mpc))
    sm.Nsmooth = len(radii_mpc)
    sm.Radius = radii_mpc.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.nseg = 1
    sd.z[0] = 0.0
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v)) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"pfgenic"
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    par.RandomSeed = seed
    par.Omega0, par.OmegaBaryon, par.Hubble100 = kp["Omega0"], kp["OmegaBaryon"], kp["Hubble100"]
    par.Sigma8, par.PrimordialIndex, par.BoxSize_htrue = kp["Sigma8"], kp["PrimordialIndex"], box

    lib.pf_compat_genic.argtypes = [C.c_double]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0
        assert lib.pf_compat_genic(pkn) == 0
        assert lib.compute_fmax() == 0
    finally:
        os.chdir(cwd)
    p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)

    dk = ic_oracle.genic(n, box, seed, pkn, kp)
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    tv_o = o.compute_fmax(radii_mpc / cell, do_lpt=True)
    po = o.products()
    assert tv_o[-1] > 0 and np.allclose(tv, tv_o, rtol=1e-11)
    ulp = np.spacing(np.maximum(np.abs(po["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.mean(np.abs(p["Fmax"].astype(np.float64) - po["Fmax"]) > 2 * ulp) < 1e-4
    assert np.max(np.abs(p["Vel"].astype(np.float64) - po["Vel"])) <= 4e-7 * np.max(np.abs(po["Vel"]))


@pytest.mark.gpu
def test_reentrant_displacements_recompute_build(tmp_path):
    """The adapter built with the reference's default -DRECOMPUTE_DISPLACEMENTS (104-byte product_data, context kept
    after compute_fmax): fragment's sequence shift_all_displacements(); compute_displacements(0, 0, z)
    (src/fragment.c:398-410) rewrites the Vel* fields of the host records and nothing else; with
    pf_compat_scale_dependent the growth functions are sampled into the NkBINS tables (src/cosmo.c:1728-1755)."""
    import __graft_entry__ as g_entry
    so = os.path.join(ROOT, "pinocchio_amd", "libpf_compat_recompute.so")
    if not os.path.exists(so):
        g_entry.build()
    lib = C.CDLL(so)
    rec = np.dtype([("Rmax", "<i4"), ("Fmax", "<f4"), ("Vel", "<f4", 3), ("Vel_2LPT", "<f4", 3), ("Vel_3LPT_1", "<f4", 3),
                    ("Vel_3LPT_2", "<f4", 3), ("Vel_prev", "<f4", 3), ("Vel_2LPT_prev", "<f4", 3),
                    ("Vel_3LPT_1_prev", "<f4", 3), ("Vel_3LPT_2_prev", "<f4", 3)])
    n, cell = 32, 2.0
    dk = np.ascontiguousarray(synth.make_density(n, seed=31))
    radii_mpc = np.array([4.0, 2.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g0 = synth.growth_multipliers()
    scale = {0.0: np.ones(4), 1.0: np.array([0.6, 0.36, 0.216, 0.216])}     # growth at the two "redshifts"

    def growth(o, z, k):    # mildly k-dependent, like an f(R) table; k in rad/cell (quirk Q3)
        return float(g0[o] * scale[z][o] * (1.0 + 0.03 * (o + 1) * np.log10(max(k, 1e-3) / 1e-3)))

    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = n * cell
    assert lib.set_one_grid(0) == 0
    raw = np.zeros(n ** 3 * 104 + 64, dtype=np.uint8)
    raw = raw[(-raw.ctypes.data) % 32:][:n ** 3 * 104]
    C.c_void_p.in_dll(lib, "products").value = raw.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = dk.ctypes.data
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.zeros(len(radii_mpc)); var = np.ones(len(radii_mpc))
    sm.Nsmooth = len(radii_mpc)
    sm.Radius = radii_mpc.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.nseg = 2
    sd.z[0], sd.z[1] = 0.0, 1.0
    C.c_int.in_dll(lib, "pf_compat_scale_dependent").value = 1
    knots = (Knots * 64).in_dll(lib, "pf_invgrow_knots_radius")
    for i in range(len(radii_mpc)):
        knots[i].size = len(x)
        knots[i].x = x.ctypes.data_as(C.POINTER(C.c_double))
        knots[i].y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, o=o: growth(o, z, k)) for o in range(4)]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"pfrecomp"
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    par.RandomSeed = 31

    lib.compute_displacements.argtypes = [C.c_int, C.c_int, C.c_double]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0
        assert lib.compute_fmax() == 0
        recs = raw.view(rec).reshape(n, n, n)
        first = recs.copy()
        for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):       # shift_all_displacements
            recs[name + "_prev"] = recs[name]
        assert lib.compute_displacements(0, 0, 1.0) == 0
    finally:
        os.chdir(cwd)

    kbin = 10.0 ** (-3.0 + 0.5 * np.arange(10))
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y)
    want = {}
    for z in (0.0, 1.0):
        for k in range(4):
            t = np.array([growth(k, z, kk) for kk in kbin])
            o.set_growth_table(k + 1, np.log10(np.abs(t)), sign=float(np.sign(t[0])))
        if z == 0.0:
            o.compute_fmax(radii_mpc / cell, do_lpt=True)
        else:
            o.displacements(compute_sources=False)
        want[z] = o.products()
    for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        amp = np.max(np.abs(want[0.0][name]))
        assert np.max(np.abs(first[name].astype(np.float64) - want[0.0][name])) <= 4e-7 * amp, name
        assert np.max(np.abs(recs[name].astype(np.float64) - want[1.0][name])) <= 4e-7 * amp, name
        assert np.array_equal(recs[name + "_prev"], first[name])
        assert np.max(np.abs(recs[name] - first[name])) > 0.1 * amp
    assert np.array_equal(recs["Fmax"], first["Fmax"]) and np.array_equal(recs["Rmax"], first["Rmax"])
    ulp = np.spacing(np.maximum(np.abs(want[0.0]["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
    assert np.mean(np.abs(first["Fmax"].astype(np.float64) - want[0.0]["Fmax"]) > 2 * ulp) < 1e-4


@pytest.mark.gpu
def test_fft_module_seam_like_the_density_writer(lib):
    """src/pinocchio.c:146-150: write_in_cvector(kdensity); reverse_transform; write_from_rvector(density) -- and the
    per-component drivers compute_first_derivatives / compute_second_derivatives (src/fmax.c:193-258) on host arrays"""
    import np_restatement as npr
    n, cell = 32, 2.0
    dk = np.ascontiguousarray(synth.make_density(n, seed=13))
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = n * cell
    assert lib.set_one_grid(0) == 0
    cvec = np.zeros((n, n, n // 2 + 1), dtype=np.complex128)
    rvec = np.zeros(2 * n * n * (n // 2 + 1))          # the reference sizes rvector_fft like the complex array
    C.cast(C.c_void_p.in_dll(lib, "cvector_fft"), C.POINTER(C.c_void_p))[0] = cvec.ctypes.data
    C.cast(C.c_void_p.in_dll(lib, "rvector_fft"), C.POINTER(C.c_void_p))[0] = rvec.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = dk.ctypes.data
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v)) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    dp = C.POINTER(C.c_double)
    lib.write_in_cvector.argtypes = [C.c_int, dp]
    lib.write_from_cvector.argtypes = [C.c_int, dp]
    lib.write_in_rvector.argtypes = [C.c_int, dp]
    lib.write_from_rvector.argtypes = [C.c_int, dp]
    lib.reverse_transform.restype = C.c_double
    lib.forward_transform.restype = C.c_double
    lib.compute_first_derivatives.argtypes = [C.c_double, C.c_int, C.c_int, dp]
    lib.compute_second_derivatives.argtypes = [C.c_double, C.c_int]

    assert lib.compute_fft_plans() == 0
    density = np.zeros((n, n, n))
    lib.write_in_cvector(0, dk.view(np.float64).ctypes.data_as(dp))
    assert lib.reverse_transform(0) >= 0.0
    lib.write_from_rvector(0, density.ctypes.data_as(dp))
    want = np.fft.irfftn(dk, s=(n, n, n), axes=(0, 1, 2))
    assert np.max(np.abs(density - want)) <= 1e-13 * np.max(np.abs(want))
    # and back: ReadWhiteNoise.c:161-222 (forward_transform, write_from_cvector)
    lib.write_in_rvector(0, want.ctypes.data_as(dp))
    assert lib.forward_transform(0) >= 0.0
    spec = np.zeros_like(dk)
    lib.write_from_cvector(0, spec.view(np.float64).ctypes.data_as(dp))
    ref = np.fft.rfftn(want, axes=(0, 1, 2))
    assert np.max(np.abs(spec - ref)) <= 1e-12 * np.max(np.abs(ref))

    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.order, sd.redshift = 1, 0.0
    assert lib.compute_first_derivatives(0.0, 0, 1, dk.view(np.float64).ctypes.data_as(dp)) == 0
    p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)
    for ia in (1, 2, 3):
        w = npr.derivative(dk, 0.0, ia, 0, g[0])
        assert np.max(np.abs(p["Vel"][..., ia - 1] - w)) <= 2e-7 * np.max(np.abs(w))
    assert not p["Vel_2LPT"].any()

    hes = [np.zeros((n, n, n)) for _ in range(6)]
    ptrs = (dp * 6)(*[h.ctypes.data_as(dp) for h in hes])
    pp = (C.POINTER(dp) * 1)(C.cast(ptrs, C.POINTER(dp)))
    C.c_void_p.in_dll(lib, "second_derivatives").value = C.cast(pp, C.c_void_p).value
    sd.order = 0
    assert lib.compute_second_derivatives(3.0, 0) == 0     # R in Mpc: Rsmooth = R / CellSize = 1.5 cells
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk)
    ho = o.second_derivatives(1.5)
    for i in range(6):
        assert np.max(np.abs(hes[i] - ho[i])) <= 1e-12 * np.max(np.abs(ho[i]))
    assert lib.finalize_fft() == 0


@pytest.mark.gpu
def test_tabulated_ct_build_through_the_reference_driver(lib, tmp_path):
    """-DTABULATED_CT flow of compute_fmax (src/fmax.c:66-150): per radius second derivatives, initialize_collapse_times
    (table on the device, written to pinocchio.<run>.CTtable.out in the reference's binary format), collapse times by
    interpolation; then the same run reading the table back through params.CTtableFile"""
    n, cell = 32, 2.0
    dk = np.ascontiguousarray(synth.make_density(n, seed=23))
    radii_mpc = np.array([4.0, 2.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 0)
    o.set_density(dk); o.set_invgrow(x, y); o.set_growth(g)
    var = o.compute_fmax(radii_mpc / cell, do_lpt=False) * 1.05      # stands in for the expected variance
    o.set_tabulated_ct(var)
    tv_o = o.compute_fmax(radii_mpc / cell, do_lpt=True)
    po = o.products()

    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = n * cell
    assert lib.set_one_grid(0) == 0
    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    kd = C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))
    kd[0] = dk.ctypes.data
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.zeros(len(radii_mpc))
    sm.Nsmooth = len(radii_mpc)
    sm.Radius = radii_mpc.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.nseg = 1
    sd.z[0] = 0.0
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v)) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"pfct"
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    par.Omega0, par.OmegaLambda, par.Hubble100 = 0.25, 0.75, 0.7
    par.CTtableFile = b"none"
    C.c_int.in_dll(lib, "pf_compat_scale_dependent").value = 0
    C.c_int.in_dll(lib, "pf_compat_tabulated_ct").value = 1

    def check():
        p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)
        assert np.allclose(tv, tv_o, rtol=1e-12)
        ulp = np.spacing(np.maximum(np.abs(po["Fmax"]), 1.0).astype(np.float32)).astype(np.float64)
        assert np.mean(np.abs(p["Fmax"].astype(np.float64) - po["Fmax"]) > 2 * ulp) < 1e-4
        assert np.mean(p["Rmax"] != po["Rmax"]) < 1e-3
        assert np.max(np.abs(p["Vel_2LPT"].astype(np.float64) - po["Vel_2LPT"])) <= 4e-7 * np.max(np.abs(po["Vel_2LPT"]))
        return p["Fmax"].copy()

    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0
        assert lib.compute_fmax() == 0
        first = check()
        fname = tmp_path / "pinocchio.pfct.CTtable.out"
        assert fname.stat().st_size == 4 + 3 * 8 + 3 * 4 + len(radii_mpc) * (4 + 8 * 250000)
        raw = fname.read_bytes()
        assert np.frombuffer(raw[:4], dtype=np.int32)[0] == 1 and np.frombuffer(raw[28:40], dtype=np.int32).tolist() == [250000, 100, 50]
        tab1 = np.frombuffer(raw[40 + 4 + 2000000 + 4:40 + 2 * (4 + 2000000)], dtype=np.float64).reshape(50, 50, 100)
        t_o, _ = o.ct_build(1, var[1])
        nz = (tab1 != 0) & (t_o != 0)
        assert np.mean(np.abs(tab1[nz] - t_o[nz]) > 1e-11 * np.maximum(1.0, t_o[nz])) < 3e-3
        # second run: the table comes from the file
        par.CTtableFile = str(fname).encode()
        prod[:] = 0
        tv[:] = 0
        assert lib.compute_fft_plans() == 0
        assert lib.compute_fmax() == 0
        assert np.array_equal(check(), first)
        # "pinocchio.x parameterfile 1" (src/pinocchio.c:100-131): straight from the initialisation, only the tables of all
        # radii, written to params.CTtableFile itself
        only = tmp_path / "only.CTtable"
        par.CTtableFile = str(only).encode()
        lib.initialize_collapse_times.argtypes = [C.c_int, C.c_int]
        assert lib.finalize_fft() == 0 and lib.compute_fft_plans() == 0
        for i in range(len(radii_mpc)):
            assert lib.initialize_collapse_times(i, 1) == 0
        assert lib.finalize_fft() == 0
        assert only.read_bytes() == raw
        # -DELL_SNG: the same flow with the table filled by the ODE model (type code 3 in the file header)
        cosmo = np.array([0.25, 0.75, 0.0, 0.0])
        d_in = np.array([float(g[0]) * 1.28e-5] * len(radii_mpc))
        o.set_collapse_model(1, cosmo, d_in)
        tv_o[:] = o.compute_fmax(radii_mpc / cell, do_lpt=True)
        po = o.products()
        hub = HUBBLE_FN(lambda z: 70.0 * float(np.sqrt(0.25 * (1 + z) ** 3 + 0.75)))
        C.c_void_p.in_dll(lib, "pf_Hubble").value = C.cast(hub, C.c_void_p).value
        gm = GROWTH_FN(lambda z, k: float(g[0]) * (1.28e-5 if z > 1000.0 else 1.0))   # GrowingMode(z(a = 1e-5), k) and at z = 0
        C.c_void_p.in_dll(lib, "pf_GrowingMode").value = C.cast(gm, C.c_void_p).value
        C.c_int.in_dll(lib, "pf_compat_ell_sng").value = 1
        par.CTtableFile = b"none"
        par.RunFlag = b"pfsng"
        prod[:] = 0
        tv[:] = 0
        assert lib.compute_fft_plans() == 0
        assert lib.compute_fmax() == 0
        p = prod.view(oracle_lib.PRODUCT_DTYPE).reshape(n, n, n)
        assert np.allclose(tv, tv_o, rtol=1e-12)
        d = np.abs(p["Fmax"].astype(np.float64) - po["Fmax"])
        assert np.mean(d > 1e-4 * np.maximum(1.0, po["Fmax"])) < 1e-3 and (po["Fmax"] >= 1.0).mean() > 0.05
        assert np.frombuffer((tmp_path / "pinocchio.pfsng.CTtable.out").read_bytes()[:4], dtype=np.int32)[0] == 3
    finally:
        os.chdir(cwd)
        C.c_int.in_dll(lib, "pf_compat_tabulated_ct").value = 0
        C.c_int.in_dll(lib, "pf_compat_ell_sng").value = 0
        par.CTtableFile = b"none"


@pytest.mark.gpu
def test_reference_validation_run_through_the_adapter(lib, tmp_path):
    """The reference's committed validation run (HMF_Validation/, tests/golden/hmf_validation_kat.json) driven the way
    main() drives it, through the reference-named entry points only: initial conditions from seed + cosmology
    (pf_compat_genic in place of GenIC_large), compute_fmax, and the pinocchio.<run>.FmaxPDF.out it writes -- compared
    with the logged sigma per radius, the collapsed count and the committed histogram file"""
    import json
    import ic_oracle
    with open(os.path.join(ROOT, "tests", "golden", "hmf_validation_kat.json")) as fh:
        kat = json.load(fh)
    kp = kat["params"]
    n = kp["GridSize"]
    box = kp["BoxSize_h100"] / kp["Hubble100"]
    radii_mpc = np.array(kat["radii_Mpc"])
    x, y = ic_oracle.growth_table_lcdm(kp["Omega0"])
    g = synth.growth_multipliers()

    grid = C.cast(C.c_void_p.in_dll(lib, "MyGrids"), C.POINTER(Grid)).contents
    for i in range(3):
        grid.GSglobal[i] = n
    grid.Ntotal = n ** 3
    grid.BoxSize = box
    assert lib.set_one_grid(0) == 0
    prod = np.zeros(n ** 3 * 56 + 64, dtype=np.uint8)
    prod = prod[(-prod.ctypes.data) % 32:][:n ** 3 * 56]
    C.c_void_p.in_dll(lib, "products").value = prod.ctypes.data
    C.cast(C.c_void_p.in_dll(lib, "kdensity"), C.POINTER(C.c_void_p))[0] = None
    sm = Smoothing.in_dll(lib, "Smoothing")
    tv = np.zeros(len(radii_mpc)); var = np.array(kat["variance"])
    sm.Nsmooth = len(radii_mpc)
    sm.Radius = radii_mpc.ctypes.data_as(C.POINTER(C.c_double))
    sm.Variance = var.ctypes.data_as(C.POINTER(C.c_double))
    sm.TrueVariance = tv.ctypes.data_as(C.POINTER(C.c_double))
    sd = ScaleDep.in_dll(lib, "ScaleDep")
    sd.nseg = 1
    sd.z[0] = 0.0
    kn = Knots.in_dll(lib, "pf_invgrow_knots")
    kn.size = len(x)
    kn.x = x.ctypes.data_as(C.POINTER(C.c_double))
    kn.y = y.ctypes.data_as(C.POINTER(C.c_double))
    fns = [GROWTH_FN(lambda z, k, v=v: float(v)) for v in g]
    for name, fn in zip(("pf_GrowingMode", "pf_GrowingMode_2LPT", "pf_GrowingMode_3LPT_1", "pf_GrowingMode_3LPT_2"), fns):
        C.c_void_p.in_dll(lib, name).value = C.cast(fn, C.c_void_p).value
    par = Params.in_dll(lib, "params")
    par.RunFlag = b"test"
    par.GridSize[0] = par.GridSize[1] = par.GridSize[2] = n
    par.RandomSeed = kp["RandomSeed"]
    par.Omega0, par.OmegaBaryon, par.Hubble100 = kp["Omega0"], kp["OmegaBaryon"], kp["Hubble100"]
    par.Sigma8, par.PrimordialIndex, par.BoxSize_htrue = kp["Sigma8"], kp["PrimordialIndex"], box
    par.CTtableFile = b"none"
    for flag in ("pf_compat_scale_dependent", "pf_compat_tabulated_ct", "pf_compat_ell_sng"):
        C.c_int.in_dll(lib, flag).value = 0

    lib.pf_compat_genic.argtypes = [C.c_double]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.compute_fft_plans() == 0
        assert lib.pf_compat_genic(kat["PkNorm"]) == 0
        assert lib.compute_fmax() == 0
        written = [int(l.split()[2]) for l in open("pinocchio.test.FmaxPDF.out") if not l.startswith("#")]
    finally:
        os.chdir(cwd)
    assert np.all(np.abs(np.sqrt(tv) - np.array(kat["computed_sigma"])) <= 6e-5)        # the log prints 4 decimals
    want = np.array(kat["FmaxPDF"], dtype=np.int64)
    got = np.array(written, dtype=np.int64)
    assert got.sum() == n ** 3 and abs(int(got[10:].sum()) - kat["collapsed"]) <= 5
    assert np.abs(got - want).sum() <= 200
