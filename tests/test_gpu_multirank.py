"""The slab-decomposed path (what an 8-GPU run executes) on ONE GPU: P contexts
(ranks) driven by P host threads meet in the exchange through the library's
in-process fabric (device-to-device copies instead of RCCL).  Results must not
depend on the decomposition: the per-line transforms are the same operations
whatever P is, only the small reductions change their summation order."""
import os
import threading

import numpy as np
import pytest

from pinocchio_amd import synth

pytestmark = pytest.mark.gpu


def run_ranks_timed(api, n, P, body, field_bytes=8):
    return run_ranks(api, n, P, body, field_bytes, timing=True)


def run_ranks(api, n, P, body, field_bytes=8, timing=False, delay_us=0):
    from pinocchio_amd import _lib
    L = _lib.load()
    fab = L.pf_fabric_create(P)
    assert fab
    if delay_us:
        assert L.pf_fabric_set_delay(fab, delay_us) == 0
    ctxs = [api.Fmax(n, rank=r, nranks=P, field_bytes=field_bytes, timing=timing) for r in range(P)]
    for c in ctxs:
        assert L.pf_fabric_attach(fab, c.h) == 0
    out, err = [None] * P, [None] * P

    def work(r):
        try:
            out[r] = body(ctxs[r], r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    alive = [t.is_alive() for t in th]
    for c in ctxs:
        if not any(alive):
            c.close()
    if not any(alive):
        L.pf_fabric_destroy(fab)
    assert not any(alive), "rank threads hung"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.fixture(scope="module")
def api():
    from pinocchio_amd import api as _api
    return _api


@pytest.mark.parametrize("replicate", [1, 0])
# (48, 96, 120: grid sizes that are not a power of two -- slabs of 24, 24 and 15 planes, split by multiply-high in the mixed-radix passes)
@pytest.mark.parametrize("n,P,pipeline", [(32, 2, 1), (64, 4, 1), (64, 8, 1), (64, 4, 0), (16, 16, 1), (32, 16, 0), (256, 8, 1),
                                          (48, 2, 1), (96, 4, 1), (120, 8, 0), (200, 8, 1),
                                          (96, 3, 1), (120, 6, 1), (200, 5, 0), (240, 12, 1)])   # (rank counts that are not a power of two: mixed-radix grids take any divisor)
def test_slab_ranks_match_single_rank(api, n, P, pipeline, replicate, monkeypatch):
    # replicate = 1 (default): every rank holds the whole delta(k) and the passes that start from it exchange nothing;
    # 0: every transform goes through the all-to-all
    monkeypatch.setenv("PF_REPLICATE_DK", str(replicate))
    # pipeline = 1 (default): double-buffered exchange on the communication stream, the all-to-all of transform i+1
    # issued before the y/z passes of transform i; 0: one buffer set, everything on one stream
    monkeypatch.setenv("PF_PIPELINE", str(pipeline))
    dk = synth.make_density(n, seed=17 + P)
    dk[0, 0, 0] = 0.21 * n ** 3  # DC mode lives on rank 0 and must reach every rank
    radii = np.array([4.0, 2.0, 1.4, 1.0, 0.0])       # odd count: both buffer sets end up last
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    nxl = n // P

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y)
        f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        h = None
        f.compute_second_derivatives(1.0)
        h = [f.second_derivative(i) for i in range(6)]
        return tv, f.products(), f.Fmax_PDF(), h

    with api.Fmax(n) as f1:
        f1.set_density(dk)
        f1.set_invgrow(x, y)
        f1.set_growth(g)
        tv1 = f1.compute_fmax(radii, do_lpt=True)
        p1 = f1.products()
        pdf1 = f1.Fmax_PDF()
        f1.compute_second_derivatives(1.0)
        h1 = [f1.second_derivative(i) for i in range(6)]

    res = run_ranks(api, n, P, body)
    for r in range(P):
        tv, p, pdf, h = res[r]
        sl = slice(r * nxl, (r + 1) * nxl)
        assert np.allclose(tv, tv1, rtol=1e-13)            # all-reduced: same on every rank
        assert np.array_equal(pdf, pdf1)                   # histogram summed over ranks
        for i in range(6):
            assert np.array_equal(h[i], h1[i][sl]), (r, i)  # identical per-line transforms
        assert np.array_equal(p["Fmax"], p1["Fmax"][sl])
        assert np.array_equal(p["Rmax"], p1["Rmax"][sl])
        for name in ("Vel", "Vel_2LPT", "Vel_3LPT_1"):
            assert np.array_equal(p[name], p1[name][sl]), name
        # the 3LPT(b) source carries the all-reduced mean of the 2LPT source: ulp-level differences allowed
        a, b = p["Vel_3LPT_2"].astype(np.float64), p1["Vel_3LPT_2"][sl].astype(np.float64)
        assert np.max(np.abs(a - b)) <= 2e-7 * np.max(np.abs(b))


@pytest.mark.parametrize("fb", [8, 4])
def test_replicated_spectrum_leaves_only_the_lpt_transposes(api, fb, monkeypatch):
    """PF_REPLICATE_DK (default on with more than one rank): every rank gathers the whole delta(k) once (an integer all-reduce
    of a zero-padded array: exact for fp64 and fp32 fields) and the second derivatives of every radius and the Zel'dovich
    displacements start from it -- each rank transforms every x-line and keeps its own slab.  What still travels: the three
    forward transforms of the LPT sources, the Hessian of the 2LPT potential (3 fields) and three displacement pairs:
    12 of the 50 field exchanges of a 12-radius step (here 12 of 5 * 3 + 2 + 12 = 29).  Same results bit for bit."""
    n, P = 64, 4
    nxl = n // P
    dk = synth.make_density(n, seed=31)
    radii = np.array([8.0, 4.0, 2.0, 1.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g)
        f.compute_fmax(radii, do_lpt=True)           # includes the one-off gather
        f.reset_kernel_stats()
        tv = f.compute_fmax(radii, do_lpt=True)
        ex = [k for k in f.kernel_stats() if k["name"] == "exchange"]
        return tv, f.products(), (ex[0]["launches"] if ex else 0)

    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PF_REPLICATE_DK", mode)
        monkeypatch.setenv("PF_EXCHANGE_ROWS", "0")
        out[mode] = run_ranks(api, n, P, body, field_bytes=fb, timing=True)
    for r in range(P):
        assert np.array_equal(out["1"][r][0], out["0"][r][0])
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.array_equal(out["1"][r][1][name], out["0"][r][1][name]), (r, name)
        assert out["0"][r][2] == 5 * 3 + 2 + 12 and out["1"][r][2] == 12, (out["0"][r][2], out["1"][r][2])


@pytest.mark.parametrize("P", [1, 2, 4])
def test_transposed_spectra_at_the_boundary(api, P):
    """params.use_transposed_fft (PFFT_TRANSPOSED_OUT on slabs, src/fmax-pfft.c:92, 271-281): every spectrum crosses the
    interface as this rank's ky-slab in [ky_local][kx][kz] order.  Same field in -> the very same products, source spectra
    and transforms out (bit for bit: the device layout and every kernel are the same, only the import / export differ)."""
    n = 32
    dk = synth.make_density(n, seed=9)
    dk[0, 0, 0] = 0.3 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([2.0, 0.0])
    nl = n // P
    dkt = np.ascontiguousarray(dk.transpose(1, 0, 2))                   # [ky][kx][kz]
    rng = np.random.default_rng(4)
    real = rng.standard_normal((n, n, n))

    def body(f, r, transposed):
        f.set_transposed_spectra(transposed)
        sl = slice(r * nl, (r + 1) * nl)
        f.set_density(dkt[sl] if transposed else dk[sl])
        f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        kv = f.kvector(1)
        spec = f.forward_transform(real[sl])
        back = f.reverse_transform(dkt[sl] if transposed else dk[sl])
        der = f.compute_derivative(dkt[sl] if transposed else dk[sl], 1, 2, 1.0, 0)
        return tv, f.products(), f.density(), kv, spec, back, der

    def run(transposed):
        if P == 1:
            with api.Fmax(n) as f:
                return [body(f, 0, transposed)]
        return run_ranks(api, n, P, lambda f, r: body(f, r, transposed))

    plain, tr = run(False), run(True)
    for r in range(P):
        assert np.array_equal(plain[r][0], tr[r][0])
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.array_equal(plain[r][1][name], tr[r][1][name]), (r, name)
        assert np.array_equal(plain[r][5], tr[r][5]) and np.array_equal(plain[r][6], tr[r][6])
    # spectra out: the transposed ones are the ky-slabs of the same global arrays
    for k in (2, 3, 4):
        whole = np.concatenate([plain[r][k] for r in range(P)], axis=0)                  # [kx][ky][kz]
        whole_t = np.concatenate([tr[r][k] for r in range(P)], axis=0)                   # [ky][kx][kz]
        assert np.array_equal(whole.transpose(1, 0, 2), whole_t), k
    assert np.array_equal(np.concatenate([tr[r][2] for r in range(P)], axis=0), dkt)


@pytest.mark.parametrize("P", [2, 8])
def test_state_machine_on_slabs(api, P):
    """the sequence test of tests/test_gpu_parity.py with more than one rank (P = 2: replicated delta(k), gathered again after every
    new density; P = 8: exchanging path): every rank drives its long-lived context through the same seeded sequence, and after
    each step holds the slab of what a fresh single-rank context computes"""
    n = 32
    nxl = n // P
    x, y = synth.invgrow_table("lcdm")
    g0 = synth.growth_multipliers()
    rng = np.random.default_rng(5)
    ops = []
    seed, g, radii = 1, g0, np.array([2.0, 1.0, 0.0])
    for _ in range(14):
        op = int(rng.integers(0, 4))
        if op == 0:
            seed += 1
        elif op == 1:
            g = g0 * rng.uniform(0.5, 1.5, 4)
        elif op == 3:
            radii = np.array([float(rng.uniform(1.5, 4.0)), float(rng.uniform(0.5, 1.4)), 0.0])
        ops.append((op, seed, g.copy(), radii.copy()))
    want = []
    for op, seed, g, radii in ops:                                  # single-rank truth for every step that computes
        with api.Fmax(n) as f:
            f.set_density(synth.make_density(n, seed=seed)); f.set_invgrow(x, y); f.set_growth(g)
            tv = f.compute_fmax(radii, do_lpt=True)
            want.append((tv, f.products()))

    def body(f, r):
        f.set_invgrow(x, y)
        f.set_density(synth.make_density(n, seed=1)[r * nxl:(r + 1) * nxl]); f.set_growth(g0)
        got = []
        for op, seed, g, radii in ops:
            if op == 0:
                f.set_density(synth.make_density(n, seed=seed)[r * nxl:(r + 1) * nxl])
            f.set_growth(g)
            if op == 2:                                             # apart, with the Hessian disturbed in between
                tv = f.sweep(radii)
                f.compute_second_derivatives(1.3)
                f.compute_displacements(1, 1)
            else:
                tv = f.compute_fmax(radii, do_lpt=True)
            got.append((tv, f.products()))
        return got

    res = run_ranks(api, n, P, body)
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        for (tv, p), (tv0, p0) in zip(res[r], want):
            assert np.allclose(tv, tv0, rtol=1e-13)
            for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1"):
                assert np.array_equal(p[name], p0[name][sl]), (r, name)
            a, b = p["Vel_3LPT_2"].astype(np.float64), p0["Vel_3LPT_2"][sl].astype(np.float64)
            assert np.max(np.abs(a - b)) <= 2e-7 * np.max(np.abs(b))


@pytest.mark.parametrize("P", [2, 8])
def test_double_precision_products_on_slabs(api, P):
    """PF_FLAG_DOUBLE_PRODUCTS (-DDOUBLE_PRECISION_PRODUCTS) with more than one rank: fp64 Fmax and displacement columns, bit for
    bit those of one rank (P = 2: replicated delta(k); P = 8: every transform through the all-to-all)"""
    n = 32
    dk = synth.make_density(n, seed=14)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([2.0, 1.0, 0.0])
    nxl = n // P

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g)
        f.compute_fmax(radii, do_lpt=True)
        return f.products(), f.Fmax_PDF()

    with api.Fmax(n, double_products=True) as f1:
        f1.set_density(dk); f1.set_invgrow(x, y); f1.set_growth(g)
        f1.compute_fmax(radii, do_lpt=True)
        p1, h1 = f1.products(), f1.Fmax_PDF()
    from pinocchio_amd import _lib
    L = _lib.load()
    fab = L.pf_fabric_create(P)
    ctxs = [api.Fmax(n, rank=r, nranks=P, double_products=True) for r in range(P)]
    for c in ctxs:
        assert L.pf_fabric_attach(fab, c.h) == 0
    out = [None] * P
    th = [threading.Thread(target=lambda r=r: out.__setitem__(r, body(ctxs[r], r))) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for c in ctxs:
        c.close()
    L.pf_fabric_destroy(fab)
    assert p1["Fmax"].dtype == np.float64
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        assert out[r] is not None and np.array_equal(out[r][1], h1)
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1"):
            assert np.array_equal(out[r][0][name], p1[name][sl]), (r, name)
        a, b = out[r][0]["Vel_3LPT_2"], p1["Vel_3LPT_2"][sl]
        assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b))      # carries the all-reduced mean of the 2LPT source


@pytest.mark.parametrize("order", [2, 1])
def test_lower_lpt_orders_on_slabs(api, order):
    """pf_set_lpt_order (a build without -DTHREE_LPT / -DTWO_LPT) with the exchange pipeline: fewer fields go through it"""
    n, P = 32, 2
    dk = synth.make_density(n, seed=6)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = np.array([1.5, 0.0])
    nxl = n // P

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g); f.set_lpt_order(order)
        f.compute_fmax(radii, do_lpt=True)
        return f.products()

    with api.Fmax(n) as f1:
        f1.set_density(dk); f1.set_invgrow(x, y); f1.set_growth(g); f1.set_lpt_order(order)
        f1.compute_fmax(radii, do_lpt=True)
        p1 = f1.products()
    res = run_ranks(api, n, P, body)
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
            assert np.array_equal(res[r][name], p1[name][sl]), (r, name)
    assert np.abs(p1["Vel"]).max() > 0 and (np.abs(p1["Vel_2LPT"]).max() > 0) == (order == 2) and not p1["Vel_3LPT_1"].any()


def test_synth_density_is_decomposition_independent(api):
    n, P = 64, 4
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([2.0, 0.0])

    def body(f, r):
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        tv = f.sweep(radii)
        return tv, f.products()["Fmax"]

    with api.Fmax(n) as f1:
        f1.synth_density(synth.SEED, 2.5, -2.0)
        f1.set_invgrow(x, y)
        tv1 = f1.sweep(radii)
        fm1 = f1.products()["Fmax"]
    res = run_ranks(api, n, P, body)
    nxl = n // P
    for r in range(P):
        assert np.allclose(res[r][0], tv1, rtol=1e-12)
        d = np.abs(res[r][1].astype(np.float64) - fm1[r * nxl:(r + 1) * nxl].astype(np.float64))
        # the normalisation sigma(R=0)=2.5 goes through an all-reduce: fields agree to ~1e-15, Fmax to fp32 ulps
        assert np.mean(d > 0) < 1e-3 and d.max() < 1e-3


@pytest.mark.parametrize("P", [2, 8])
def test_fp32_fields_on_slabs(api, P):
    n = 64
    dk = synth.make_density(n, seed=5)
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([7.0, 4.0, 2.0, 0.0])        # the first two are band-limited: in-band rows and columns only on the wire
    nxl = n // P

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y)
        f.compute_fmax(radii, do_lpt=True)
        return f.products()

    with api.Fmax(n, field_bytes=4) as f1:
        f1.set_density(dk)
        f1.set_invgrow(x, y)
        f1.compute_fmax(radii, do_lpt=True)
        p1 = f1.products()
    res = run_ranks(api, n, P, body, field_bytes=4)
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        assert np.array_equal(res[r]["Fmax"], p1["Fmax"][sl])
        assert np.array_equal(res[r]["Vel"], p1["Vel"][sl])


def test_fabric_exchange_selftest(api):
    def body(f, r):
        return f.L.pf_debug_exchange(f.h, 4096)
    assert run_ranks(api, 64, 4, body) == [0, 0, 0, 0]


_RCCL_CODE = """
import ctypes as C, os, sys
sys.path.insert(0, %r)
from pinocchio_amd import api
for self_send in ("0", "1"):          # the rank's own block: a local copy beside the group (default), or ncclSend / ncclRecv to itself inside it
    os.environ["PF_RCCL_SELF_SEND"] = self_send
    with api.Fmax(64) as f:
        idbuf = (C.c_ubyte * 128)()
        assert f.L.pf_rccl_unique_id(C.cast(idbuf, C.c_void_p)) == 0
        assert f.L.pf_init_rccl(f.h, C.cast(idbuf, C.c_void_p)) == 0
        assert f.L.pf_debug_exchange(f.h, 1 << 20) == 0, self_send
print("RCCL_OK")
"""

_TORCH_CODE = """
import os, socket, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pinocchio_amd import api, dist as pfdist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for kind in ("torch", "rccl"):
    with api.Fmax(64) as f:
        keep = pfdist.install_exchange(f, dist, torch, kind=kind)
        assert f.L.pf_debug_exchange(f.h, 1 << 20) == 0, kind
        del keep
# what bench.py --gpus N does: vote-guarded negotiation on the NCCL (= RCCL) process group, either preference
for pref in ("rccl", "torch"):
    with api.Fmax(64) as f:
        name, keep = pfdist.negotiate_exchange(f, dist, torch, preferred=pref, device="cuda")
        assert name == pref, (name, pref)
        assert f.L.pf_debug_exchange(f.h, 1 << 16) == 0
        if name == "rccl":
            assert f.L.pf_release_rccl(f.h) == 0      # communicator destroyed, callbacks cleared
            assert f.L.pf_debug_exchange(f.h, 1 << 16) != 0
        del keep
import bench     # the exit verdict of a multi-rank bench run travels on the backend's device (an NCCL-only group has no CPU backend)
assert bench.broadcast_verdict(dist, torch, True, "nccl") is True and bench.broadcast_verdict(dist, torch, False, "nccl") is False
dist.destroy_process_group()
print("TORCH_OK")
"""


def _run_isolated(code, token):
    """own process: the ROCm RCCL and the copy bundled with torch must not meet in one address space"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code % root], capture_output=True, text=True, timeout=600)
    assert token in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_rccl_exchange_single_rank():
    """the built-in RCCL exchange (dlopen, ncclCommInitRank, grouped send/recv, all-reduce) on a 1-rank communicator"""
    _run_isolated(_RCCL_CODE, "RCCL_OK")


def test_torch_exchange_single_rank():
    """bench.py's two exchange kinds after torch.distributed (backend nccl = RCCL) is up: `torch` = collectives on
    tensors aliasing the library's device buffers on the library's stream, `rccl` = the built-in exchange"""
    _run_isolated(_TORCH_CODE, "TORCH_OK")


def test_genic_slabs_match_single_rank(api):
    """every rank generates its own k-space slab of the reference's initial conditions: same field as one rank"""
    n, P = 64, 4
    args = (486604, 64 / 0.7, 0.25, 0.044, 0.7, 0.96)
    radii = np.array([2.0, 0.0])
    x, y = synth.invgrow_table("lcdm")

    def body(f, r):
        f.genic_density(*args, pknorm=2.0e7)
        f.set_invgrow(x, y)
        tv = f.sweep(radii)
        return tv, f.products()["Fmax"]

    with api.Fmax(n) as f1:
        f1.genic_density(*args, pknorm=2.0e7)
        f1.set_invgrow(x, y)
        tv1 = f1.sweep(radii)
        fm1 = f1.products()["Fmax"]
    res = run_ranks(api, n, P, body)
    nxl = n // P
    for r in range(P):
        assert np.allclose(res[r][0], tv1, rtol=1e-13)
        assert np.array_equal(res[r][1], fm1[r * nxl:(r + 1) * nxl])


def test_fft_module_seam_on_slabs(api):
    """forward / reverse transform, compute_derivative and the spectrum taps with host slabs in the boundary layout on
    P ranks (import and export regroup through one all-to-all each): bitwise the single-rank slabs"""
    n, P = 32, 4
    nxl = n // P
    rng = np.random.default_rng(5)
    real = rng.standard_normal((n, n, n))
    dk = synth.make_density(n, seed=9)
    dk[0, 0, 0] = -0.2 * n ** 3
    with api.Fmax(n) as f1:
        spec1 = f1.forward_transform(real)
        back1 = f1.reverse_transform(dk)
        d12 = f1.compute_derivative(dk, 1, 2, 1.0, 0)
        d30 = f1.compute_derivative(dk, 3, 0, 0.0, 1)
        f1.set_density(dk)
        dens1 = f1.density()

    def body(f, r):
        sl = slice(r * nxl, (r + 1) * nxl)
        out = [f.forward_transform(real[sl]), f.reverse_transform(dk[sl]), f.compute_derivative(dk[sl], 1, 2, 1.0, 0),
               f.compute_derivative(dk[sl], 3, 0, 0.0, 1)]
        f.set_density(dk[sl])
        out.append(f.density())
        return out

    res = run_ranks(api, n, P, body)
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        for got, want in zip(res[r], (spec1, back1, d12, d30, dens1)):
            assert np.array_equal(got, want[sl])
    assert np.array_equal(dens1, dk)


def test_snapshot_blocks_and_selection_on_slabs(api):
    """particle IDs are global (1 + global cell index, src/write_snapshot.c:648-664); selection and sort are per slab,
    like the reference's per-task sort"""
    n, P = 32, 4
    nxl = n // P
    x, y = synth.invgrow_table("lcdm")

    def body(f, r):
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y)
        f.sweep(np.array([2.0, 0.0]))
        return f.block("ID  "), f.block("FMAX"), f.select_sorted(1.0)

    res = run_ranks(api, n, P, body)
    for r in range(P):
        ids, fm, (idx, fs) = res[r]
        assert np.array_equal(ids, 1 + r * nxl * n * n + np.arange(nxl * n * n, dtype=np.uint32))
        sel = np.flatnonzero(fm >= np.float32(1.0))
        order = sel[np.lexsort((sel, -fm[sel].astype(np.float64)))]
        assert len(idx) > 0 and np.array_equal(idx, order.astype(np.uint32)) and np.array_equal(fs, fm[order])


def test_build_options_on_slabs(api):
    """SCALE_DEPENDENT growth tables (k_apply_growth needs the slab's global ky), per-radius splines, and the
    TABULATED_CT table (built by every rank for itself) on P ranks: bitwise the single-rank products"""
    n, P = 32, 4
    nxl = n // P
    dk = synth.make_density(n, seed=19)
    radii = np.array([2.0, 1.0, 0.0])
    splines = [synth.invgrow_table("lcdm", omega0=om) for om in (0.25, 0.30, 0.35)]
    g = synth.growth_multipliers()
    j = np.arange(10)
    tabs = [np.log10(abs(g[o]) * (1.0 + 0.04 * (o + 1) * j)) for o in range(4)]
    signs = [1.0, 1.0, -1.0, 1.0]
    var = np.array([0.9, 1.8, 6.0])

    def run(f, sl):
        f.set_density(dk[sl])
        for i, (x, y) in enumerate(splines):
            f.set_invgrow(x, y, ismooth=i)
        for o in range(4):
            f.set_growth_table(o + 1, tabs[o], sign=signs[o])
        f.compute_fmax(radii, do_lpt=True)
        a = f.products()
        f.set_tabulated_ct(var)
        f.sweep(radii)
        b = f.products()
        f.set_tabulated_ct([])
        return a, b

    with api.Fmax(n) as f1:
        a1, b1 = run(f1, slice(0, n))
    res = run_ranks(api, n, P, lambda f, r: run(f, slice(r * nxl, (r + 1) * nxl)))
    assert np.mean(a1["Fmax"] != b1["Fmax"]) > 0.3          # the table really replaced the direct solve
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        a, b = res[r]
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1"):
            assert np.array_equal(a[name], a1[name][sl]), name
        assert np.array_equal(b["Fmax"], b1["Fmax"][sl]) and np.array_equal(b["Rmax"], b1["Rmax"][sl])


@pytest.mark.parametrize("P,pipeline", [(2, 1), (4, 1), (8, 0), (16, 1)])
def test_band_limited_radii_exchange_only_their_rows(api, P, pipeline, monkeypatch):
    """smoothing radii whose window is below 2^-60 beyond |k| = band: only the in-band slab rows of each block go
    through the all-to-all (pf_alltoallv_fn); same products bit for bit as one rank, and as whole-block exchanges,
    with fewer bytes on the wire"""
    n = 64
    nxl = n // P
    monkeypatch.setenv("PF_PIPELINE", str(pipeline))
    monkeypatch.setenv("PF_REPLICATE_DK", "0")                  # the sweep's own transposes are what is under test here
    dk = synth.make_density(n, seed=23)
    radii = np.array([8.0, 6.0, 4.0, 3.2, 1.0, 0.0])           # bands 12, 16, 24, 30, none, none of 32
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g)
        f.reset_kernel_stats()
        tv = f.compute_fmax(radii, do_lpt=True)
        ex = [k for k in f.kernel_stats() if k["name"] == "exchange"]
        return tv, f.products(), (ex[0]["alg_bytes"] if ex else 0.0)

    with api.Fmax(n) as f1:
        f1.set_density(dk); f1.set_invgrow(x, y); f1.set_growth(g)
        tv1 = f1.compute_fmax(radii, do_lpt=True)
        p1 = f1.products()
    out = {}
    for rows in ("1", "0"):
        monkeypatch.setenv("PF_EXCHANGE_ROWS", rows)
        out[rows] = run_ranks_timed(api, n, P, body)
    for r in range(P):
        sl = slice(r * nxl, (r + 1) * nxl)
        for rows in ("1", "0"):
            tv, p, nbytes = out[rows][r]
            assert np.allclose(tv, tv1, rtol=1e-13)
            for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1"):
                assert np.array_equal(p[name], p1[name][sl]), (rows, name)
    sent = {rows: sum(o[2] for o in out[rows]) for rows in ("1", "0")}
    assert 0.0 < sent["1"] < 0.8 * sent["0"], sent      # 4 of 6 radii keep 39-95 % of their rows and 41-97 % of their columns, LPT exchanges whole


@pytest.mark.parametrize("replicate", [0, 1])
@pytest.mark.parametrize("P,delay_us,fb", [(2, 4000, 8), (4, 2000, 8), (8, 1000, 8), (4, 2000, 4)])
def test_pipelined_exchange_with_late_communication(api, P, delay_us, fb, replicate, monkeypatch):
    """The fabric's all-to-all is asynchronous on the device (events, no stream synchronisation), so the exchange of
    transform i+1 really runs on the communication stream beside the y/z passes and the solve of transform i.  Here every
    exchange first idles its stream for milliseconds -- longer than all the kernels of a radius at this size -- so the
    compute stream runs far ahead: whatever the pipeline of pf_api.hip fails to wait for (receive set not yet filled,
    send set overwritten before it was pulled, receive set overwritten while the y-pass still reads it) changes the
    results, which must stay bitwise those of one rank.  Sweep (3 fields per item, band-limited and full radii, odd count)
    and the displacement pipelines (2 fields, 1 field per item).  replicate = 1: the sweep exchanges nothing, the LPT part still
    does (and reuses the receive set the sweep wrote in place)."""
    monkeypatch.setenv("PF_REPLICATE_DK", str(replicate))
    n = 64
    nxl = n // P
    dk = synth.make_density(n, seed=29)
    radii = np.array([8.0, 4.0, 3.2, 1.0, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        tv2 = f.compute_fmax(radii[1:], do_lpt=True)       # a second run: buffer sets start from the other parity
        return tv, tv2, f.products()

    with api.Fmax(n, field_bytes=fb) as f1:
        f1.set_density(dk); f1.set_invgrow(x, y); f1.set_growth(g)
        tv1 = f1.compute_fmax(radii, do_lpt=True)
        tv1b = f1.compute_fmax(radii[1:], do_lpt=True)
        p1 = f1.products()
    res = run_ranks(api, n, P, body, field_bytes=fb, delay_us=delay_us)
    for r in range(P):
        tv, tv2, p = res[r]
        sl = slice(r * nxl, (r + 1) * nxl)
        assert np.allclose(tv, tv1, rtol=1e-13) and np.allclose(tv2, tv1b, rtol=1e-13)
        for name in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1"):
            assert np.array_equal(p[name], p1[name][sl]), (r, name)
        a, b = p["Vel_3LPT_2"].astype(np.float64), p1["Vel_3LPT_2"][sl].astype(np.float64)
        assert np.max(np.abs(a - b)) <= 2e-7 * np.max(np.abs(b))


@pytest.mark.parametrize("fault,n,P,delay_us", [("recv", 64, 4, 2000), ("send", 64, 4, 0)])
def test_late_communication_exposes_a_missing_wait(api, fault, n, P, delay_us, monkeypatch):
    """the same set-up with one wait of the pipeline taken out on purpose (PF_DEBUG_PIPELINE_FAULT, read in pf_create): the
    results must change -- the test above can see what it is there to see.  "recv": the compute stream does not wait for
    the exchange it consumes (exposed by a late communication stream); "send": the exchange does not wait for the x-pass
    that fills its blocks (the injected fault also holds the compute stream back by 2 ms before every x-pass, so the copies
    are certain to start first)"""
    monkeypatch.setenv("PF_REPLICATE_DK", "0")                  # every radius goes through the exchange pipeline
    nxl = n // P
    dk = synth.make_density(n, seed=29)
    radii = np.array([8.0, 4.0, 3.2, 1.0, 0.0])
    x, y = synth.invgrow_table("lcdm")

    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y)
        f.sweep(radii)
        return f.products()["Fmax"]

    with api.Fmax(n) as f1:
        f1.set_density(dk); f1.set_invgrow(x, y)
        f1.sweep(radii)
        fm1 = f1.products()["Fmax"]
    monkeypatch.setenv("PF_DEBUG_PIPELINE_FAULT", fault)
    res = run_ranks(api, n, P, body, delay_us=delay_us)
    assert any(not np.array_equal(res[r], fm1[r * nxl:(r + 1) * nxl]) for r in range(P))


def test_config4_all_eight_ranks_at_full_size_on_one_gpu(api, monkeypatch):
    """BASELINE config 4 on its own workload -- 1024^3, fp64, eight ranks, Fmax sweep + 2LPT / 3LPT -- with all eight ranks alive on ONE GPU,
    their all-to-alls moved by the in-process fabric (device-to-device copies where an 8-GPU node has RCCL over xGMI): 128-plane slabs, the
    band-limited exchanges of the smoothed radii, the transposes of the LPT source spectra, every reduction.  One buffer set per rank
    (PF_PIPELINE=0: 31.5 GB each; the pipeline's second set -- 38 GB each -- would not fit eight times, and is covered at 64^3 .. 256^3
    above).  Fmax, Rmax and the first three displacement fields bit for bit the single-GPU run, the 3LPT(b) field to the order of its
    all-reduced sum.  What is left for real peers: the RCCL transport itself."""
    n, P = 1024, 8
    monkeypatch.setenv("PF_PIPELINE", "0")
    monkeypatch.setenv("PF_REPLICATE_DK", "0")
    nxl = n // P
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    radii = synth.radii_ladder(12)[[2, 8, 11]]          # one band-limited radius, one full, R = 0
    names = ("FMAX", "RMAX", "ZEL ", "2LPT", "31PT", "32PT")
    with api.Fmax(n) as f1:
        f1.synth_density(synth.SEED, 2.5, -2.0)
        f1.set_invgrow(x, y); f1.set_growth(g)
        tv1 = f1.compute_fmax(radii, do_lpt=True)
        pdf1 = f1.Fmax_PDF()
        gold = {nm: f1.block(nm) for nm in names}

    def body(f, r):
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        pdf = f.Fmax_PDF()
        return tv, pdf, {nm: f.block(nm) for nm in names}, f.device_bytes

    res = run_ranks(api, n, P, body)
    nc = nxl * n * n
    for r in range(P):
        tv, pdf, blk, dev = res[r]
        assert np.allclose(tv, tv1, rtol=1e-13) and np.array_equal(pdf, pdf1), r
        assert dev < 32e9
        for nm in names[:5]:
            assert np.array_equal(blk[nm], gold[nm][r * nc:(r + 1) * nc]), (r, nm)
        a, b = blk["32PT"].astype(np.float64), gold["32PT"][r * nc:(r + 1) * nc].astype(np.float64)
        assert np.max(np.abs(a - b)) <= 2e-7 * np.max(np.abs(b)), r
    assert np.sqrt(tv1[-1]) == pytest.approx(2.5, rel=1e-10)


@pytest.mark.parametrize("fb", [8, 4])
def test_loopback_slab_passes_on_fields_whose_answer_is_known(api, fb):
    """512^3 on eight ranks, ONE rank on its own behind the loopback exchange, spectrum exchanged (PF_REPLICATE_DK=0): every pass of
    such a slab is linear, and a field that repeats with the slab thickness along x and whose spectrum repeats with it along ky comes
    back from the loopback exactly as it would from the peers (every rank holds the same numbers).  So the x-pass, the block layout the
    exchange moves, the y-pass on the received blocks and the z-passes are held against numpy at the slab geometry of the run, in
    both directions: reverse_transform (x, exchange, y, c2r) and forward_transform (r2c, y, exchange, x)."""
    n, P = 512, 8
    nxl, h = n // P, n // 2
    rng = np.random.default_rng(40 + fb)
    tol = 2e-13 if fb == 8 else 3e-5
    os.environ["PF_REPLICATE_DK"] = "0"
    try:
        for r in (0, 5):
            with api.Fmax(n, rank=r, nranks=P, field_bytes=fb) as f:
                f.set_transposed_spectra(True)                       # spectra cross the boundary as this rank's ky-slab: no regrouping exchange
                f._chk(f.L.pf_set_loopback_exchange(f.h, 1 << 20))   # every hand-back copies
                # reverse: S[kx, ky, kz] = a[kx] g[ky mod nyl, kz], a on multiples of P and Hermitian (so that A(x) = ifft(a) is real, of period nxl)
                a = np.zeros(n, dtype=np.complex128)
                for kx in range(0, h + 1, P):
                    a[kx] = rng.standard_normal() + (0.0 if kx in (0, h) else 1j * rng.standard_normal())
                    a[(-kx) % n] = np.conj(a[kx])
                A = np.fft.ifft(a)
                assert np.max(np.abs(A.imag)) < 1e-15 and np.allclose(A[:nxl], A[nxl:2 * nxl])
                g = rng.standard_normal((nxl, h + 1)) + 1j * rng.standard_normal((nxl, h + 1))
                spec = np.ascontiguousarray(g[:, None, :] * a[None, :, None])                # this rank's ky-slab [nyl][kx][kz]: the same for every rank
                got = f.reverse_transform(spec)
                R = np.fft.irfft(np.fft.ifft(np.tile(g, (P, 1)), axis=0), n=n, axis=1)     # the (y, z) factor of the whole box's field
                want = A.real[r * nxl:(r + 1) * nxl, None, None] * R[None, :, :]
                assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), ("reverse", r, fb)
                # forward: f(x, y, z) = A(x mod nxl) c(y) w(z), c on multiples of P (its spectrum repeats with nyl)
                Ax = rng.standard_normal(nxl)
                cy = np.zeros(n); cy[::P] = rng.standard_normal(n // P)
                wz = rng.standard_normal(n)
                real = np.ascontiguousarray(Ax[:, None, None] * cy[None, :, None] * wz[None, None, :])
                got = f.forward_transform(real)                                               # [nyl (ky of this rank)][kx][kz]
                Fy = np.fft.fft(cy)
                want = Fy[r * nxl:(r + 1) * nxl, None, None] * np.fft.fft(np.tile(Ax, P))[None, :, None] * np.fft.rfft(wz)[None, None, :]
                assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), ("forward", r, fb)
    finally:
        del os.environ["PF_REPLICATE_DK"]


@pytest.mark.parametrize("n,fb,ranks", [(512, 8, tuple(range(8))), (512, 4, tuple(range(8))), (1024, 8, tuple(range(8)))])
def test_loopback_slab_with_the_whole_spectrum_is_the_single_gpu_runs_slab(api, n, fb, ranks):
    """512^3 on eight ranks, ONE rank on its own, the whole delta(k) on the rank (PF_REPLICATE_DK=1) and generated by it (pf_genic_density
    behind the loopback exchange): the sweep exchanges nothing, and the rank's Fmax / Rmax are the single-GPU run's on its slab, bit for
    bit -- what tests/test_gpu_config5.py relies on at 2048^3, where no single-GPU run exists.  1024^3 fp64 on eight ranks is BASELINE
    config 4 at its own size: every rank of it runs its sweep on its own slab geometry (128 planes, the kernels and launch shapes of
    the 8-GPU run) and reproduces the single-GPU box bit for bit, variances and histogram adding up; what only real peers can show is the exchange itself and the LPT half behind it"""
    P = 8
    nxl = n // P
    x, y = synth.invgrow_table("lcdm")
    radii = np.array([6.0, 1.5, 0.0])
    genic = dict(seed=77, box_true_mpc=float(n) / 0.7, omega0=0.25, omega_baryon=0.044, hubble100=0.7, primordial_index=0.96, pknorm=2.0e7)
    with api.Fmax(n, field_bytes=fb) as f1:
        f1.genic_density(**genic)
        f1.set_invgrow(x, y)
        tv1 = f1.sweep(radii)
        fm1 = f1.block("FMAX").reshape(n, n, n)
        rm1 = f1.block("RMAX").reshape(n, n, n)
        pdf1 = f1.Fmax_PDF()
    os.environ["PF_REPLICATE_DK"] = "1"
    try:
        tv, pdf = np.zeros(3), np.zeros(210, dtype=np.uint64)
        for r in ranks:
            with api.Fmax(n, rank=r, nranks=P, field_bytes=fb) as f:
                f.set_invgrow(x, y)
                f._chk(f.L.pf_set_loopback_exchange(f.h, 0))
                f.genic_density(**genic)
                tv += f.sweep(radii)
                pdf += f.Fmax_PDF()
                assert np.array_equal(f.block("FMAX").reshape(nxl, n, n), fm1[r * nxl:(r + 1) * nxl]), (r, fb)
                assert np.array_equal(f.block("RMAX").reshape(nxl, n, n), rm1[r * nxl:(r + 1) * nxl]), (r, fb)
        if len(ranks) == P:      # every slab ran: the contributions add up to the box's variances and histogram
            assert np.allclose(tv, tv1, rtol=1e-12) and np.array_equal(pdf, pdf1)
    finally:
        del os.environ["PF_REPLICATE_DK"]


def test_one_rank_slab_with_loopback_exchange(api):
    """the measurement aid behind `bench.py --slab-of P` (pf_set_loopback_exchange): one rank of a P-rank decomposition runs the whole
    step on its own, its own blocks handed back by the all-to-all -- copied at first, then not moved at all.  Nothing about the
    numbers is claimed except that they are finite and that the step runs in both modes, with fp64 and fp32 fields, with the
    spectrum replicated or exchanged; a context without an exchange refuses.  (What a loopback slab CAN be held against: the two tests above.)"""
    x, y = synth.invgrow_table("lcdm")
    radii = synth.radii_ladder(12)[[0, 6, 11]] * (64 / 1024.0)
    radii[-1] = 0.0
    for fb, nranks, rep in ((8, 4, "1"), (8, 8, "0"), (4, 4, "0")):
        os.environ["PF_REPLICATE_DK"] = rep
        try:
            with api.Fmax(64, rank=0, nranks=nranks, field_bytes=fb) as f:
                f.set_invgrow(x, y)
                with pytest.raises(api.PinfmaxError):
                    f.synth_density(synth.SEED, 2.5, -2.0)          # no exchange installed
                f._chk(f.L.pf_set_loopback_exchange(f.h, 1 << 20))
                f.synth_density(synth.SEED, 2.5, -2.0)
                tv = f.compute_fmax(radii, do_lpt=True)
                f._chk(f.L.pf_set_loopback_exchange(f.h, 0))
                tv2 = f.compute_fmax(radii, do_lpt=True)
                fm = f.block("FMAX")
                assert np.isfinite(tv).all() and np.isfinite(tv2).all() and np.isfinite(fm).all()
                assert fm.size == 64 ** 3 // nranks
        finally:
            del os.environ["PF_REPLICATE_DK"]
