"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/pinfmax.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pinocchio_amd", "libpinfmax_hip.so")):
        g.build()
    from pinocchio_amd import _lib
    return _lib


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pinfmax.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pf_alltoall_fn", "pf_allreduce_fn"}
    assert len(declared) >= 30
    L = lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in pinfmax.h but not exported"
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)


def test_product_layout_matches_reference_record(lib):
    lay = lib.ProductLayout()
    lib.load().pf_layout_3lpt(C.byref(lay))
    # offsets of product_data with -DTWO_LPT -DTHREE_LPT (src/pinocchio.h:233-259)
    assert (lay.stride, lay.off_Rmax, lay.off_Fmax, lay.off_Vel, lay.off_Vel_2LPT, lay.off_Vel_3LPT_1,
            lay.off_Vel_3LPT_2) == (56, 0, 4, 8, 20, 32, 44)
    from pinocchio_amd.api import PRODUCT_DTYPE
    assert PRODUCT_DTYPE.itemsize == 56
    assert [PRODUCT_DTYPE.fields[k][1] for k in ("Rmax", "Fmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2")] == \
        [0, 4, 8, 20, 32, 44]


def test_create_rejects_bad_configs_and_missing_gpu(lib, capfd):
    import torch
    L = lib.load()
    h = C.c_void_p()
    # odd, too small, too large, rank count not dividing / not a power of two; sizes that are not a power of two are
    # one-rank fp64 only (chirp-z transform path)
    for n, nranks, fb in ((101, 1, 8), (2, 1, 8), (8192, 1, 8), (64, 3, 8), (100, 2, 8), (100, 1, 4), (4096, 1, 4)):
        cfg = lib.Config(n=n, rank=0, nranks=nranks, device=0, field_bytes=fb, flags=0)
        assert L.pf_create(C.byref(h), C.byref(cfg)) != 0
        assert not h.value
        assert b"no HIP device" not in L.pf_last_error()
    cfg = lib.Config(n=64, rank=0, nranks=1, device=0, field_bytes=2, flags=0)
    assert L.pf_create(C.byref(h), C.byref(cfg)) != 0
    if not torch.cuda.is_available():
        # no silent CPU path: creation must fail loudly, in the reference's error format
        cfg = lib.Config(n=64, rank=0, nranks=1, device=0, field_bytes=8, flags=0)
        assert L.pf_create(C.byref(h), C.byref(cfg)) != 0
        assert b"no HIP device" in L.pf_last_error()
        out = capfd.readouterr().out
        assert "ERROR on task 0" in out
        from pinocchio_amd import api
        with pytest.raises(api.PinfmaxError):
            api.Fmax(64)


def test_memory_plan_of_the_baseline_configurations(lib, monkeypatch):
    """pf_plan_bytes (no device needed): what a rank of each BASELINE configuration holds at most -- the numbers the documents quote, and
    the formula pf_create's preflight holds against hipMemGetInfo.  Config 5 (2048^3 on eight ranks, fp32 fields): 199.5 GB per rank
    (round 5: 234), 234.3 with the spectrum replicated (how tests/test_gpu_config5.py runs it)"""
    from pinocchio_amd import api
    monkeypatch.delenv("PF_REPLICATE_DK", raising=False)
    at_create, peak = api.plan_bytes(2048, 8, 4)
    assert 199.0e9 < peak < 200.0e9 and at_create < peak
    assert 225.0e9 < api.plan_bytes(1024, 1, 8)[1] < 227.0e9             # the metric's box on one GPU
    assert api.plan_bytes(1024, 8, 8)[1] < 39.0e9                        # config 4, per rank
    assert api.plan_bytes(2048, 1, 8)[1] > 288.0e9                       # ... and what no single GPU holds
    monkeypatch.setenv("PF_REPLICATE_DK", "1")
    assert 234.0e9 < api.plan_bytes(2048, 8, 4)[1] < 235.0e9
    with pytest.raises(api.PinfmaxError):
        api.plan_bytes(100, 3, 8)
