"""world_size-2 (and 4) gloo test of the slab decomposition on CPU: the numpy
model of the library's multi-rank pipeline (tests/slab_model.py) against the
single-rank oracle."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, rs, q, prune=False):
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle_lib
        import slab_model
        from pinocchio_amd import synth
        dk = synth.make_density(n, seed=23)
        dk[0, 0, 0] = 0.4 * n ** 3
        nxl = n // world
        got = slab_model.hessian_slab(dist, torch, dk[rank * nxl:(rank + 1) * nxl], rs, prune=prune)
        o = oracle_lib.Oracle(n, 1)
        o.set_density(dk)
        want = o.second_derivatives(rs)
        amp = max(np.max(np.abs(w)) for w in want)
        err = max(np.max(np.abs(g - w[rank * nxl:(rank + 1) * nxl])) for g, w in zip(got, want)) / amp
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, float(err)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world,n,rs", [(2, 16, 1.0), (2, 32, 0.0), (4, 16, 0.7)])
def test_slab_model_matches_oracle(world, n, rs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, rs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert isinstance(err, float), (rank, err)
        assert err < 1e-12, (rank, err)


@pytest.mark.parametrize("world,n,rs", [(2, 32, 6.0), (4, 32, 4.0)])
def test_band_limited_exchange_model_matches_oracle(world, n, rs):
    """the band-limited form of the exchange (in-band slab rows x in-band kz columns only, one interval of rows per
    rank) on gloo: still the oracle's second derivatives, to the window weight that was dropped (< 2^-56 of the rms)"""
    import slab_model
    band = slab_model.hess_band(n, rs)
    assert band < n // 2
    rows = [slab_model.band_rows(n, world, p, band) for p in range(world)]
    assert sum(hi - lo for lo, hi in rows) == 2 * band + 1          # every in-band ky exactly once
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, rs, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert isinstance(err, float), (rank, err)
        assert err < 1e-12, (rank, err)
