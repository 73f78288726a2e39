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


@pytest.mark.parametrize("world,n,rs", [(2, 32, 6.0), (4, 32, 4.0), (3, 48, 4.0)])
def test_band_limited_exchange_model_matches_oracle(world, n, rs):
    """the band-limited form of the exchange (in-band slab rows x in-band kz columns only, one interval of rows per
    rank) on gloo: still the oracle's second derivatives, to the window weight that was dropped (< 2^-56 of the rms).
    Three ranks on a 48^3 grid (round 6: mixed-radix grids take any rank count that divides them): the middle slab straddles n / 2 and
    holds rows of both ends of the band -- it sends the hull of the two"""
    import slab_model
    band = slab_model.hess_band(n, rs)
    assert band < n // 2
    rows = [slab_model.band_rows(n, world, p, band) for p in range(world)]
    if world % 2 == 0:
        assert sum(hi - lo for lo, hi in rows) == 2 * band + 1          # every in-band ky exactly once
    else:
        assert rows[world // 2] == (0, n // world) and sum(hi - lo for lo, hi in rows) > 2 * band + 1   # the middle slab whole
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


# ---- the product's exchange negotiation (pinocchio_amd/dist.py) under a real 2-process gloo group ----
def _negotiate_worker(rank, world, port, scenario, q):
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pinocchio_amd import dist as pfdist

        calls = []

        class FakeLib:
            def __init__(self):
                self.installed = None

            def pf_exchange_buffers(self, h, send, recv, cap):
                cap._obj.value = 3 << 24      # (what byref() wraps: the capacity the self-test is sized from)
                return 0

            def pf_debug_exchange(self, h, nbytes):
                assert 0 < nbytes <= (1 << 24) // 2 and nbytes % 8 == 0
                calls.append(("selftest", self.installed))
                bad = scenario.get("selftest_fails", {}).get(self.installed, ())
                return 1 if rank in bad else 0

        class FakeCtx:
            L = FakeLib()
            h = None

        def make(name):
            class Kind:
                def __init__(self, f, dist_, torch_, device):
                    self.f = f

                def can_bind(self):
                    calls.append(("can_bind", name))
                    if rank in scenario.get("bind_raises", {}).get(name, ()):
                        raise OSError("cannot load librccl")
                    return rank not in scenario.get("cannot_bind", {}).get(name, ())

                def setup(self):
                    calls.append(("setup", name))
                    # a set-up that contains a collective: must only be entered when every rank enters it
                    t = torch.tensor([rank])
                    dist.broadcast(t, src=0)
                    self.f.L.installed = name
                    return rank not in scenario.get("setup_fails", {}).get(name, ())

                def release(self):
                    calls.append(("release", name))
                    self.f.L.installed = None
            Kind.name = name
            return Kind

        kinds = {"rccl": make("rccl"), "torch": make("torch")}
        f = FakeCtx()
        try:
            name, _ = pfdist.negotiate_exchange(f, dist, torch, preferred="rccl", device="cpu", kinds=kinds, log=lambda m: None)
        except RuntimeError:
            name = None
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, name, calls))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), []))


@pytest.mark.parametrize("scenario,chosen", [
    ({}, "rccl"),
    ({"cannot_bind": {"rccl": (1,)}}, "torch"),           # one rank cannot dlopen RCCL: nobody enters its set-up
    ({"bind_raises": {"rccl": (0,)}}, "torch"),
    ({"setup_fails": {"rccl": (0,)}}, "torch"),           # ncclCommInitRank fails on one rank: released on all, then torch
    ({"selftest_fails": {"rccl": (1,)}}, "torch"),        # wrong words on one rank only
    ({"selftest_fails": {"rccl": (1,), "torch": (0,)}}, None),
])
def test_exchange_negotiation_never_leaves_a_rank_behind(scenario, chosen):
    """bench.py --gpus N: every rank must reach the same decision, through the same number of collectives, whatever
    fails and wherever (a one-sided failure used to leave the other ranks inside a collective)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_negotiate_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [chosen] * world, res
    c0, c1 = res[0][2], res[1][2]
    assert c0 == c1, (c0, c1)                      # the same steps in the same order on both ranks
    if scenario.get("cannot_bind") or scenario.get("bind_raises"):
        assert ("setup", "rccl") not in c0         # the vote came first
    if scenario.get("setup_fails") or scenario.get("selftest_fails"):
        assert ("release", "rccl") in c0           # nothing of the failed kind stays behind


# ---- bench.py --gpus N invoked bare: it starts its own ranks ----
def test_bench_self_launches_its_ranks_without_touching_the_gpu():
    """`python bench.py --gpus 2 --dry-launch` with no WORLD_SIZE in the environment: the parent starts two fresh children
    through torch.distributed.run (RANK 0 and 1, one rendezvous on 127.0.0.1), relays their lines, returns their exit
    code, and has itself neither imported torch nor loaded the HIP library."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # (--n and --ns: prefixes of torch.distributed.run's own options -- they must reach the ranks all the same)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "64", "--ns", "3",
                        "--dry-launch"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    ranks = sorted((ln["rank"], ln["local_rank"], ln["world"]) for ln in lines if ln.get("role") == "rank")
    assert ranks == [(0, 0, 2), (1, 1, 2)], lines
    assert all(ln["launched_by_bench"] and ln["gpu_untouched"] for ln in lines if ln.get("role") == "rank")
    assert len({ln["master"] for ln in lines if ln.get("role") == "rank"}) == 1
    parent = [ln for ln in lines if ln.get("role") == "parent"]
    assert len(parent) == 1 and parent[0]["parent_gpu_untouched"] and parent[0]["children_rc"] == 0 and parent[0]["json_lines_relayed"] == 2
    assert "--nproc-per-node=2" in parent[0]["command"] and "torch.distributed.run" in parent[0]["command"]
    assert all(ln["args"]["n"] == 64 and ln["args"]["ns"] == 3 and ln["args"]["steps"] == 2 for ln in lines if ln.get("role") == "rank")
    # under a launcher (WORLD_SIZE set) nothing is started: a mismatch is still refused
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--dry-launch"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in r2.stderr


def _verdict_worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bench
        # rank 0 holds the verdict; the others arrive with the opposite opinion and leave with rank 0's
        got = [bench.broadcast_verdict(dist, torch, (rank == 0) == v, "gloo-host") for v in (True, False)]
        dist.destroy_process_group()
        q.put((rank, got))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_bench_exit_verdict_reaches_every_rank():
    """bench.broadcast_verdict on a gloo group of two: the flag that makes every rank of a multi-GPU run exit non-zero after a failed result check"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verdict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == [True, False] and res[1] == [True, False], res
