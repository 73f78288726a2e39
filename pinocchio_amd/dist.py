"""Multi-GPU plumbing for a Python host harness (bench.py): chooses and installs the slab all-to-all and the small
all-reduces of one rank's context.

kind="rccl":  the library's own exchange (csrc/pf_rccl.cpp: grouped ncclSend/ncclRecv over xGMI); torch.distributed only
              broadcasts the 128-byte ncclUniqueId.
kind="torch": every exchange is a torch.distributed all_to_all_single (all_to_all with per-sender sizes for the row-range
              form of the band-limited radii) on tensors that alias the library's device buffers (backend "nccl" = RCCL
              on ROCm), enqueued on the library's stream.

`negotiate_exchange` is collective and cannot hang on a one-sided failure: before any rank enters a communicator set-up
or a data collective of a kind, ALL ranks vote on a purely local check; after each collective step they vote on its
outcome; a kind that loses a vote is released everywhere (communicator destroyed) before the next one is tried.  The
votes themselves go through the process group that brought the ranks up (it works, or the job would not have started).

PyTorch is plumbing here (process group, streams), not the compute path.
"""
from __future__ import annotations

import ctypes as C
import sys

from . import _lib


class _DevMem:
    """exposes a raw device pointer through __cuda_array_interface__"""

    def __init__(self, ptr: int, nbytes: int, typestr: str = "|u1", itemsize: int = 1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 3}


def _vote(dist, torch, ok: bool, device) -> bool:
    """True only if every rank says ok"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class RcclKind:
    """the library's built-in exchange"""
    name = "rccl"

    def __init__(self, f, dist, torch, device):
        self.f, self.dist, self.torch, self.device = f, dist, torch, device
        self.keep = None

    def can_bind(self) -> bool:  # host side only: dlopen + symbols
        return bool(self.f.L.pf_rccl_available())

    def setup(self) -> bool:
        L, dist, torch = self.f.L, self.dist, self.torch
        idbuf = (C.c_ubyte * 128)()
        have_id = True
        if dist.get_rank() == 0:
            have_id = L.pf_rccl_unique_id(C.cast(idbuf, C.c_void_p)) == 0
        # every rank learns whether rank 0 has an id BEFORE anybody calls ncclCommInitRank
        if not _vote(dist, torch, have_id, self.device):
            return False
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, src=0)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(bytes(t.cpu().tolist()))
        self.keep = idbuf
        return L.pf_init_rccl(self.f.h, C.cast(idbuf, C.c_void_p)) == 0

    def release(self):
        self.f.L.pf_release_rccl(self.f.h)
        self.keep = None


class TorchKind:
    """torch.distributed collectives on tensors aliasing the library's buffers"""
    name = "torch"

    def __init__(self, f, dist, torch, device):
        self.f, self.dist, self.torch, self.device = f, dist, torch, device
        self.keep = None

    def can_bind(self) -> bool:
        return True

    def setup(self) -> bool:
        f, dist, torch = self.f, self.dist, self.torch
        world = dist.get_world_size()

        def _a2a(user, send, recv, bytes_per_peer, stream):
            try:
                nbytes = bytes_per_peer * world
                s = torch.as_tensor(_DevMem(send, nbytes), device="cuda")
                r = torch.as_tensor(_DevMem(recv, nbytes), device="cuda")
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_to_all_single(r, s)
                return 0
            except Exception as e:  # noqa: BLE001 -- surfaces as "all-to-all failed" in the library
                print("exchange callback failed:", e, flush=True)
                return 1

        def _a2av(user, send, recv, block_bytes, send_off, send_bytes, recv_off, recv_bytes, stream):
            # row-range form (band-limited radii): per-sender sizes, empty messages allowed
            try:
                s = torch.as_tensor(_DevMem(send, block_bytes * world), device="cuda")
                r = torch.as_tensor(_DevMem(recv, block_bytes * world), device="cuda")
                ins = [s[q * block_bytes + send_off:q * block_bytes + send_off + send_bytes] for q in range(world)]
                outs = [r[p * block_bytes + recv_off[p]:p * block_bytes + recv_off[p] + recv_bytes[p]] for p in range(world)]
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_to_all(outs, ins)
                return 0
            except Exception as e:  # noqa: BLE001
                print("row-range exchange callback failed:", e, flush=True)
                return 1

        def _ared(user, buf, count, is_u64, stream):
            try:
                t = torch.as_tensor(_DevMem(buf, count * 8, "<i8" if is_u64 else "<f8", 8), device="cuda")
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_reduce(t)
                return 0
            except Exception as e:  # noqa: BLE001
                print("all-reduce callback failed:", e, flush=True)
                return 1

        cb1, cb2, cb3 = _lib.ALLTOALL_FN(_a2a), _lib.ALLREDUCE_FN(_ared), _lib.ALLTOALLV_FN(_a2av)
        self.keep = (cb1, cb2, cb3)
        L = f.L
        return L.pf_set_exchange(f.h, cb1, None) == 0 and L.pf_set_exchange_rows(f.h, cb3, None) == 0 and \
            L.pf_set_allreduce(f.h, cb2, None) == 0

    def release(self):
        self.keep = None


class HostStagedKind:
    """Bring-up / test transport for a process group WITHOUT device collectives (backend "gloo"): every exchange is staged
    through host memory -- wait for the library's stream, copy the send blocks to the host, exchange them with the peers by
    point-to-point messages of the CPU group, copy what arrived into the receive blocks.  Slow by construction; what it is for
    is running the slab path with world_size > 1 real processes where RCCL has no second GPU to talk to (two ranks sharing one
    device), through the same callback ABI (pf_set_exchange / pf_set_exchange_rows / pf_set_allreduce) the device kinds use.
    Not in KINDS: bench.py never falls back to it."""
    name = "host"

    def __init__(self, f, dist, torch, device):
        self.f, self.dist, self.torch = f, dist, torch
        self.keep = None

    def can_bind(self) -> bool:
        return True

    def _pairwise(self, sends, recvs):
        """sends[q] / recvs[p]: CPU uint8 tensors (possibly empty) for every peer; the block to oneself is copied"""
        dist, rank, world = self.dist, self.dist.get_rank(), self.dist.get_world_size()
        if recvs[rank].numel():
            recvs[rank].copy_(sends[rank])
        ops = []
        for q in range(world):
            if q == rank:
                continue
            if sends[q].numel():
                ops.append(dist.P2POp(dist.isend, sends[q], q))
            if recvs[q].numel():
                ops.append(dist.P2POp(dist.irecv, recvs[q], q))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def setup(self) -> bool:
        f, dist, torch = self.f, self.dist, self.torch
        world = dist.get_world_size()

        def dev(ptr, nbytes):
            return torch.as_tensor(_DevMem(ptr, nbytes), device="cuda")

        def _a2a(user, send, recv, bytes_per_peer, stream):
            try:
                st = torch.cuda.ExternalStream(stream)
                with torch.cuda.stream(st):
                    st.synchronize()
                    host = dev(send, bytes_per_peer * world).cpu()
                    got = torch.empty_like(host)
                    self._pairwise([host[q * bytes_per_peer:(q + 1) * bytes_per_peer] for q in range(world)],
                                   [got[p * bytes_per_peer:(p + 1) * bytes_per_peer] for p in range(world)])
                    dev(recv, bytes_per_peer * world).copy_(got)
                    st.synchronize()
                return 0
            except Exception as e:  # noqa: BLE001 -- surfaces as "all-to-all failed" in the library
                print("host-staged exchange failed:", repr(e), flush=True)
                return 1

        def _a2av(user, send, recv, block_bytes, send_off, send_bytes, recv_off, recv_bytes, stream):
            try:
                st = torch.cuda.ExternalStream(stream)
                with torch.cuda.stream(st):
                    st.synchronize()
                    s = dev(send, block_bytes * world)
                    r = dev(recv, block_bytes * world)
                    sends = [s[q * block_bytes + send_off:q * block_bytes + send_off + send_bytes].cpu() for q in range(world)]
                    recvs = [torch.empty(int(recv_bytes[p]), dtype=torch.uint8) for p in range(world)]
                    self._pairwise(sends, recvs)
                    for p_ in range(world):
                        if recv_bytes[p_]:
                            r[p_ * block_bytes + recv_off[p_]:p_ * block_bytes + recv_off[p_] + recv_bytes[p_]].copy_(recvs[p_])
                    st.synchronize()
                return 0
            except Exception as e:  # noqa: BLE001
                print("host-staged row-range exchange failed:", repr(e), flush=True)
                return 1

        def _ared(user, buf, count, is_u64, stream):
            try:
                st = torch.cuda.ExternalStream(stream)
                with torch.cuda.stream(st):
                    st.synchronize()
                    t = torch.as_tensor(_DevMem(buf, count * 8, "<i8" if is_u64 else "<f8", 8), device="cuda")
                    host = t.cpu()
                    dist.all_reduce(host)
                    t.copy_(host)
                    st.synchronize()
                return 0
            except Exception as e:  # noqa: BLE001
                print("host-staged all-reduce failed:", repr(e), flush=True)
                return 1

        cb1, cb2, cb3 = _lib.ALLTOALL_FN(_a2a), _lib.ALLREDUCE_FN(_ared), _lib.ALLTOALLV_FN(_a2av)
        self.keep = (cb1, cb2, cb3)
        L = f.L
        return L.pf_set_exchange(f.h, cb1, None) == 0 and L.pf_set_exchange_rows(f.h, cb3, None) == 0 and \
            L.pf_set_allreduce(f.h, cb2, None) == 0

    def release(self):
        self.keep = None


KINDS = {"rccl": RcclKind, "torch": TorchKind}


def negotiate_exchange(f, dist, torch, preferred: str = "rccl", device="cuda", kinds=None, selftest_bytes: int = 1 << 20, log=None,
                       votes=None):
    """Collective over the process group.  Tries `preferred`, then the other kinds, and returns (name, keep-alive object)
    of the first one every rank could bind, set up and self-test (pf_debug_exchange: all-to-all, row-range all-to-all and
    all-reduce with a known pattern); raises RuntimeError on every rank if none works.  `kinds`: name -> class, for tests.
    `votes`: a list that receives one record per vote taken -- {"kind", "step": "bind" | "setup" | "selftest", "here": this
    rank's answer, "all": the outcome over all ranks} -- the same sequence on every rank (bench.py prints rank 0's)."""
    kinds = kinds or KINDS
    votes = votes if votes is not None else []

    def vote(kind, step, here):
        everyone = _vote(dist, torch, here, device)
        votes.append({"kind": kind, "step": step, "here": bool(here), "all": bool(everyone)})
        return everyone

    order = [preferred] + [k for k in kinds if k != preferred]
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
    rank = dist.get_rank()
    for name in order:
        k = kinds[name](f, dist, torch, device)
        try:
            local = bool(k.can_bind())
        except Exception as e:  # noqa: BLE001
            log(f"[rank {rank}] exchange '{name}': local check raised {e!r}")
            local = False
        if not vote(name, "bind", local):       # nobody has touched a communicator of this kind yet
            log(f"[rank {rank}] exchange '{name}' cannot be bound on every rank (here: {local})")
            continue
        try:
            ok = bool(k.setup())
        except Exception as e:  # noqa: BLE001
            log(f"[rank {rank}] exchange '{name}': set-up raised {e!r}")
            ok = False
        if vote(name, "setup", ok):
            try:
                # no more than a field of this context holds (three fields in the exchange buffers; small grids in tests)
                cap = C.c_size_t()
                f.L.pf_exchange_buffers(f.h, None, None, C.byref(cap))
                per_peer = min(selftest_bytes, (cap.value // 3 // dist.get_world_size()) // 8 * 8)
                ok = f.L.pf_debug_exchange(f.h, per_peer) == 0
            except Exception as e:  # noqa: BLE001
                log(f"[rank {rank}] exchange '{name}': self-test raised {e!r}")
                ok = False
            if vote(name, "selftest", ok):
                return name, k
            log(f"[rank {rank}] exchange '{name}' failed its self-test on some rank (here: {ok})")
        else:
            log(f"[rank {rank}] exchange '{name}' could not be set up on every rank (here: {ok})")
        try:
            k.release()
        except Exception as e:  # noqa: BLE001
            log(f"[rank {rank}] exchange '{name}': release raised {e!r}")
    raise RuntimeError("no working multi-GPU exchange")


def install_exchange(f, dist, torch, kind: str = "rccl"):
    """one kind, no negotiation (tests of a single kind on a one-rank group); returns the keep-alive object"""
    k = KINDS[kind](f, dist, torch, "cuda")
    if not k.can_bind() or not k.setup():
        raise RuntimeError(f"exchange '{kind}' unavailable")
    return k
