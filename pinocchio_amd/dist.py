"""Multi-GPU plumbing for a Python host harness (bench.py): installs the slab
all-to-all and the small all-reduces of one rank's context.

kind="rccl":  the library's own exchange (csrc/pf_rccl.cpp: grouped
              ncclSend/ncclRecv over xGMI); torch.distributed only broadcasts
              the 128-byte ncclUniqueId.
kind="torch": every exchange is a torch.distributed all_to_all_single (all_to_all with
              per-sender sizes for the row-range form of the band-limited radii) on
              tensors that alias the library's device buffers (backend "nccl"
              = RCCL on ROCm), enqueued on the library's stream.

PyTorch is plumbing here (process group, streams), not the compute path.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class _DevMem:
    """exposes a raw device pointer through __cuda_array_interface__"""

    def __init__(self, ptr: int, nbytes: int, typestr: str = "|u1", itemsize: int = 1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 3}


def install_exchange(f, dist, torch, kind: str = "rccl"):
    """f: api.Fmax.  Returns an object that must be kept alive as long as f is used."""
    L = f.L
    if kind == "rccl":
        idbuf = (C.c_ubyte * 128)()
        if dist.get_rank() == 0:
            f._chk(L.pf_rccl_unique_id(C.cast(idbuf, C.c_void_p)))
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        idbuf = (C.c_ubyte * 128).from_buffer_copy(raw)
        f._chk(L.pf_init_rccl(f.h, C.cast(idbuf, C.c_void_p)))
        return idbuf

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
            if is_u64:
                t = torch.as_tensor(_DevMem(buf, count * 8, "<i8", 8), device="cuda")
            else:
                t = torch.as_tensor(_DevMem(buf, count * 8, "<f8", 8), device="cuda")
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                dist.all_reduce(t)
            return 0
        except Exception as e:  # noqa: BLE001
            print("all-reduce callback failed:", e, flush=True)
            return 1

    cb1 = _lib.ALLTOALL_FN(_a2a)
    cb2 = _lib.ALLREDUCE_FN(_ared)
    cb3 = _lib.ALLTOALLV_FN(_a2av)
    f._chk(L.pf_set_exchange(f.h, cb1, None))
    f._chk(L.pf_set_exchange_rows(f.h, cb3, None))
    f._chk(L.pf_set_allreduce(f.h, cb2, None))
    return (cb1, cb2, cb3)
