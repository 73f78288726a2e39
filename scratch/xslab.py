"""x-pass on a 1/8 slab (x stride 1 MB instead of 8.4 MB): TLB-reach experiment; exchange stubbed out (timing only)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth, _lib
P = int(os.environ.get("XP", "8"))
f = api.Fmax(1024, rank=0, nranks=P, timing=True)
cb1 = _lib.ALLTOALL_FN(lambda user, s, r, b, st: 0)
cb2 = _lib.ALLREDUCE_FN(lambda user, buf, cnt, u, st: 0)
f.L.pf_set_exchange(f.h, cb1, None); f.L.pf_set_allreduce(f.h, cb2, None)
f.synth_density(synth.SEED, 2.5, -2.0)
f.compute_second_derivatives(0.0)
f.reset_kernel_stats()
for _ in range(5):
    f.compute_second_derivatives(0.0)
f.synchronize()
print("P", P, " | ".join("%s %.2f ms %.0f GB/s" % (k["name"], k["total_ms"] / k["launches"], k["alg_bytes"] / k["total_ms"] / 1e6)
                           for k in f.kernel_stats() if k["launches"]))
