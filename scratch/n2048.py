"""launch check of the N = 2048 kernels (BASELINE config 5 geometry): one rank of 16 with the exchange stubbed out,
so the numbers are meaningless but every kernel of the path runs with its 2048-point configuration"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth, _lib
for fb in (4, 8):
    f = api.Fmax(2048, rank=3, nranks=16, field_bytes=fb, timing=True)
    cb1 = _lib.ALLTOALL_FN(lambda user, s, r, b, st: 0)
    cb2 = _lib.ALLREDUCE_FN(lambda user, buf, cnt, u, st: 0)
    f.L.pf_set_exchange(f.h, cb1, None); f.L.pf_set_allreduce(f.h, cb2, None)
    f.synth_density(synth.SEED, 2.5, -2.0)
    x, y = synth.invgrow_table("lcdm")
    f.set_invgrow(x, y)
    t0 = time.perf_counter()
    tv = f.compute_fmax(np.array([8.0, 1.0, 0.0]), do_lpt=True)
    f.synchronize()
    print("field_bytes", fb, "device GB %.1f" % (f.device_bytes / 1e9), "time %.2f s" % (time.perf_counter() - t0), "tv", tv)
    print("  " + " | ".join("%s %.2f ms" % (k["name"], k["total_ms"] / k["launches"]) for k in f.kernel_stats() if k["launches"]))
    pdf = f.Fmax_PDF()
    assert int(pdf.sum()) == 2048 * 2048 * 128
    f.close()
print("ok")
