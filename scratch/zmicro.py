"""z-pass micro benchmark: five unpruned Hessian builds at 1024^3, per-kernel stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth
n = int(os.environ.get("ZN", "1024"))
f = api.Fmax(n, timing=True)
f.synth_density(synth.SEED, 2.5, -2.0)
f.compute_second_derivatives(0.0)
f.reset_kernel_stats()
for _ in range(5):
    f.compute_second_derivatives(0.0)
f.synchronize()
out = []
for k in f.kernel_stats():
    if k["launches"]:
        out.append("%s %.2f ms %.0f GB/s" % (k["name"], k["total_ms"] / k["launches"], k["alg_bytes"] / k["total_ms"] / 1e6))
print(os.environ.get("PF_ZPASS_PERSIST", "-"), " | ".join(out))
f.close()
