"""collapse micro benchmark: sweep of 3 radii (no pruning at rs=0.. small), per-kernel stats"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth
n = int(os.environ.get("ZN", "1024"))
f = api.Fmax(n, timing=True)
f.synth_density(synth.SEED, 2.5, -2.0)
x, y = synth.invgrow_table("lcdm")
f.set_invgrow(x, y)
r = np.array([1.0, 0.5, 0.0])
f.sweep(r)
f.reset_kernel_stats()
tv = f.sweep(r)
f.synchronize()
print(os.environ.get("PF_COLLAPSE_WG_PER_CU", "-"), " | ".join("%s %.2f ms %.0f GB/s" % (k["name"], k["total_ms"] / k["launches"], k["alg_bytes"] / k["total_ms"] / 1e6)
                           for k in f.kernel_stats() if k["launches"]), "sigma", float(np.sqrt(tv[-1])))
