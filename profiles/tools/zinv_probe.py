"""invariant z-pass and solve on a pruned (R = 16 cells) and an unpruned (R = 1) radius at 1024^3: ms per launch, and a checksum of
Fmax / Rmax so that variants can be seen to agree bit for bit"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pinocchio_amd import api, synth  # noqa: E402

n = int(os.environ.get("PROBE_N", "1024"))
x, y = synth.invgrow_table("lcdm")
with api.Fmax(n, timing=True) as f:
    f.synth_density(synth.SEED, 2.5, -2.0)
    f.set_invgrow(x, y)
    for rs in ([16.0, 0.0], [1.0, 0.0]):
        f.sweep(np.array(rs))
        f.reset_kernel_stats()
        for _ in range(3):
            f.sweep(np.array(rs))
        s = {k["name"]: k["total_ms"] / 3 for k in f.kernel_stats()}
        print("R=%4.1f  zinv %.2f  solve_inv %.2f  z6 %.2f  solve6 %.2f" % (rs[0], s["zpass_c2r_hess_6to3inv"], s["collapse_inv"], s["zpass_c2r_hess_6"], s["collapse"]))
    if n <= 256:
        p = f.products()
        print("crc Fmax %08x Rmax %08x" % (zlib.crc32(p["Fmax"].tobytes()), zlib.crc32(p["Rmax"].tobytes())))
    else:
        fm = f.block("FMAX")
        print("crc Fmax %08x" % zlib.crc32(fm.tobytes()))
