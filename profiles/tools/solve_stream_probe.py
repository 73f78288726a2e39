"""the bench step (12 radii + 3LPT at 1024^3 unless PROBE_N says otherwise; PROBE_FB=4: fp32 fields) with the solve of the sweep on
its own stream beside the z-pass of the next radius (PF_SOLVE_BESIDE_Z=1, the default) or in line (0): ms per step over
PROBE_STEPS steps and a checksum of Fmax / Rmax / one displacement column, so that the two orders can be seen to agree bit for bit"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pinocchio_amd import api, synth  # noqa: E402

n = int(os.environ.get("PROBE_N", "1024"))
steps = int(os.environ.get("PROBE_STEPS", "4"))
fb = int(os.environ.get("PROBE_FB", "8"))
x, y = synth.invgrow_table("lcdm")
radii = synth.radii_ladder(12)
timing = bool(int(os.environ.get("PROBE_TIMING", "0")))
with api.Fmax(n, field_bytes=fb, timing=timing) as f:
    f.synth_density(synth.SEED, 2.5, -2.0)
    f.set_invgrow(x, y)
    f.compute_fmax(radii, do_lpt=True)
    f.synchronize()
    if timing:
        f.reset_kernel_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        tv = f.compute_fmax(radii, do_lpt=True)
    f.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fm = f.block("FMAX")
    rm = f.block("RMAX")
    zel = f.block("ZEL ")
    print("PF_SOLVE_BESIDE_Z=%s  %.1f ms per step  crc Fmax %08x Rmax %08x Zel %08x  tv[-1] %.15g tv[0] %.15g" % (
        os.environ.get("PF_SOLVE_BESIDE_Z", "-"), 1e3 * dt, zlib.crc32(fm.tobytes()), zlib.crc32(rm.tobytes()), zlib.crc32(zel.tobytes()),
        tv[-1], tv[0]))
    if timing:
        for k in f.kernel_stats():
            if k["launches"]:
                print("   %-26s %4d launches  %8.2f ms per step" % (k["name"], k["launches"], k["total_ms"] / steps))
