"""general (library-transform) path timing: 12 radii + 3LPT at sizes that are not a power of two, and forced at 512"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth
for n, force in ((200, 0), (384, 0), (768, 0), (512, 1), (512, 0)):
    os.environ["PF_GENERAL"] = str(force)
    f = api.Fmax(n, timing=True)
    f.synth_density(synth.SEED, 2.5, -2.0)
    x, y = synth.invgrow_table("lcdm")
    f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
    r = synth.radii_ladder(12) * n / 1024.0
    r[-1] = 0.0
    f.compute_fmax(r, do_lpt=True); f.synchronize()
    t0 = time.perf_counter(); f.compute_fmax(r, do_lpt=True); f.synchronize(); dt = time.perf_counter() - t0
    print("n %4d general %d : %.1f ms  %.3e cells/s  device %.1f GB" % (n, f.L and (force or (n & (n - 1)) != 0), 1e3 * dt, n ** 3 / dt, f.device_bytes / 1e9))
    f.close()
