"""Thread scaling of the CPU oracle (the cpu_baseline leg of bench.py) on this host: n^3 box, two radii + the 3LPT part"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from pinocchio_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = [int(t) for t in sys.argv[2:]] or [8, 32, 64, 128, 256]
dk = synth.philox_density(n, synth.SEED, 2.5, -2.0)
x, y = synth.invgrow_table("lcdm")
full = synth.radii_ladder(12)
radii = np.array([full[6], 0.0])
print("host threads:", os.cpu_count(), flush=True)
for t in threads:
    if t > (os.cpu_count() or 1):
        continue
    o = oracle_lib.Oracle(n, t)
    o.set_density(dk)
    o.set_invgrow(x, y)
    o.set_growth(synth.growth_multipliers())
    t0 = time.perf_counter()
    o.compute_fmax(radii, do_lpt=True)
    dt = time.perf_counter() - t0
    tm = o.timers()
    print(f"n={n} threads={t:4d}  total {dt:7.2f} s  deriv {tm['deriv']:6.2f}  fft {tm['fft']:6.2f}  coll {tm['coll']:6.2f}  lpt {tm['lpt']:6.2f}", flush=True)
    o.close()
