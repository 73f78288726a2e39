"""What field does one rank of `bench.py --slab-of P` solve?  (loopback exchange: the rank's own blocks come back as every peer's)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from pinocchio_amd import api, synth
n, P = 256, 8
radii = synth.radii_ladder(12)
x, y = synth.invgrow_table("lcdm")
out = {}
for mode in ("slab", "box"):
    f = api.Fmax(n, rank=0, nranks=P if mode == "slab" else 1, device=0)
    if mode == "slab":
        f._chk(f.L.pf_set_loopback_exchange(f.h, 1 << 20))
    f.synth_density(synth.SEED, 2.5, -2.0)
    f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
    tv = f.compute_fmax(radii, do_lpt=False)
    fm = f.block("FMAX"); rm = f.block("RMAX")
    fm = np.asarray(fm).reshape(-1, n, n)
    print(mode, fm.shape, "TrueVariance", np.round(tv[[0, 5, 11]], 4), "collapsed", float((fm >= 1).mean()), "Fmax == -10:", float((fm <= -9).mean()),
          "mean |Fmax| per y mod 8:", np.round([np.abs(fm[:, j::8, :]).mean() for j in range(8)], 3))
    f.close() if hasattr(f, "close") else None
