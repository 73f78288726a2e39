"""How close the stored fp32 Fmax of the HIP path is to the oracle's, cell by cell (both flavours of the solve): fractions of cells
that differ at all, by more than 1 and 2 fp32 ulp, and the largest difference; n^3 box, all twelve radii of the bench ladder."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from pinocchio_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dk = synth.philox_density(n, synth.SEED, 2.5, -2.0)
x, y = synth.invgrow_table("lcdm")
radii = synth.radii_ladder(12) * n / 1024.0 if len(sys.argv) > 2 else synth.radii_ladder(12)
o = oracle_lib.Oracle(n, 16)
o.set_density(dk); o.set_invgrow(x, y); o.set_growth(synth.growth_multipliers())
o.compute_fmax(radii, do_lpt=False)
want = o.products()["Fmax"]
rwant = o.products()["Rmax"]
for flavour in ("fast", "exact"):
    if flavour == "exact":
        os.environ["PF_EXACT_LIBM"] = "1"
    else:
        os.environ.pop("PF_EXACT_LIBM", None)
    with api.Fmax(n) as f:
        f.set_density(dk); f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
        f.compute_fmax(radii, do_lpt=False)
        p = f.products()
    got = p["Fmax"]
    ulp = np.spacing(np.maximum(np.abs(want), 1.0).astype(np.float32)).astype(np.float64)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    print(f"n={n} {flavour:5s}: differ at all {np.mean(d > 0):.2e}, > 1 ulp {np.mean(d > ulp):.2e}, > 2 ulp {np.mean(d > 2 * ulp):.2e} "
          f"({int(np.sum(d > 2 * ulp))} cells), > 1e-3 abs {int(np.sum(d > 1e-3))} cells, max {d.max():.3e}; Rmax differs on {np.mean(p['Rmax'] != rwant):.2e}", flush=True)
