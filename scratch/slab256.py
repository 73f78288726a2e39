"""slab path at a larger size than the suite uses: 256^3 on 4 and 8 virtual ranks (in-process fabric) against one rank, bitwise"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pinocchio_amd import api, synth
import test_gpu_multirank as T
n = 256
x, y = synth.invgrow_table("lcdm")
g = synth.growth_multipliers()
radii = synth.radii_ladder(12)[[0, 2, 4, 6, 9, 11]] * (n / 1024.0) * 4   # 16 .. 0 cells at 256: band-limited ones first
print("radii", radii)
dk = synth.philox_density(n, synth.SEED, 2.5, -2.0)   # the same host array for every decomposition
with api.Fmax(n) as f1:
    f1.set_density(dk)
    f1.set_invgrow(x, y); f1.set_growth(g)
    tv1 = f1.compute_fmax(radii, do_lpt=True)
    p1 = f1.products()
for P in (4, 8):
    nxl = n // P
    def body(f, r):
        f.set_density(dk[r * nxl:(r + 1) * nxl])
        f.set_invgrow(x, y); f.set_growth(g)
        tv = f.compute_fmax(radii, do_lpt=True)
        return tv, f.products()
    t0 = time.time()
    res = T.run_ranks(api, n, P, body)
    ok = True
    for r in range(P):
        tv, p = res[r]
        sl = slice(r * nxl, (r + 1) * nxl)
        ok &= bool(np.allclose(tv, tv1, rtol=1e-12))
        d = np.abs(p["Fmax"].astype(np.float64) - p1["Fmax"][sl].astype(np.float64))
        same = float(np.mean(p["Fmax"] == p1["Fmax"][sl]))
        ok &= same > 0.999 and bool(np.array_equal(p["Vel"], p1["Vel"][sl]) or np.max(np.abs(p["Vel"] - p1["Vel"][sl])) < 1e-6 * np.max(np.abs(p1["Vel"])))
        print("P", P, "rank", r, "Fmax identical fraction %.6f" % same, "max |dF| %.2e" % d.max(), "Rmax equal %.6f" % float(np.mean(p["Rmax"] == p1["Rmax"][sl])))
    print("P", P, "ok" if ok else "MISMATCH", "%.1f s" % (time.time() - t0))
