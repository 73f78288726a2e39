#!/usr/bin/env python3
"""Per-radius kernel times of the sweep at 1024^3 (or --n): one pf_sweep per prefix of the ladder is not needed -- the
kernel classes are timed per radius by running the sweep on ONE radius at a time is not the same path (last radius keeps six
components), so this script runs the full ladder and differences the kernel statistics of ladders of growing length.
Simpler and exact: a sweep over [R_i, R_last] for every i; the R_last part is subtracted."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pinocchio_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
radii = synth.radii_ladder(12)
x, y = synth.invgrow_table("lcdm")
with api.Fmax(a.n, timing=True) as f:
    f.synth_density(synth.SEED, 2.5, -2.0)
    f.set_invgrow(x, y)
    f.sweep(radii)

    def run(rs):
        f.reset_kernel_stats()
        for _ in range(a.reps):
            f.sweep(np.array(rs))
        return {k["name"]: k["total_ms"] / a.reps for k in f.kernel_stats()}

    base = run([0.0])
    print("%6s %10s %10s %10s %10s" % ("R", "xpass", "ypass", "zpass_inv", "solve_inv"))
    for r in radii[:-1]:
        s = run([r, 0.0])
        print("%6.2f %10.3f %10.3f %10.3f %10.3f" % (r, s["xpass_hess_1to3"] - base["xpass_hess_1to3"], s["ypass_hess_3to6"] - base["ypass_hess_3to6"],
                                                     s.get("zpass_c2r_hess_6to3inv", 0.0), s.get("collapse_inv", 0.0)))
    print("%6.2f %10.3f %10.3f %10.3f %10.3f   (six components)" % (0.0, base["xpass_hess_1to3"], base["ypass_hess_3to6"], base["zpass_c2r_hess_6"], base["collapse"]))
