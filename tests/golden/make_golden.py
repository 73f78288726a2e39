"""Writes tests/golden/sweep_n16.npz: inputs and expected outputs of the whole
path at N=16, produced by the CPU oracle (oracle/pf_oracle.c).  Data only.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib  # noqa: E402
from pinocchio_amd import synth  # noqa: E402


def main():
    n = 16
    dk = synth.make_density(n, seed=synth.SEED)
    radii = np.array([3.0, 2.0, 1.0, 0.5, 0.0])
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    o = oracle_lib.Oracle(n, 1)  # single thread: bit-reproducible TrueVariance
    o.set_density(dk)
    o.set_invgrow(x, y)
    o.set_growth(g)
    tv = o.compute_fmax(radii, do_lpt=True)
    p = o.products()
    np.savez_compressed(
        os.path.join(HERE, "sweep_n16.npz"), n=n, dk=dk, radii=radii, spline_x=x, spline_y=y, growth=g,
        true_variance=tv, Fmax=p["Fmax"], Rmax=p["Rmax"], Vel=p["Vel"], Vel_2LPT=p["Vel_2LPT"],
        Vel_3LPT_1=p["Vel_3LPT_1"], Vel_3LPT_2=p["Vel_3LPT_2"], pdf=o.fmax_pdf())


if __name__ == "__main__":
    main()
