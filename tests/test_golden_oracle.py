"""The committed golden vectors are reproduced by the oracle (guards the fixture
against drifting away from its generator)."""
import os

import numpy as np

import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_reproduces_golden():
    z = np.load(os.path.join(GOLD, "sweep_n16.npz"))
    o = oracle_lib.Oracle(int(z["n"]), 1)
    o.set_density(z["dk"])
    o.set_invgrow(z["spline_x"], z["spline_y"])
    o.set_growth(z["growth"])
    tv = o.compute_fmax(z["radii"], do_lpt=True)
    p = o.products()
    assert np.array_equal(tv, z["true_variance"])
    for k in ("Fmax", "Rmax", "Vel", "Vel_2LPT", "Vel_3LPT_1", "Vel_3LPT_2"):
        assert np.array_equal(p[k], z[k]), k
    assert np.array_equal(o.fmax_pdf(), z["pdf"])
