"""Writes tests/golden/mg256_kat.json from the run the reference commits under tests/only_HMF_tests/MOD_GRAV_and_SCALE_DEP
(V5.0, 256^3, box 256 Mpc/h, seed 486604, fixed-amplitude initial conditions, E&H P(k) without baryons, Omega0 = 0.3,
-DTABULATED_CT -DELL_SNG -DMOD_GRAV_FR -DFR0=1.e-8 -DSCALE_DEPENDENT on 4 tasks): the only reference-held output of the
collapse-time tables, the Nadkarni-Ghosh & Singhal ODE model and the f(R) force modification.
Data only: parameters, logged radii / variances / sigmas, collapsed count, 210-bin Fmax PDF, the first rows of the linear growth
table and its logarithmic slope there (the model starts at a = 1e-5, below the table: my_spline_eval extrapolates linearly in
log-log, src/cosmo.c:2016-2027; the file prints d ln D / d ln a = 1 at its first row).
Needs /root/reference; run once.

    python tests/golden/make_mg256_kat.py
"""
import json
import os
import re

ROOT = "/root/reference/tests/only_HMF_tests/MOD_GRAV_and_SCALE_DEP/"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    log = open(ROOT + "log_MOD_GRAV_and_SCALE_DEP").read().splitlines()
    radii, var, sig = [], [], []
    for l in log:
        m = re.match(r"\s+\d+\)\s+Radius=\s*([0-9.]+), Variance=\s*([0-9.]+)", l)
        if m:
            radii.append(float(m.group(1)))
            var.append(float(m.group(2)))
        m = re.search(r"expected sigma:\s*([0-9.]+), computed sigma:\s*([0-9.]+)", l)
        if m:
            sig.append((float(m.group(1)), float(m.group(2))))
    coll = [int(re.search(r"to z=0: (\d+)", l).group(1)) for l in log if "Number of collapsed particles" in l][0]
    pk = [float(re.search(r"spectrum: ([0-9.e+]+)", l).group(1)) for l in log if "Normalization constant for the power spectrum" in l][0]
    fr0 = [float(re.search(r"f_R0=\s*([0-9.e+-]+)", l).group(1)) for l in log if "Hu-Sawicki" in l][0]
    assert any("will be tabulated" in l for l in log) and any("non-random modules" in l for l in log)
    pdf = [int(l.split()[2]) for l in open(ROOT + "pinocchio.LCDM_MOD_GRAV_and_SCALE_DEP.FmaxPDF.out") if not l.startswith("#")]
    sd = [l.split() for l in open(ROOT + "pinocchio.LCDM_MOD_GRAV_and_SCALE_DEP.scaledep.out") if not l.startswith("#")]
    assert len(radii) == 9 and len(sig) == 9 and len(pdf) == 210 and sum(pdf) == 256 ** 3
    # at the first rows every k-bin of the table holds the same (general-relativity) growth
    assert all(len(set(r[1:11])) == 1 for r in sd[:4])
    kat = {"_provenance": "Reference's committed run tests/only_HMF_tests/MOD_GRAV_and_SCALE_DEP (V5.0, TABULATED_CT + ELL_SNG + MOD_GRAV_FR, FR0 = 1e-8, "
                          "SCALE_DEPENDENT, 4 tasks).  Data only.",
           "params": {"GridSize": 256, "BoxSize_h100": 256.0, "RandomSeed": 486604, "Omega0": 0.3, "OmegaLambda": 0.7, "OmegaBaryon": 0.0,
                      "Hubble100": 0.70, "Sigma8": 0.8, "PrimordialIndex": 0.96, "FixedIC": 1, "FR0": fr0},
           "PkNorm": pk, "radii_Mpc": radii, "variance": var, "expected_sigma": [s[0] for s in sig], "computed_sigma": [s[1] for s in sig],
           "collapsed": coll, "FmaxPDF": pdf, "growth_first_rows_a_D1": [[float(r[0]), float(r[1])] for r in sd[:2]],
           "dlnD_dlna_first_row": float(sd[0][41])}
    json.dump(kat, open(os.path.join(HERE, "mg256_kat.json"), "w"))
    print("wrote mg256_kat.json:", coll, fr0, pk)


if __name__ == "__main__":
    main()
