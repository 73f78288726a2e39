"""Writes tests/golden/hmf256_kat.json from two more runs the reference commits with their outputs
(tests/only_HMF_tests/{RECOMPUTE_DISPLACEMENTS_LCDM, SCALE_DEP_LCDM}: V5.0, 256^3, box 256 Mpc/h, seed 486604, E&H P(k) with
OmegaBaryon = 0, Omega0 = 0.3, sigma8 = 0.8, nine radii, fixed-amplitude initial conditions ("non-random modules of the Fourier
modes" = params.FixedIC, src/GenIC.c:375); the second one a -DSCALE_DEPENDENT build).  Data only: parameters, logged radii /
variances / sigmas, collapsed count, the 210-bin Fmax PDF, and the linear growth column of the run's scaledep.out (the table
behind SPLINE_INVGROW[ismooth], src/initialization.c:1704-1708; 6 significant digits).  Needs /root/reference; run once.

    python tests/golden/make_hmf256_kat.py
"""
import json
import os
import re

ROOT = "/root/reference/tests/only_HMF_tests/"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse(d, logname, flag):
    log = open(ROOT + d + "/" + logname).read().splitlines()
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
    pdf = [int(l.split()[2]) for l in open(ROOT + d + f"/pinocchio.{flag}.FmaxPDF.out") if not l.startswith("#")]
    assert len(radii) == 9 and len(sig) == 9 and len(pdf) == 210 and sum(pdf) == 256 ** 3
    assert any("non-random modules" in l for l in log)
    return dict(radii_Mpc=radii, variance=var, expected_sigma=[s[0] for s in sig], computed_sigma=[s[1] for s in sig], collapsed=coll,
                PkNorm=pk, FmaxPDF=pdf)


def main():
    a = parse("RECOMPUTE_DISPLACEMENTS_LCDM", "log_RECOMPUTE", "LCDM_RECOMPUTE")
    b = parse("SCALE_DEP_LCDM", "log_SCALE_DEP", "LCDM_SCALE_DEP")
    assert a == b  # in LCDM the scale-dependent build must (and does) give the very same numbers
    sd = [l.split() for l in open(ROOT + "SCALE_DEP_LCDM/pinocchio.LCDM_SCALE_DEP.scaledep.out") if not l.startswith("#")]
    kat = {"_provenance": "Reference's committed runs tests/only_HMF_tests/RECOMPUTE_DISPLACEMENTS_LCDM and SCALE_DEP_LCDM (V5.0; the two logs and "
                          "Fmax PDFs are identical, as they must be in LCDM).  Data only.",
           "params": {"GridSize": 256, "BoxSize_h100": 256.0, "RandomSeed": 486604, "Omega0": 0.3, "OmegaLambda": 0.7, "OmegaBaryon": 0.0,
                      "Hubble100": 0.70, "Sigma8": 0.8, "PrimordialIndex": 0.96, "FixedIC": 1},
           **a,
           "scaledep_a_D1": [[float(r[0]), float(r[1])] for r in sd]}
    json.dump(kat, open(os.path.join(HERE, "hmf256_kat.json"), "w"))
    print("wrote hmf256_kat.json:", a["collapsed"], a["computed_sigma"])


if __name__ == "__main__":
    main()
