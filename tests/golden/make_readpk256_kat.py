"""Writes tests/golden/readpk256_kat.json from a fourth run the reference commits with its outputs:
tests/only_HMF_tests/READ_PK_TABLE_and_SCALE_DEP (V5.0, -DREAD_PK_TABLE -DSCALE_DEPENDENT; 256^3, box 256 Mpc/h, seed 486604,
FileWithInputSpectrum CAMBTable, Sigma8 0 = trust the table, fixed-amplitude initial conditions, ten radii).  Data only:
parameters, the z = 0 CAMB table the run read (Custom_scale_dep/custom_pk_149.dat: k [h/Mpc], P [(Mpc/h)^3], 503 rows -- the knots
of SPLINE[SP_PK], src/cosmo.c:1290-1330), logged radii / variances / sigmas / Sigma8, collapsed count, the 210-bin Fmax PDF and the
linear growth column of the run's scaledep.out (identical in all ten k bins: the growth of this run's tables does not depend on
scale, so SPLINE_INVGROW[ismooth] is the same table for every radius, src/initialization.c:1596-1708).  Needs /root/reference.

    python tests/golden/make_readpk256_kat.py
"""
import json
import os
import re

D = "/root/reference/tests/only_HMF_tests/READ_PK_TABLE_and_SCALE_DEP/"
HERE = os.path.dirname(os.path.abspath(__file__))
FLAG = "LCDM_READ_PK_TABLE_and_SCALE_DEP"


def main():
    log = open(D + "log_READ_PK_and_SCALE_DEP").read().splitlines()
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
    s8 = [float(re.search(r"Sigma8=([0-9.]+)", l).group(1)) for l in log if "Normalization of the provided P(k)" in l][0]
    pdf = [int(l.split()[2]) for l in open(D + f"pinocchio.{FLAG}.FmaxPDF.out") if not l.startswith("#")]
    assert len(radii) == 10 and len(sig) == 10 and len(pdf) == 210 and sum(pdf) == 256 ** 3
    assert any("non-random modules" in l for l in log) and any("read from CAMB files" in l for l in log)
    reds = [l.split() for l in open(D + "CAMB_redshifts/redshifts_file.txt") if l.strip()]
    assert len(reds) == 150 and float(reds[-1][1]) == 0.0          # the last CAMB file is z = 0 (src/cosmo.c:1268-1272)
    pk = [[float(x) for x in l.split()] for l in open(D + "Custom_scale_dep/custom_pk_149.dat") if l.strip()]
    sd = [[float(x) for x in l.split()] for l in open(D + f"pinocchio.{FLAG}.scaledep.out") if not l.startswith("#")]
    assert all(len(set(r[1:11])) == 1 for r in sd)                   # ten k bins, one growth: scale independent
    kat = {"_provenance": "Reference's committed run tests/only_HMF_tests/READ_PK_TABLE_and_SCALE_DEP (V5.0).  Data only.",
           "params": {"GridSize": 256, "BoxSize_h100": 256.0, "RandomSeed": 486604, "Omega0": 0.3175, "OmegaLambda": 0.6825, "OmegaBaryon": 0.049,
                      "Hubble100": 0.6711, "Sigma8": 0.0, "PrimordialIndex": 0.9624, "FixedIC": 1},
           "radii_Mpc": radii, "variance": var, "expected_sigma": [s[0] for s in sig], "computed_sigma": [s[1] for s in sig],
           "Sigma8_of_the_table": s8, "collapsed": coll, "FmaxPDF": pdf,
           "camb_z0_k_hMpc_P": pk, "scaledep_a_D1": [[r[0], r[1]] for r in sd]}
    json.dump(kat, open(os.path.join(HERE, "readpk256_kat.json"), "w"))
    print("wrote readpk256_kat.json:", coll, s8, len(pk))


if __name__ == "__main__":
    main()
