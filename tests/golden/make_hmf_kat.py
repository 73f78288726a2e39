"""Writes tests/golden/hmf_validation_kat.json from the reference's committed validation run
(HMF_Validation/).  Data only: parameters, logged numbers, the Fmax histogram, two columns of the
cosmology table.  Needs /root/reference; run once in the build container.

    python tests/golden/make_hmf_kat.py
"""
import json
import os
import re

ROOT = "/root/reference/HMF_Validation/"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    log = open(ROOT + "log_RUN.txt").read().splitlines()
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
    pdf = [int(l.split()[2]) for l in open(ROOT + "pinocchio.test.FmaxPDF.out") if not l.startswith("#")]
    cos = [l.split() for l in open(ROOT + "pinocchio.test.cosmology.out") if not l.startswith("#")]
    kat = {
        "_provenance": "Reference's own committed validation run: HMF_Validation/{parameter_file, log_RUN.txt, "
                       "pinocchio.test.FmaxPDF.out, pinocchio.test.cosmology.out} (V5.1, flags -DTWO_LPT -DTHREE_LPT "
                       "-DELL_CLASSIC -DNORADIATION, 1 task, GSL 2.7.1, FFTW 3.3.10). Data only: parameters, logged "
                       "radii/variances/sigmas, collapsed count, 210-bin Fmax PDF, growth table columns 1 and 7 (6 significant digits).",
        "params": {"GridSize": 128, "BoxSize_h100": 128.0, "RandomSeed": 486604, "Omega0": 0.25, "OmegaLambda": 0.75,
                   "OmegaBaryon": 0.044, "Hubble100": 0.70, "Sigma8": 0.8, "PrimordialIndex": 0.96},
        "PkNorm": pk, "radii_Mpc": radii, "variance": var, "expected_sigma": [s[0] for s in sig],
        "computed_sigma": [s[1] for s in sig], "collapsed": coll, "FmaxPDF": pdf,
        "growth_a_D": [[float(r[0]), float(r[6])] for r in cos]}
    assert len(radii) == 9 and len(sig) == 9 and len(pdf) == 210 and sum(pdf) == 128 ** 3
    json.dump(kat, open(os.path.join(HERE, "hmf_validation_kat.json"), "w"))


if __name__ == "__main__":
    main()
