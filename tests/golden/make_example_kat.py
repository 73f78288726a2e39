"""Writes tests/golden/example_kat.json from the log of the reference's example run (example/log: V5.1 with the Makefile's default
flags, 128^3 on 4 tasks, box 500 Mpc/h, seed 486604, Eisenstein & Hu P(k) with the logged normalisation, Rayleigh-sampled
amplitudes, seven radii): parameters, logged radii / variances / sigmas and the collapsed-cell count.  (The FmaxPDF file kept beside
the log belongs to another run -- a CAMBTable run like example/parameter_file, 741 412 collapsed cells -- and is not used.)
Data only.  Needs /root/reference.

    python tests/golden/make_example_kat.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    log = open("/root/reference/example/log").read().splitlines()
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
    val = lambda key: [l.split()[-1] for l in log if l.startswith(key)][0]
    assert len(radii) == 7 and len(sig) == 7 and val("FixedIC") == "0" and val("PairedIC") == "0" and val("FileWithInputSpectrum") == "no"
    assert any("Radiation is not included" in l for l in log)
    kat = {"_provenance": "Reference's committed example/log (V5.1, default Makefile flags, 4 MPI tasks).  Data only.",
           "params": {"GridSize": 128, "BoxSize_h100": 500.0, "RandomSeed": int(val("RandomSeed")), "Omega0": float(val("Omega0")),
                      "OmegaLambda": float(val("OmegaLambda")), "OmegaBaryon": float(val("OmegaBaryon")), "Hubble100": float(val("Hubble100")),
                      "Sigma8": float(val("Sigma8")), "PrimordialIndex": float(val("PrimordialIndex")), "FixedIC": 0},
           "PkNorm": pk, "radii_Mpc": radii, "variance": var, "expected_sigma": [s[0] for s in sig], "computed_sigma": [s[1] for s in sig],
           "collapsed": coll}
    json.dump(kat, open(os.path.join(HERE, "example_kat.json"), "w"))
    print("wrote example_kat.json:", coll, kat["params"], pk)


if __name__ == "__main__":
    main()
