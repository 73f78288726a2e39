"""The fp32-field contract from data (round 6): the distribution of |Fmax(fp32 fields) - Fmax(reference)| on the cells
with Fmax >= 0.5, per number of radii swept -- against the oracle at 256^3 (fed with the fp32-rounded delta(k) the device holds AND
with the fp64 one), and against the fp64-field run of the same modes at 1024^3.  Run on the GPU box from the repository root:
    python3 profiles/tools/fp32_contract.py > gpurun_out/r06/fp32_contract.json
The oracle is the checker here (this is a measurement script of the test infrastructure, not a product path)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from pinocchio_amd import _lib, api, synth  # noqa: E402

QS = [0.5, 0.9, 0.99, 0.999, 0.9999]


def dist_of(got, want):
    sel = want >= 0.5
    d = np.abs(got[sel].astype(np.float64) - want[sel].astype(np.float64))
    q = np.quantile(d, QS)
    return {"cells_F_ge_0.5": int(sel.sum()), "quantiles": {str(k): float(v) for k, v in zip(QS, q)}, "max": float(d.max()),
            "fraction_le_1e-4": float(np.mean(d <= 1e-4)), "fraction_le_1e-3": float(np.mean(d <= 1e-3)),
            "fraction_exactly_equal": float(np.mean(d == 0.0))}


def main():
    out = {"kernel_source_sha": _lib.source_sha(), "what": "abs(Fmax(fp32 fields) - Fmax(reference)) on cells with reference Fmax >= 0.5"}
    x, y = synth.invgrow_table("lcdm")
    # ---- 256^3 against the oracle, 1 .. 12 radii of the bench ladder (scaled to the box), sigma(R = 0) = 2.5
    n = 256
    dk = synth.philox_density(n, synth.SEED, 2.5, -2.0)
    dk32 = dk.astype(np.complex64).astype(np.complex128)          # what a context with fp32 fields holds
    ladder = synth.radii_ladder(12) * (n / 1024.0)
    ladder[-1] = 0.0
    rows = []
    for ns in (1, 3, 6, 12):
        radii = ladder[-ns:]
        with api.Fmax(n, field_bytes=4) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.sweep(radii)
            got = f.block("FMAX")
        with api.Fmax(n) as f:
            f.set_density(dk)
            f.set_invgrow(x, y)
            f.sweep(radii)
            got64 = f.block("FMAX")
        row = {"n": n, "radii": ns}
        for name, spec in (("oracle_on_fp64_density", dk), ("oracle_on_fp32_rounded_density", dk32)):
            o = oracle_lib.Oracle(n, 0)
            o.set_density(spec)
            o.set_invgrow(x, y)
            o.compute_fmax(radii, do_lpt=False)
            want = o.products()["Fmax"].reshape(-1)
            row["fp32_fields_vs_" + name] = dist_of(got, want)
            if name == "oracle_on_fp64_density":
                row["fp64_fields_vs_oracle"] = dist_of(got64, want)
            del o
        row["fp32_fields_vs_fp64_fields"] = dist_of(got, got64)
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    out["n256_vs_oracle"] = rows
    # ---- 1024^3: fp32 fields against fp64 fields of the same modes, 3 and 12 radii
    n = 1024
    rows = []
    for ns in (3, 12):
        radii = synth.radii_ladder(12)[[2, 8, 11]] if ns == 3 else synth.radii_ladder(12)
        with api.Fmax(n) as f:
            f.synth_density(synth.SEED, 2.5, -2.0)
            f.set_invgrow(x, y)
            f.sweep(radii)
            fm64 = f.block("FMAX")
        with api.Fmax(n, field_bytes=4) as f:
            f.synth_density(synth.SEED, 2.5, -2.0)
            f.set_invgrow(x, y)
            f.sweep(radii)
            fm = f.block("FMAX")
        row = {"n": n, "radii": ns, "fp32_fields_vs_fp64_fields": dist_of(fm, fm64)}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
        del fm, fm64
    out["n1024_vs_fp64_fields"] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
