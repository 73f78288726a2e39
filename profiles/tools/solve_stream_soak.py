"""Soak of the solve stream: many consecutive sweeps + displacements of one context (changing densities and ladders of two to eight
radii, fp64 and fp32 fields, 32^3 .. 128^3) with the solve beside the next z-pass, every result compared by checksum with the first
occurrence of the same (density, ladder) -- four of them made with every kernel in line.  An event missing between the streams
would show as a mismatch.  Run from the repo root on the GPU box: python3 profiles/tools/solve_stream_soak.py"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.getcwd())
from pinocchio_amd import api, synth
x, y = synth.invgrow_table("lcdm")
bad = 0
for fb in (8, 4):
    for n in (32, 64, 128):
        ref = {}
        for mode in ("0", "1"):
            os.environ["PF_SOLVE_BESIDE_Z"] = mode
            with api.Fmax(n, field_bytes=fb) as f:
                f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
                for it in range(40 if mode == "1" else 4):
                    seed = 100 + (it % 4)
                    f.set_density(synth.make_density(n, seed=seed))
                    radii = np.array([n / 16.0, n / 40.0, 2.0, 1.5, 0.9, 0.6, 0.3, 0.0])[(it % 3):]
                    tv = f.compute_fmax(radii, do_lpt=True)
                    p = f.products()
                    key = (seed, len(radii))
                    sig = (zlib.crc32(p["Fmax"].tobytes()), zlib.crc32(p["Rmax"].tobytes()), zlib.crc32(p["Vel_3LPT_2"].tobytes()), tuple(tv))
                    if mode == "0" or key not in ref:
                        ref.setdefault(key, sig)
                    elif ref[key] != sig:
                        bad += 1; print("MISMATCH", fb, n, it, key)
        print("fb", fb, "n", n, "keys", len(ref), "bad so far", bad, flush=True)
print("SOAK", "FAILED" if bad else "OK")
