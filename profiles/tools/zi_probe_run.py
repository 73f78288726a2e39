"""Phase probe of the invariant z-pass (profiles/r04_notes.md, section 3f): cycle stamps at the phase boundaries of k_c2r_invariants,
wave by wave.  Needs a scratch build of the library with profiles/r04_zi_probe.diff applied to csrc/pf_fft_kernels.hip (as of commit
ac95c49) and built as the variant `zprobe` (profiles/tools/mk_variant.sh zprobe <patched file> "-I."); run on a GPU box from the repo root."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.getcwd())
os.environ["PINFMAX_LIB"] = os.path.join(os.getcwd(), "pinocchio_amd/csrc/build_zprobe/libpinfmax_hip_zprobe.so")
os.environ["PF_SOLVE_BESIDE_Z"] = "0"
import numpy as np
from pinocchio_amd import api, synth, _lib
L = _lib.load()
L.pf_debug_zi_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = 1024
with api.Fmax(n) as f:
    f.synth_density(synth.SEED, 2.5, -2.0)
    x, y = synth.invgrow_table("lcdm"); f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
    radii = synth.radii_ladder(12)
    f.compute_fmax(radii, do_lpt=False); f.synchronize()
    out = (C.c_ulonglong * 48)()
    L.pf_debug_zi_probe(out, 1)
    f.compute_fmax(radii, do_lpt=False); f.synchronize()
    L.pf_debug_zi_probe(out, 0)
    names = ["A load+write+sync", "B fold", "C stages", "D rows+barrier", "E reduction", "F barrier"]
    for l in range(6):
        v = np.array(out[8 * l:8 * l + 7], dtype=np.float64)
        it = v[6]
        tot = v[:6].sum()
        print("wave %d (component %d): %.0f cycles per row: " % (l, l, tot / it) + "  ".join("%s %.0f" % (nme.split()[0], c / it) for nme, c in zip(names, v[:6])))
