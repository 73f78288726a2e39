"""Phase probe of k_strided (scratch build `sprobe`): cycles per job and wave, x-pass 1 -> 3 and y-pass 3 -> 6 of one unpruned Hessian."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
os.environ["PINFMAX_LIB"] = os.path.join(os.getcwd(), "pinocchio_amd/csrc/build_sprobe/libpinfmax_hip_sprobe.so")
import numpy as np
from pinocchio_amd import api, synth, _lib
L = _lib.load()
L.pf_debug_st_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = 1024
with api.Fmax(n) as f:
    f.synth_density(synth.SEED, 2.5, -2.0)
    f.compute_second_derivatives(0.0); f.synchronize()
    out = (C.c_ulonglong * 256)()
    L.pf_debug_st_probe(out, 1)
    for _ in range(3):
        f.compute_second_derivatives(0.0)
    f.synchronize()
    L.pf_debug_st_probe(out, 0)
    v = np.array(out[:], dtype=np.float64).reshape(2, 16, 8)
    names = ["start", "tile wait + filter", "issue next tile", "stages", "issue stores"]
    for which, label in ((0, "x-pass 1 -> 3"), (1, "y-pass 3 -> 6")):
        a = v[which]
        wgs = a[:, 6]; jobs = a[:, 5]
        print(label, "workgroups", wgs[0], "jobs", jobs[0])
        tot = a[:, :5].sum(axis=1) / wgs
        print("  cycles per workgroup: mean %.0f (min %.0f max %.0f over waves)" % (tot.mean(), tot.min(), tot.max()))
        for i, nme in enumerate(names):
            per = a[:, i] / (wgs if i == 0 else jobs)
            print("  %-22s %7.0f cycles per %s  (waves: min %.0f max %.0f)" % (nme, per.mean(), "workgroup" if i == 0 else "job", per.min(), per.max()))
