import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth, _lib
n = 1024
f = api.Fmax(n)
f.synth_density(synth.SEED, 2.5, -2.0)
x, y = synth.invgrow_table("lcdm")
f.set_invgrow(x, y)
f.sweep(np.array([1.0, 0.0]))
lay = _lib.ProductLayout(); f.L.pf_layout_3lpt(C.byref(lay))
out = np.empty((n, n, n), dtype=api.PRODUCT_DTYPE)
for rep in range(3):
    t0 = time.perf_counter(); f._chk(f.L.pf_get_products(f.h, out.ctypes.data_as(C.c_void_p), C.byref(lay))); t1 = time.perf_counter()
    print("get_products call", rep, "%.2f s %.1f GB/s" % (t1 - t0, out.nbytes / (t1 - t0) / 1e9))
b = np.empty(n ** 3, dtype=np.float32)
for rep in range(3):
    t0 = time.perf_counter(); f._chk(f.L.pf_get_block(f.h, b"FMAX", 4, b.ctypes.data_as(C.c_void_p))); t1 = time.perf_counter()
    print("get_block FMAX call", rep, "%.2f s %.1f GB/s" % (t1 - t0, b.nbytes / (t1 - t0) / 1e9))
