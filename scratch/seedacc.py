"""accuracy of the hardware seeds and of shortened refinements (pf_debug_math taps 9-12)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from pinocchio_amd import api
ld = np.longdouble
rng = np.random.default_rng(5)
n = 200000
a = rng.standard_normal(n) * 10.0 ** rng.integers(-100, 100, n)
b = rng.standard_normal(n) * 10.0 ** rng.integers(-100, 100, n)
x = np.abs(a)
dp = C.POINTER(C.c_double)
def run(f, which, a, b=None):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b if b is not None else np.ones_like(a)); out = np.empty_like(a)
    f._chk(f.L.pf_debug_math(f.h, which, a.ctypes.data_as(dp), b.ctypes.data_as(dp), len(a), out.ctypes.data_as(dp)))
    return out
def ulps(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got.astype(ld) - want.astype(ld)) / np.spacing(np.abs(want))
with api.Fmax(128) as f:
    r = run(f, 11, x); print('rcp seed   max rel err 2^%.1f' % np.log2(np.max(np.abs(r.astype(ld) * x.astype(ld) - 1))))
    r = run(f, 12, x); print('rsq seed   max rel err 2^%.1f' % np.log2(np.max(np.abs(r.astype(ld) ** 2 * x.astype(ld) - 1)) / 2))
    for w, name, want in ((0, 'div, two Newton steps (current)', a.astype(ld) / b.astype(ld)), (9, 'div, one Newton step', a.astype(ld) / b.astype(ld)),
                          (1, 'sqrt, two iterations (current)', np.sqrt(x.astype(ld))), (10, 'sqrt, one iteration', np.sqrt(x.astype(ld)))):
        got = run(f, w, a if w in (0, 9) else x, b)
        u = ulps(got, want.astype(np.float64))
        exact = want.astype(np.float64)
        print('%-34s max %.2f ulp, not correctly rounded: %.2e of the cases' % (name, u.max(), np.mean(got != exact)))
