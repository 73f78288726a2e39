"""Writes tests/golden/lpt_analytic.json: closed-form known answers for the displacement half of the path
(second derivatives at R = 0, the 2LPT / 3LPT sources and the twelve displacement components) for density fields
made of a few plane waves.  Data only; derived here from the reference's FORMULAS, not from oracle/pf_oracle.c
and not through any FFT:

    python tests/golden/make_lpt_analytic.py

The reference holds no output of its `Vel*` fields, so this is the strongest pin available for
rows A11-A13 / f-3: the answers below are exact (Gaussian-rational Fourier coefficients), independent of grid size
as long as nothing aliases, and every step cites the line of the reference it restates.

Mode algebra.  A real periodic field on the N^3 grid is a finite sum f(x) = sum_m c[m] exp(2 pi i m.x / N),
m an integer triple, c[-m] = conj c[m].  With the reference's transforms (unnormalised r2c, c2r times 1/N^3,
src/fmax-pfft.c:191-228) the half-spectrum it holds is N^3 c[m], and the k-space multipliers act on c[m]:

  * compute_derivative(ia, ib), second derivative (src/fmax-pfft.c:366-373, greens_function :444-456):
        c[m] -> c[m] * k_a k_b / k^2 = c[m] * m_a m_b / |m|^2      (rational), k = 0 left UNTOUCHED
  * compute_derivative(ia, 0), first derivative (same lines, then the swap :379-384 = multiplication by i):
        c[m] -> i c[m] * k_a / k^2 * growth = i c[m] * (N / 2 pi) * m_a / |m|^2 * growth;   at k = 0 the swap turns the
        real coefficient into a purely imaginary one, which the c2r ignores: the mean drops out
  * sources, cell by cell (src/LPT.c:64-93), with d0..d5 = (1,1) (2,2) (3,3) (1,2) (1,3) (2,3):
        S2  = d0 d1 + d0 d2 + d1 d2 - d3^2 - d4^2 - d5^2
        S3a = 3 [ d0 (d1 d2 - d5 d5) - d3 (d3 d2 - d4 d5) + d4 (d3 d5 - d4 d1) ]
        S3b = 2 (d0 + d1 + d2) S2  -  2 sum_{a<=b} w_ab phi2_ab d_ab,  w = 1 on the diagonal, 2 off it (src/LPT.c:112-137),
        phi2_ab = second derivative (a, b) of the field whose spectrum is FFT(S2) (mean of S2 untouched: it enters all six)
  * displacements (src/LPT.c:181-228, src/fmax.c:342-345, src/fmax-pfft.c:563-631):
        Vel = D[delta] g1, Vel_2LPT = D[S2] g2, Vel_3LPT_1 = D[S3a] g3a, Vel_3LPT_2 = D[S3b] g3b  (D = first derivative)
    stored in units of N / (2 pi) so that the coefficients stay rational.
Products of fields are convolutions of the coefficient tables: exact in Fractions.  All waves have |m_i| <= 2, so the
third-order products reach |m_i| <= 6: no aliasing and no Nyquist mode for N >= 16.

A second, independent route (sympy) checks the algebra inside this script before anything is written: the Hessian is
taken by symbolic differentiation of the potential, the sources by the formulas above on those expressions, and the
displacement fields must satisfy div Psi = -g (S - <S>), curl Psi = 0 at random points to 30 digits.
"""
import json
import os
import random
from fractions import Fraction as Fr

import mpmath
import sympy

HERE = os.path.dirname(os.path.abspath(__file__))
mpmath.mp.dps = 40

ZERO = (Fr(0), Fr(0))


# ------------------------------------------------------------------ Gaussian-rational mode tables ----
def cmul(a, b):
    return (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])


def fadd(f, g, sg=Fr(1)):
    out = dict(f)
    for m, c in g.items():
        o = out.get(m, ZERO)
        out[m] = (o[0] + sg * c[0], o[1] + sg * c[1])
    return {m: c for m, c in out.items() if c != ZERO}


def fscale(f, s):
    return {m: (c[0] * s, c[1] * s) for m, c in f.items() if c != ZERO}


def fmul(f, g):
    out = {}
    for m1, c1 in f.items():
        for m2, c2 in g.items():
            m = (m1[0] + m2[0], m1[1] + m2[1], m1[2] + m2[2])
            p = cmul(c1, c2)
            o = out.get(m, ZERO)
            out[m] = (o[0] + p[0], o[1] + p[1])
    return {m: c for m, c in out.items() if c != ZERO}


def second_derivative(f, a, b):
    """k_a k_b / k^2, k = 0 untouched (src/fmax-pfft.c:366-373, 444-456)"""
    out = {}
    for m, c in f.items():
        m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2]
        out[m] = c if m2 == 0 else (c[0] * Fr(m[a] * m[b], m2), c[1] * Fr(m[a] * m[b], m2))
    return {m: c for m, c in out.items() if c != ZERO}


def first_derivative(f, a, growth):
    """i k_a / k^2 * growth in units of N / (2 pi); the k = 0 mode becomes imaginary and is dropped by the c2r
    (src/fmax-pfft.c:366-384)"""
    out = {}
    for m, c in f.items():
        m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2]
        if m2 == 0:
            continue
        s = Fr(m[a], m2) * growth
        out[m] = (-c[1] * s, c[0] * s)  # times i
    return {m: c for m, c in out.items() if c != ZERO}


def check_real(f):
    for m, c in f.items():
        cc = f.get((-m[0], -m[1], -m[2]), ZERO)
        assert cc == (c[0], -c[1]), ("not Hermitian", m)


def waves_to_modes(waves, dc):
    """delta(x) = dc + sum_j [ a_j cos(2 pi m_j.x / N) - b_j sin(2 pi m_j.x / N) ]  <=>  c[m_j] = (a_j + i b_j) / 2"""
    f = {}
    for m, a, b in waves:
        m = tuple(m)
        f = fadd(f, {m: (Fr(a) / 2, Fr(b) / 2), (-m[0], -m[1], -m[2]): (Fr(a) / 2, -Fr(b) / 2)})
    if dc != 0:
        f = fadd(f, {(0, 0, 0): (Fr(dc), Fr(0))})
    return f


PAIRS = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]  # storage order of the reference (src/LPT.c:36-44)


def derive(waves, dc, growth):
    delta = waves_to_modes(waves, dc)
    check_real(delta)
    d = [second_derivative(delta, a, b) for a, b in PAIRS]
    # src/LPT.c:64-70
    s2 = fadd(fadd(fmul(d[0], d[1]), fmul(d[0], d[2])), fmul(d[1], d[2]))
    for i in (3, 4, 5):
        s2 = fadd(s2, fmul(d[i], d[i]), Fr(-1))
    # src/LPT.c:73-82
    t1 = fmul(d[0], fadd(fmul(d[1], d[2]), fmul(d[5], d[5]), Fr(-1)))
    t2 = fmul(d[3], fadd(fmul(d[3], d[2]), fmul(d[4], d[5]), Fr(-1)))
    t3 = fmul(d[4], fadd(fmul(d[3], d[5]), fmul(d[4], d[1]), Fr(-1)))
    s3a = fscale(fadd(fadd(t1, t2, Fr(-1)), t3), Fr(3))
    # src/LPT.c:84-86, then :112-137
    s3b = fscale(fmul(fadd(fadd(d[0], d[1]), d[2]), s2), Fr(2))
    phi2 = [second_derivative(s2, a, b) for a, b in PAIRS]
    for i in range(6):
        s3b = fadd(s3b, fmul(phi2[i], d[i]), Fr(-2) * (1 if i < 3 else 2))
    for f in d + [s2, s3a, s3b] + phi2:
        check_real(f)
    vel = {}
    for name, src, g in (("Vel", delta, growth[0]), ("Vel_2LPT", s2, growth[1]), ("Vel_3LPT_1", s3a, growth[2]),
                         ("Vel_3LPT_2", s3b, growth[3])):
        vel[name] = [first_derivative(src, a, g) for a in range(3)]
        for f in vel[name]:
            check_real(f)
    return dict(delta=delta, d=d, s2=s2, s3a=s3a, s3b=s3b, phi2=phi2, vel=vel)


# ------------------------------------------------------------------------------- sympy cross-check ----
def modes_to_sympy(f, X, unit):
    """sum_m c[m] exp(i unit m.X) as a real expression"""
    e = sympy.Integer(0)
    for m, c in f.items():
        ph = unit * (m[0] * X[0] + m[1] * X[1] + m[2] * X[2])
        e += sympy.Rational(c[0].numerator, c[0].denominator) * sympy.cos(ph) - sympy.Rational(c[1].numerator, c[1].denominator) * sympy.sin(ph)
    return e


def sympy_check(waves, dc, growth, r):
    """Independent route: potential by hand per wave, Hessian by symbolic differentiation, sources by the formulas on those
    expressions; the mode tables must agree at random points, and the displacements must be the irrotational fields whose
    divergence is minus growth times the mean-free source."""
    X = sympy.symbols("x y z", real=True)
    u = sympy.Symbol("u", positive=True)  # 2 pi / N
    rat = lambda q: sympy.Rational(Fr(q).numerator, Fr(q).denominator)
    # potential with laplacian(phi) = delta - dc:  phi = -sum (a cos - b sin) / k^2
    phi = sympy.Integer(0)
    for m, a, b in waves:
        ph = u * (m[0] * X[0] + m[1] * X[1] + m[2] * X[2])
        k2 = u * u * (m[0] ** 2 + m[1] ** 2 + m[2] ** 2)
        phi -= (rat(a) * sympy.cos(ph) - rat(b) * sympy.sin(ph)) / k2
    # second derivatives of the reference: phi_,ab plus the untouched mean in every component
    d = [sympy.diff(phi, X[a], X[b]) + rat(dc) for a, b in PAIRS]
    s2 = d[0] * d[1] + d[0] * d[2] + d[1] * d[2] - d[3] ** 2 - d[4] ** 2 - d[5] ** 2
    s3a = 3 * (d[0] * (d[1] * d[2] - d[5] * d[5]) - d[3] * (d[3] * d[2] - d[4] * d[5]) + d[4] * (d[3] * d[5] - d[4] * d[1]))
    rng = random.Random(12345)
    pts = [[sympy.Rational(rng.randrange(0, 16000), 1000) for _ in range(3)] for _ in range(4)]
    uval = 2 * sympy.pi / 16

    def val(e, p):
        return sympy.N(e.subs({X[0]: p[0], X[1]: p[1], X[2]: p[2], u: uval}), 30)

    def same(e1, e2, what, scale=1):
        for p in pts:
            a, b = val(e1, p), val(e2, p)
            assert abs(a - b) <= sympy.Float(10) ** -25 * (scale + abs(a)), (what, a, b)

    for i in range(6):
        same(d[i], modes_to_sympy(r["d"][i], X, u), "d%d" % i)
    same(s2, modes_to_sympy(r["s2"], X, u), "s2")
    same(s3a, modes_to_sympy(r["s3a"], X, u), "s3a")
    # phi2_ab: laplacian-inverse of the mean-free S2, differentiated, plus the untouched mean <S2> in all six
    s2m = r["s2"].get((0, 0, 0), ZERO)[0]
    s2free = {m: c for m, c in r["s2"].items() if m != (0, 0, 0)}
    pot2 = {m: (-c[0] / (m[0] ** 2 + m[1] ** 2 + m[2] ** 2), -c[1] / (m[0] ** 2 + m[1] ** 2 + m[2] ** 2)) for m, c in s2free.items()}
    pot2e = modes_to_sympy(pot2, X, u) / (u * u)
    same(sympy.diff(pot2e, X[0], 2) + sympy.diff(pot2e, X[1], 2) + sympy.diff(pot2e, X[2], 2), modes_to_sympy(s2free, X, u), "laplacian phi2")
    phi2 = [sympy.diff(pot2e, X[a], X[b]) + rat(s2m) for a, b in PAIRS]
    for i in range(6):
        same(phi2[i], modes_to_sympy(r["phi2"][i], X, u), "phi2_%d" % i)
    s3b = 2 * (d[0] + d[1] + d[2]) * modes_to_sympy(r["s2"], X, u)
    for i in range(6):
        s3b -= 2 * (1 if i < 3 else 2) * phi2[i] * d[i]
    same(s3b, modes_to_sympy(r["s3b"], X, u), "s3b")
    # displacements: Psi (in grid units) = (1/u) * table; div Psi = -g (S - <S>), curl Psi = 0
    for name, src, g in (("Vel", r["delta"], growth[0]), ("Vel_2LPT", r["s2"], growth[1]), ("Vel_3LPT_1", r["s3a"], growth[2]),
                         ("Vel_3LPT_2", r["s3b"], growth[3])):
        psi = [modes_to_sympy(r["vel"][name][a], X, u) / u for a in range(3)]
        free = {m: c for m, c in src.items() if m != (0, 0, 0)}
        div = sum(sympy.diff(psi[a], X[a]) for a in range(3))
        same(div, -rat(g) * modes_to_sympy(free, X, u), "div " + name)
        for a, b in ((0, 1), (0, 2), (1, 2)):
            same(sympy.diff(psi[a], X[b]), sympy.diff(psi[b], X[a]), "curl " + name)


# --------------------------------------------------------------------------------------- output ----
def table_json(f):
    return [[list(m), str(c[0]), str(c[1])] for m, c in sorted(f.items())]


def evaluate(f, n, cells):
    """mpmath values of the field at the listed cells (x, y, z) of the n^3 grid"""
    out = []
    for (x, y, z) in cells:
        s = mpmath.mpf(0)
        for m, c in f.items():
            ph = 2 * mpmath.pi * ((m[0] * x + m[1] * y + m[2] * z) % n) / n
            s += mpmath.mpf(c[0].numerator) / c[0].denominator * mpmath.cos(ph) - mpmath.mpf(c[1].numerator) / c[1].denominator * mpmath.sin(ph)
        out.append(float(s))
    return out


# ---- SCALE_DEPENDENT build: growth per mode (src/cosmo.c:1728-1755 InterpolateGrowth, :1789-1819 GrowingMode*) ----
# compute_derivative hands |k| in rad/cell to GrowingMode*(z, k) (src/fmax-pfft.c:339-364): the multiplier of mode m is
#   sign * 10^( (1 - w) T[kk] + w T[kk+1] ),  t = (log10 k - LOGKMIN) / DELTALOGK, kk = (int) t, w = t - kk,
#   T[0] below kmin = 10^LOGKMIN, T[nk-1] above kmax = 10^(LOGKMIN + (nk-1) DELTALOGK);  k = 2 pi |m| / N.
# T = the log10-growth of each k-bin at the redshift of the call (what the spline of each bin returns).  The displacement
# tables above scale mode by mode: coefficient(m) = coefficient at unit growth * multiplier(|m|).  Evaluated here in 40 digits.
SD = dict(logkmin=Fr(-1), dlogk=Fr(1, 4), sign=[1, 1, -1, 1],
          T=[[Fr(-1, 20), Fr(-1, 10), Fr(-1, 5), Fr(-3, 10), Fr(-9, 20), Fr(-3, 5)],
             [Fr(-2, 5), Fr(-9, 20), Fr(-1, 2), Fr(-3, 5), Fr(-7, 10), Fr(-9, 10)],
             [Fr(-1), Fr(-19, 20), Fr(-9, 10), Fr(-4, 5), Fr(-3, 4), Fr(-7, 10)],
             [Fr(-9, 10), Fr(-1), Fr(-11, 10), Fr(-6, 5), Fr(-5, 4), Fr(-13, 10)]])


def growth_of_mode(order, m2, n):
    T = [mpmath.mpf(t.numerator) / t.denominator for t in SD["T"][order]]
    nk = len(T)
    lkmin = mpmath.mpf(SD["logkmin"].numerator) / SD["logkmin"].denominator
    dlk = mpmath.mpf(SD["dlogk"].numerator) / SD["dlogk"].denominator
    k = 2 * mpmath.pi * mpmath.sqrt(m2) / n
    if k < mpmath.power(10, lkmin):
        v = T[0]
    elif k > mpmath.power(10, lkmin + (nk - 1) * dlk):
        v = T[nk - 1]
    else:
        t = (mpmath.log10(k) - lkmin) / dlk
        kk = int(mpmath.floor(t))
        w = t - kk
        assert 1e-9 < w < 1 - 1e-9 and kk + 1 < nk, "a mode on a bin edge would hang on the last bit of log10"
        v = w * T[kk + 1] + (1 - w) * T[kk]
    return SD["sign"][order] * mpmath.power(10, v)


def scale_dependent_section(r):
    m2s = sorted({m[0] ** 2 + m[1] ** 2 + m[2] ** 2 for fs in r["vel"].values() for f in fs for m in f})
    gk = {}
    for n in (16, 32, 64):
        gk[str(n)] = [{str(m2): float(growth_of_mode(o, m2, n)) for m2 in m2s} for o in range(4)]
    return dict(logkmin=str(SD["logkmin"]), dlogk=str(SD["dlogk"]), sign=SD["sign"], T=[[str(t) for t in row] for row in SD["T"]], gk=gk)


CASES = [
    # three waves in general position, no mean; growth multipliers of the EdS normalisation (src/cosmo.c:250-257, sign of
    # the 3LPT_1 term from GrowingMode_3LPT_1, src/cosmo.c:1810)
    dict(name="three_waves", dc=Fr(0), growth=[Fr(1), Fr(3, 7), Fr(-1, 9), Fr(5, 42)],
         waves=[((1, 2, 0), Fr(3, 4), Fr(-1, 2)), ((2, -1, 1), Fr(1, 2), Fr(1, 3)), ((0, 1, 2), Fr(-2, 5), Fr(1, 4))]),
    # two waves plus a non-zero mean: the k = 0 mode of delta and of S2 survives in all six second derivatives (k^2 = 0 is
    # skipped, src/fmax-pfft.c:368), other growth values (a later redshift, a modified-gravity run)
    dict(name="two_waves_with_mean", dc=Fr(1, 8), growth=[Fr(4, 5), Fr(1, 4), Fr(-1, 16), Fr(3, 32)],
         waves=[((2, 0, 1), Fr(1), Fr(1, 2)), ((-1, 2, 2), Fr(-3, 5), Fr(2, 5))]),
    # a single wave: rank-one tensor, all three sources vanish identically (only Zel'dovich moves)
    dict(name="one_wave", dc=Fr(0), growth=[Fr(1), Fr(3, 7), Fr(-1, 9), Fr(5, 42)],
         waves=[((1, 1, 2), Fr(2), Fr(-1))]),
    # waves along the axes with a common axis: S2 couples only pairs with different directions
    dict(name="axis_waves", dc=Fr(0), growth=[Fr(1), Fr(3, 7), Fr(-1, 9), Fr(5, 42)],
         waves=[((2, 0, 0), Fr(1), Fr(0)), ((0, 1, 0), Fr(0), Fr(1)), ((0, 0, 2), Fr(1, 2), Fr(1, 2)), ((1, 1, 0), Fr(1, 3), Fr(0))]),
]


def main():
    rng = random.Random(486604)
    out = dict(description="closed-form LPT known answers; coefficient tables [m, re, im] of f(x) = sum c[m] exp(2 pi i m.x/N); "
                           "vel tables in units of N/(2 pi); made by tests/golden/make_lpt_analytic.py", cases=[])
    for case in CASES:
        r = derive(case["waves"], case["dc"], case["growth"])
        sympy_check(case["waves"], case["dc"], case["growth"], r)
        if case["name"] == "one_wave":
            assert not r["s2"] and not r["s3a"] and not r["s3b"]
        mmax = max(max(abs(x) for x in m) for f in [r["s3a"], r["s3b"], r["s2"], r["delta"]] for m in f) if r["s2"] else 2
        assert mmax <= 6
        n = 16
        cells = [(rng.randrange(n), rng.randrange(n), rng.randrange(n)) for _ in range(96)]
        sample = dict(n=n, cells=[list(c) for c in cells])
        sample["d"] = [evaluate(f, n, cells) for f in r["d"]]
        sample["s2"] = evaluate(r["s2"], n, cells)
        sample["s3a"] = evaluate(r["s3a"], n, cells)
        sample["s3b"] = evaluate(r["s3b"], n, cells)
        # displacement samples in grid units: N / (2 pi) applied in 40 digits
        fac = n / (2 * mpmath.pi)
        sample["vel"] = {k: [[float(fac * mpmath.mpf(v)) for v in evaluate(f, n, cells)] for f in fs] for k, fs in r["vel"].items()}
        out["cases"].append(dict(
            name=case["name"], dc=str(case["dc"]), growth=[str(g) for g in case["growth"]],
            waves=[[list(m), str(a), str(b)] for m, a, b in case["waves"]], max_mode=mmax,
            delta=table_json(r["delta"]), d=[table_json(f) for f in r["d"]], s2=table_json(r["s2"]), s3a=table_json(r["s3a"]),
            s3b=table_json(r["s3b"]), phi2=[table_json(f) for f in r["phi2"]],
            vel={k: [table_json(f) for f in fs] for k, fs in r["vel"].items()}, sample=sample,
            scale_dependent=scale_dependent_section(r)))
        print(case["name"], "modes:", len(r["delta"]), len(r["s2"]), len(r["s3a"]), len(r["s3b"]), "max |m_i|", mmax)
    with open(os.path.join(HERE, "lpt_analytic.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "lpt_analytic.json"))


if __name__ == "__main__":
    main()
