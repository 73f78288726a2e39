// pf_collapse_core.h -- per-cell ellipsoidal-collapse solve (fp64), device code.
//
// Follows the arithmetic of the reference operation by operation so that the
// stored fp32 Fmax agrees with the CPU path:
//   inverse_collapse_time  src/collapse_times.c:679-776
//   ord                    src/collapse_times.c:1354-1362
//   ell / ell_classic      src/collapse_times.c:404-427 / 114-221
//   InverseGrowingMode     src/cosmo.c:1822-1832, my_spline_eval :2016-2027,
//                          gsl_spline_eval of a natural cubic spline (GSL 2.7.1 cspline.c)
// Quirks kept on purpose: pow(x, 0.333333333333333) not
// cbrt (Q5); -10 sentinel without failure flag (Q4); acos argument may leave
// [-1,1] by round-off -> NaN -> the Fmax comparison is false (Q4); arithmetic
// middle in ord (A10).
//
// Also compiled for the host by tests/cpu_emul/collapse_emul.cpp (unit test of
// this header against the oracle's known answers; not a CPU path of the library).
#pragma once
#include <math.h>
#include "pf_gtab.h"
#include "pf_c3tab.h"

// No FMA contraction in the solver: several steps subtract nearly equal numbers
// (r*r - q*q*q at a double root, the cubic's -(s + q/s) - a1/3), where a fused
// multiply-add changes the result by far more than an ulp.  The reference is
// plain IEEE arithmetic; with contraction off only the libm calls can differ.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#ifndef PF_HD
#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif
#endif

#define PF_PI 3.14159265358979323846 /* src/pinocchio.h:56 */
#define PF_SMALL 1.e-20              /* src/collapse_times.c:38 */

// natural cubic spline view: knots x,y, second-derivative coefficients c, and the per-interval b_i, d_i that GSL's
// cspline_eval derives from them at every call (coeff_calc) -- computed once on the host with the very same IEEE
// operations, so the evaluation below is bit-identical to gsl_spline_eval and three divisions shorter.
struct pf_spline_view {
  const double *x, *y, *c, *b, *d;
  int n;
  // optional start index per bin of a uniform grid (pf_spline_lut_*): the interval search starts there instead of bisecting;
  // the interval found is the same (host test against GSL's bisection on even and wildly uneven knots).  Two geometries:
  //   walk   : PF_SPLINE_LUT_BINS bins over [x[0], x[n-1]]; the search starts one bin early and walks forward (any knots)
  //   direct : bins of width 2^k from a multiple of 2^k (the bin of v is then exact up to a sliver of one rounding below a
  //            bin edge, where it may come out one too high); usable when no bin holds more than one knot, i.e. when the
  //            table entries of neighbouring bins differ by at most one: the interval is lut[bin] or one of its two
  //            neighbours, decided by two comparisons with a pair of knots read together -- no loop
  const unsigned short *lut = nullptr;
  double lut_inv_w = 0.0, lut_x0 = 0.0;
  int lut_direct = 0;
  double x_first = 0.0, x_last = 0.0;  // x[0], x[n-1] when lut is set (registers instead of two broadcast reads per call)
  // fast flavour: the composite 10^(-S(log10 D)) as one table of polynomials in D (pf_gtab.h); rec == nullptr: none
  pf_gtab_view gt;
  // fast flavour: the table form of the cosine triple (pf_cos3_of_acos_tab); nullptr: the single polynomial
  const double *c3tab = nullptr;
};
#define PF_SPLINE_LUT_BINS 4096

PF_HD double pf_spline_eval(const pf_spline_view &s, double v) {
  const double *xa = s.x, *ya = s.y, *ca = s.c;
  const int last = s.n - 1;
  const double x0 = s.lut ? s.x_first : xa[0], xl = s.lut ? s.x_last : xa[last];
  if (v < x0) return ya[0] + (v - xa[0]) * (ya[1] - ya[0]) / (xa[1] - xa[0]);
  if (v > xl)
    return ya[last] + (v - xa[last]) * (ya[last] - ya[last - 1]) / (xa[last] - xa[last - 1]);
  // gsl_interp_bsearch(xa, v, 0, n-1): the largest i <= n-2 with xa[i] <= v
  int ilo = 0;
  double xlo;
  if (s.lut && s.lut_direct) {
    int bin = (int)((v - s.lut_x0) * s.lut_inv_w);
    bin = bin < 0 ? 0 : (bin > PF_SPLINE_LUT_BINS - 1 ? PF_SPLINE_LUT_BINS - 1 : bin);
    const int st = s.lut[bin];
    const double xs = xa[st], xn = xa[st + 1];  // st <= n-2: both in range
    const bool up = st + 1 < last && xn <= v;
    ilo = up ? st + 1 : st;
    xlo = up ? xn : xs;
    if (xs > v) { ilo = st - 1; xlo = xa[ilo]; }  // the sliver below a bin edge with a knot on that edge (v >= x[0], so st >= 1 here)
  } else {
    if (s.lut) {
      int bin = (int)((v - x0) * s.lut_inv_w) - 1;  // one bin early: immune to the rounding of the bin index
      bin = bin < 0 ? 0 : (bin > PF_SPLINE_LUT_BINS - 1 ? PF_SPLINE_LUT_BINS - 1 : bin);
      ilo = s.lut[bin];
      while (ilo + 1 < last && xa[ilo + 1] <= v) ilo++;
    } else {
      int ihi = last;
      while (ihi > ilo + 1) {
        int i = (ihi + ilo) >> 1;
        if (xa[i] > v) ihi = i; else ilo = i;
      }
    }
    xlo = xa[ilo];
  }
  const double delx = v - xlo;
  return ya[ilo] + delx * (s.b[ilo] + delx * (ca[ilo] + delx * s.d[ilo]));
}

// geometry of the start table for knots xa[0..n-1].  Direct form: bin width w = 2^k, the smallest power of two that covers the
// knot range with PF_SPLINE_LUT_BINS - 1 bins, first bin edge x0 = the multiple of w at or below xa[0] (bin edges x0 + b w are
// then exact doubles and (v - x0) / w is exact at every edge).  Walk form: PF_SPLINE_LUT_BINS equal bins over the knot range.
PF_HD void pf_spline_lut_geometry(const double *xa, int n, bool direct, double &x0, double &inv_w) {
  const double range = xa[n - 1] - xa[0];
  if (direct) {
    int e;
    (void)frexp(range / (double)(PF_SPLINE_LUT_BINS - 2), &e);  // 2^(e-1) <= range / (B - 2) < 2^e
    const double w = ldexp(1.0, e);
    x0 = floor(xa[0] / w) * w;
    inv_w = ldexp(1.0, -e);
  } else {
    x0 = xa[0];
    inv_w = (double)PF_SPLINE_LUT_BINS / range;
  }
}
// lut[bin] = largest i <= n-2 with x[i] <= lower edge of the bin (0 when the edge lies below x[0]); one entry per call
PF_HD unsigned short pf_spline_lut_entry(const double *xa, int n, int bin, double x0, double inv_w, bool direct) {
  const double v = direct ? x0 + (double)bin / inv_w : xa[0] + bin * ((xa[n - 1] - xa[0]) / (double)PF_SPLINE_LUT_BINS);
  int ilo = 0, ihi = n - 1;
  while (ihi > ilo + 1) {
    int i = (ihi + ilo) >> 1;
    if (xa[i] > v) ihi = i; else ilo = i;
  }
  return (unsigned short)ilo;
}

// host side of the above: b_i = dy/dx - dx (c_{i+1} + 2 c_i)/3, d_i = (c_{i+1} - c_i)/(3 dx)   (GSL coeff_calc)
inline void pf_spline_bd(const double *xa, const double *ya, const double *ca, int n, double *b, double *d) {
  for (int i = 0; i + 1 < n; i++) {
    const double dx = xa[i + 1] - xa[i], dy = ya[i + 1] - ya[i];
    b[i] = (dy / dx) - dx * (ca[i + 1] + 2.0 * ca[i]) / 3.0;
    d[i] = (ca[i + 1] - ca[i]) / (3.0 * dx);
  }
  b[n - 1] = d[n - 1] = 0.0;
}

// ---- the three transcendental hot spots, in two flavours ---------------------------------------
// FAST = false: the reference's own calls (cos x3, acos, pow(x, 0.333333333333333), pow(10., y), log10, exp, IEEE / and sqrt).
// FAST = true : algebraically identical forms that cost a fraction of the instructions on gfx950
//   cos((acos x + 2 pi k)/3), k = 0, 1, 2      ->  the roots of the Chebyshev cubic from one polynomial and two square roots (pf_cos3_of_acos)
//   x / constant, x / a / b, 1 / 10^y           ->  x * (1/constant), x / (a*b), 10^-y
//   pow(x, 0.333333333333333)                 ->  cbrt(x) * (1 - d ln x), d = 1/3 - 0.333333333333333
//   pow(10., y), exp, log10                   ->  series forms (pf_exp10_series, pf_exp_series, pf_log10_pos)
// Each differs from the reference call by about one ulp, like the device libm differs from glibc.
// One Horner step r z + c as ONE v_fma_f64.  Left to itself the compiler selects the two-address v_fmac_f64 and, because
// the coefficient is loop invariant and lives on in its VGPR pair, pays a v_mov_b64 copy in front of every step (60 of
// the 1154 vector instructions of the collapse kernel).  Same operation, same rounding: results do not change.
PF_HD double pf_horner(double r, double z, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(r), "v"(z), "v"(c));
  return d;
#else
  return fma(r, z, c);
#endif
}
// The same step with the coefficient in a scalar register pair (a table in constant memory, fetched by scalar loads beside
// the vector stream): left to itself the compiler copies such a coefficient into a VGPR pair (two v_mov_b32) for its
// two-address v_fmac_f64 -- three vector instructions per step instead of one.
PF_HD double pf_horner_s(double r, double z, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(r), "v"(z), "s"(c));
  return d;
#else
  return fma(r, z, c);
#endif
}
// Division and square root of the fast flavour on the device: the hardware seeds (v_rcp_f64, v_rsq_f64) refined by
// one Newton / Goldschmidt step and one final residual correction -- 6 and 10 operations instead of the 12 and 22 of the
// IEEE expansions, which spend the difference on a second step and on operand scaling for the subnormal and overflow ranges.  The arguments
// here are the cubic's coefficients and discriminants (normal range; zero handled); the result is the correctly rounded
// one except for rare 1-ulp cases.  The host build (tests) and the exact flavour keep the plain operators.
PF_HD double pf_div_fast(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);      // 2^-24.4 (measured, scratch/seedacc.py)
  r = fma(fma(-b, r, 1.0), r, r);          // 2^-48.8
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);         // exact residual times r: a second Newton step on r changes no result
#else
  return a / b;
#endif
}
#ifndef PF_SQRT_BRANCHLESS
#define PF_SQRT_BRANCHLESS 1  // (0 in an A/B build: zero, negative and NaN arguments through a branch to the library's sqrt)
#endif
PF_HD double pf_sqrt_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
#if PF_SQRT_BRANCHLESS
  // no branch: a negative or NaN argument gives NaN through the seed itself, as the library's sqrt; a zero (seed +inf) keeps its own
  // value by a seed of zero.  Round 5: the branch cut the solve into small blocks, each with its own exec-mask bookkeeping and with
  // hazard s_nops the scheduler had nothing to fill with -- six square roots per cell
  const double y = x == 0.0 ? 0.0 : __builtin_amdgcn_rsq(x);  // 2^-24.2
#else
  if (!(x > 0.0)) return x == 0.0 ? x : sqrt(x);  // zero, negative and NaN as the library
  const double y = __builtin_amdgcn_rsq(x);  // 2^-24.2
#endif
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);        // 2^-47.8
  return fma(fma(-g, g, x), h, g);
#else
  return sqrt(x);
#endif
}

// The cosine triple of the trigonometric root formula without its transcendentals.  c_k = cos((acos x + 2 pi k) / 3) are
// the roots of the Chebyshev cubic 4 c^3 - 3 c = x.  With y = sqrt((1 + x) / 2) = cos(3 theta / 2), theta = acos(x) / 3,
// the largest root is c_1 = cos(2/3 acos y), and (1 - c_1) / (1 - x) = h(y) is analytic on [0, 1] (nearest singularity
// y = -1): one degree-22 polynomial in t = 2 y - 1 (Chebyshev fit in 60-digit arithmetic, truncation 7e-18, within 1 ulp in
// double Horner form; tests/test_collapse_core.py against mpmath).  Then c_1 = 1 - (1 - x) h, sin(theta) = sqrt((1 - x) h
// (1 + c_1)) -- 1 - c_1^2 without the cancellation at x -> 1 -- and the rotation by 2 pi / 3 gives c_2, c_3.  52 operations
// where acos (55) + sincos with its rotation (34) took 89; each result within ~2 ulp of the reference's cos(acos(.)/3)
// calls where those are well conditioned, and conditioned like them at x -> +-1 (a degenerate pair of roots).  |x| > 1 by
// round-off gives NaN in all three, as acos does (quirk Q4).
#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ double pf_c3_coef[23] = {
#else
static const double pf_c3_coef[23] = {
#endif
    1.43396340762163678e-11,  -4.29427803767481323e-11, 4.61317068002735218e-11,  -1.38050045298289521e-10, 6.19161692721708188e-10,
    -1.85273563663425955e-09, 5.24898621624865300e-09,  -1.56978011134299560e-08, 4.71941580314714327e-08,  -1.41043393253484189e-07,
    4.21175556679195000e-07,  -1.25745808061941212e-06, 3.75178490461992615e-06,  -1.11842872488483324e-05, 3.33048358182611446e-05,
    -9.90342153831625456e-05, 2.93911199246382684e-04,  -8.69842889830885636e-04, 2.56357950483709473e-03,  -7.50383677308579921e-03,
    2.16901626784005916e-02,  -6.09591300458922972e-02, 1.55970371254014639e-01};
PF_HD void pf_cos3_of_acos(double x, double &c1, double &c2, double &c3) {
  const double s = pf_sqrt_fast(1.0 + x);                 // NaN for x < -1
  const double t = fma(s, 1.41421356237309504880, -1.0);  // 2 y - 1
  double h = pf_c3_coef[0];
#pragma unroll
  for (int i = 1; i < 23; i++) h = pf_horner_s(h, t, pf_c3_coef[i]);
  const double u = (1.0 - x) * h;                         // 1 - c1
  const double cs = 1.0 - u;
  const double sn = pf_sqrt_fast(u * (1.0 + cs));         // NaN for x > 1
  const double g = 0.86602540378443864676 * sn;          // sin(2 pi / 3) sin(theta)
  c1 = fma(0.0, sn, cs);                                  // (... in all three)
  c2 = -0.5 * cs - g;
  c3 = -0.5 * cs + g;
}

// The same triple with h from a table of short polynomials (pf_c3tab.h: 32 pieces of t in [-1, 1], degree 7 in u = t - centre,
// Chebyshev interpolants fitted in 60-digit arithmetic by profiles/tools/make_c3tab.py; 2 KB, in LDS in the cell kernels): the piece
// from s itself (t + 1 = s sqrt 2), 7 Horner steps instead of 22 -- 13 operations for h where the single polynomial takes 22.  Each
// piece lies 64 and more of its half-widths from the singularity at t = -3: truncation below 1e-17, the result within the same
// 2 ulp of the reference's calls as the single polynomial (tests/test_collapse_core.py, both forms against mpmath).
PF_HD void pf_cos3_of_acos_tab(const double *tab, double x, double &c1, double &c2, double &c3) {
  const double s = pf_sqrt_fast(1.0 + x);                 // NaN for x < -1
  int i = (int)(s * (PF_C3_BINS * 0.70710678118654752440));
#if !defined(__HIP_DEVICE_COMPILE__)
  if (!(s >= 0.0)) i = 0;                                 // (the device's conversion gives 0 for NaN; the NaN itself goes on through u)
#endif
  i = i < PF_C3_BINS - 1 ? i : PF_C3_BINS - 1;            // x = 1: the last piece
  const double u = fma((double)i, -2.0 / PF_C3_BINS, fma(s, 1.41421356237309504880, -1.0 / PF_C3_BINS));  // t - centre_i
  const double *r = tab + i * (PF_C3_DEG + 1);
  double h = r[PF_C3_DEG];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = PF_C3_DEG - 1; k >= 0; k--) h = fma(h, u, r[k]);
  const double w = (1.0 - x) * h;                         // 1 - c1
  const double cs = 1.0 - w;
  const double sn = pf_sqrt_fast(w * (1.0 + cs));         // NaN for x > 1
  const double g = 0.86602540378443864676 * sn;
  c1 = fma(0.0, sn, cs);
  c2 = -0.5 * cs - g;
  c3 = -0.5 * cs + g;
}
// (tab: the table where the caller has one at hand -- a compile-time fact in the kernels -- else the single polynomial)
PF_HD void pf_cos3_fast(const double *tab, double x, double &c1, double &c2, double &c3) {
  if (tab) pf_cos3_of_acos_tab(tab, x, c1, c2, c3);
  else pf_cos3_of_acos(x, c1, c2, c3);
}

// x / Y for a constant Y, correctly rounded, in three operations (Markstein): with c = RN(1/Y), q0 = RN(x c),
// the residual x - Y q0 is exact in an fma and q0 + residual c rounds to RN(x / Y) -- the same double the
// reference's division produces, so the q^3 < r^2 sentinel test sees the reference's own q and r
template <int Y> PF_HD double pf_div_const(double x) {
  const double c = 1.0 / (double)Y;
  const double q0 = x * c;
  return fma(fma(-(double)Y, q0, x), c, q0);
}

// the reference's three cos calls (exact flavour; the fast one never forms t: pf_cos3_of_acos)
PF_HD void pf_cos3_libm(double t, double &c1, double &c2, double &c3) {
  const double inv_3 = 1.0 / 3.0;
  c1 = cos(t * inv_3);
  c2 = cos((t + 2. * PF_PI) * inv_3);
  c3 = cos((t + 4. * PF_PI) * inv_3);
}
template <bool FAST> PF_HD double pf_pow_third(double x) {
  if (!FAST) return pow(x, 0.333333333333333);
  // x^(1/3 - d) = cbrt(x) exp(-d ln x); d = 1/3 - (the double nearest to 0.333333333333333) = 3.5157e-16, so ln x is
  // needed to ~1e-2 only: exponent plus a quadratic in the mantissa (log2 m on [1/2, 1) to 0.009)
  int e;
  const double m = frexp(x, &e);
  const double log2x = (double)e + pf_horner(pf_horner(-1.34752114, m, 3.98979292), m, -2.64898574);
  return cbrt(x) * fma(log2x, -(3.515706244646329e-16 * 0.6931471805599453), 1.0);
}
// exp and 10^y of the fast flavour in 20 and 21 operations (the library calls cost 42 and 44 on gfx950, half of them
// moves of their coefficients): k = rint(x log2 e), r = x - k ln2 in two steps (fdlibm's split of ln 2), e^r by one
// degree-11 polynomial on |r| <= ln2 / 2 (1 + r + r^2 g(r), g a Chebyshev fit in 60-digit arithmetic, relative error
// 1.6e-17 with the coefficients rounded to double), scaled by 2^k.  Within 1 ulp of the correctly rounded value
// (tests/test_collapse_core.py).  The argument is clamped to the range where the result is finite and non-zero plus one
// step, so overflow and underflow come out of the scaling as inf and 0; a NaN argument (not reachable from the callers:
// ordered eigenvalues over a positive trace, a spline value) would give 0.
#ifndef PF_EXP_ONE_ASM
#define PF_EXP_ONE_ASM 1  // (0 in an A/B build: one inline-assembly statement per Horner step)
#endif
PF_HD double pf_exp_reduced(double r) {
#if defined(__HIP_DEVICE_COMPILE__) && PF_EXP_ONE_ASM
  // The twelve steps as ONE assembly statement, coefficients in scalar registers: between two single-instruction statements the
  // compiler places an s_nop (it cannot see what they are), and a coefficient in a vector register costs two v_mov per step.
  double p = 2.51003854955103203e-08;
  asm("v_fma_f64 %0, %0, %1, %2\n\t"
      "v_fma_f64 %0, %0, %1, %3\n\t"
      "v_fma_f64 %0, %0, %1, %4\n\t"
      "v_fma_f64 %0, %0, %1, %5\n\t"
      "v_fma_f64 %0, %0, %1, %6\n\t"
      "v_fma_f64 %0, %0, %1, %7\n\t"
      "v_fma_f64 %0, %0, %1, %8\n\t"
      "v_fma_f64 %0, %0, %1, %9\n\t"
      "v_fma_f64 %0, %0, %1, %10\n\t"
      "v_fma_f64 %0, %0, %1, 1.0\n\t"
      "v_fma_f64 %0, %0, %1, 1.0"
      : "+v"(p)
      : "v"(r), "s"(2.76200884454097462e-07), "s"(2.75572684599970641e-06), "s"(2.48015212959543761e-05), "s"(1.98412698630536177e-04),
        "s"(1.38888889172137167e-03), "s"(8.33333333333006153e-03), "s"(4.16666666666241289e-02), "s"(1.66666666666666685e-01),
        "s"(5.00000000000000111e-01));
  return p;
#else
  double p = 2.51003854955103203e-08;
  p = pf_horner(p, r, 2.76200884454097462e-07);
  p = pf_horner(p, r, 2.75572684599970641e-06);
  p = pf_horner(p, r, 2.48015212959543761e-05);
  p = pf_horner(p, r, 1.98412698630536177e-04);
  p = pf_horner(p, r, 1.38888889172137167e-03);
  p = pf_horner(p, r, 8.33333333333006153e-03);
  p = pf_horner(p, r, 4.16666666666241289e-02);
  p = pf_horner(p, r, 1.66666666666666685e-01);
  p = pf_horner(p, r, 5.00000000000000111e-01);
  p = fma(p, r, 1.0);
  return fma(p, r, 1.0);
#endif
}
PF_HD double pf_exp_series(double x) {
  x = fmin(fmax(x, -746.0), 710.0);  // e^-746 = 0 and e^710 = inf in double: k stays a small integer for every x
  const double kd = rint(x * 1.44269504088896338700e+00);
  double r = fma(kd, -6.93147180369123816490e-01, x);
  r = fma(kd, -1.90821492927058770002e-10, r);
  return ldexp(pf_exp_reduced(r), (int)kd);
}
PF_HD double pf_exp10_series(double y) {
  y = fmin(fmax(y, -324.0), 309.0);
  const double kd = rint(y * 3.32192809488736218171e+00);
  double r = fma(kd, -3.01029995663611771306e-01, y);  // fdlibm's split of log10 2: k hi is exact for |k| < 2^13
  r = fma(kd, -3.69423907715893078616e-13, r);
  return ldexp(pf_exp_reduced(r * 2.30258509299404590109e+00), (int)kd);
}
template <bool FAST> PF_HD double pf_pow10(double y) {
  if (!FAST) return pow(10., y);
  return pf_exp10_series(y);
}

// log10 of a positive finite double in ~45 operations (the library call costs 105 on gfx950): x = m 2^e with m in
// [sqrt(1/2), sqrt(2)), log m by the classical s = (m-1)/(m+1) series (the seven-term even polynomial of fdlibm's
// e_log.c, remainder < 2^-58), then (e ln2 + log m) / ln10 with ln2 split in two.  Within 2 ulp of the correctly rounded
// value; it only feeds the abscissa of the inverse-growth spline.
PF_HD double pf_log10_pos(double x) {
  int e;
  double m = frexp(x, &e);                        // m in [0.5, 1)
  if (m < 0.70710678118654752440) { m += m; e -= 1; }
  const double f = m - 1.0;
  const double s = pf_div_fast(f, 2.0 + f);
  const double z = s * s;
  const double w = z * z;
  // (pf_horner: one v_fma_f64 per step -- as plain fma calls the coefficient, which lives on in its registers, is copied in
  //  front of every two-address v_fmac_f64; same operations, same rounding)
  const double t1 = w * pf_horner(w, pf_horner(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * pf_horner(w, pf_horner(w, pf_horner(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  // ln x = dk ln2_hi + (f - (hfsq - (s (hfsq + R) + dk ln2_lo)))
  const double lnx_lo = fma(s, hfsq + R, dk * 1.90821492927058770002e-10);
  const double lnx = fma(dk, 6.93147180369123816490e-01, f - (hfsq - lnx_lo));
  return lnx * 4.34294481903251816668e-01;          // 1 / ln 10
}

template <bool FAST = false> PF_HD double pf_inverse_growing_mode(const pf_spline_view &s, double D) {
  if (FAST) {
    double Y;
    if (s.gt.rec && pf_gtab_eval(s.gt, D, Y)) return Y - 1.;  // (D inside the table: every collapsing cell but the odd extreme one)
    return pf_exp10_series(-pf_spline_eval(s, pf_log10_pos(D))) - 1.;
  }
  return 1. / pf_pow10<FAST>(pf_spline_eval(s, log10(D))) - 1.;
}

// ell_classic (src/collapse_times.c:291-402) in four pieces, so that a kernel can run the two expensive branches of the
// cubic on cells regrouped by branch (k_collapse_grouped); pf_ell_classic below is their plain composition.
struct pf_cubic { double a1, q, r, disc; };  // x^3 + a1 x^2 + a2 x + a3 reduced: q, r and disc = r^2 - q^3
// everything up to the branch on the discriminant.  Returns 0: ell is final (the degenerate branches), 1: one real root
// (disc > 0), 2: three real roots
template <bool FAST = false> PF_HD int pf_ell_setup(double l1, double l2, double l3, double &ell, pf_cubic &c) {
  const double del = l1 + l2 + l3;
  const double det = l1 * l2 * l3;
  if (fabs(l1) < PF_SMALL) {
    ell = -0.1;
    return 0;
  }
  const double den = FAST ? det * (1. / 126.) + 5. * l1 * del * (del - l1) * (1. / 84.)
                          : det / 126. + 5. * l1 * del * (del - l1) / 84.;
  if (fabs(den) < PF_SMALL) {
    if (fabs(del - l1) < PF_SMALL) {
      ell = (l1 > 0.0) ? 1. / l1 : -.1;
    } else {
      const double dis = 7. * l1 * (l1 + 6. * del);
      if (dis < 0.0) {
        ell = -.1;
      } else {
        ell = (7. * l1 - sqrt(dis)) / (3. * l1 * (l1 - del));
        if (ell < 0.) ell = -.1;
      }
    }
    return 0;
  }
  const double rden = FAST ? pf_div_fast(1.0, den) : 1.0 / den;
  const double a1 = FAST ? 3. * l1 * (del - l1) * (1. / 14.) * rden : 3. * l1 * (del - l1) / 14. * rden;
  const double a1_2 = a1 * a1;
  const double a2 = l1 * rden;
  const double a3 = -1.0 * rden;
  c.a1 = a1;
  c.q = FAST ? pf_div_const<9>(a1_2 - 3. * a2) : (a1_2 - 3. * a2) / 9.;
  c.r = FAST ? pf_div_const<54>(2. * a1_2 * a1 - 9. * a1 * a2 + 27. * a3)
             : (2. * a1_2 * a1 - 9. * a1 * a2 + 27. * a3) / 54.;
  c.disc = c.r * c.r - c.q * c.q * c.q;
  return c.disc > 0 ? 1 : 2;
}
template <bool FAST = false> PF_HD double pf_ell_one_root(const pf_cubic &c) {
  const double a1 = c.a1, q = c.q, r = c.r, r_2_q_3 = c.disc;
  const double fabs_r = fabs(r);
  const double sq = pf_pow_third<FAST>((FAST ? pf_sqrt_fast(r_2_q_3) : sqrt(r_2_q_3)) + fabs_r);
  // fabs(r)/r is +-1 for every finite non-zero r (and NaN at r = 0, kept)
  double ell = FAST ? -(r != 0. ? copysign(1.0, r) : fabs_r / r) * (sq + pf_div_fast(q, sq)) - a1 * (1.0 / 3)
                    : -fabs_r / r * (sq + q / sq) - a1 / 3.;
  if (ell < 0.) ell = -.1;
  return ell;
}
template <bool FAST = false> PF_HD double pf_ell_three_roots(const pf_cubic &c, const double *c3tab = nullptr) {
  const double a1 = c.a1, q = c.q, r = c.r;
  const double sq = 2 * (FAST ? pf_sqrt_fast(q) : sqrt(q));
  const double inv_3 = 1.0 / 3;
  double c1, c2, c3;
  if (FAST) pf_cos3_fast(c3tab, pf_div_fast(2 * r, q * sq), c1, c2, c3);
  else pf_cos3_libm(acos(2 * r / q / sq), c1, c2, c3);
  double s1 = -sq * c1 - a1 * inv_3;
  double s2 = -sq * c2 - a1 * inv_3;
  double s3 = -sq * c3 - a1 * inv_3;
  if (s1 < 0.) s1 = 1.e10;
  if (s2 < 0.) s2 = 1.e10;
  if (s3 < 0.) s3 = 1.e10;
  double ell = FAST ? fmin(fmin(s1, s2), s3) : (s1 < s2 ? s1 : s2);
  if (!FAST) ell = (s3 < ell ? s3 : ell);
  if (ell == 1.e10) ell = -.1;
  return ell;
}
// the correction of the collapsing ellipsoids (:395-400)
template <bool FAST = false> PF_HD double pf_ell_finish(double ell, double l1, double l2, double l3) {
  const double del = l1 + l2 + l3;
  if (del > 0. && ell > 0.) {
    const double inv_del = FAST ? pf_div_fast(1.0, del) : 1.0 / del;
    const double arg = -6.5 * (l1 - l2) * inv_del - 2.8 * (l2 - l3) * inv_del;
    ell += -.364 * inv_del * (FAST ? pf_exp_series(arg) : exp(arg));
  }
  return ell;
}
template <bool FAST = false> PF_HD double pf_ell_classic(double l1, double l2, double l3, const double *c3tab = nullptr) {
  double ell = 0.0;
  pf_cubic c;
  const int kind = pf_ell_setup<FAST>(l1, l2, l3, ell, c);
  if (kind == 1) ell = pf_ell_one_root<FAST>(c);
  else if (kind == 2) ell = pf_ell_three_roots<FAST>(c, c3tab);
  return pf_ell_finish<FAST>(ell, l1, l2, l3);
}

// Eigenvalues of the symmetric tensor d = {11,22,33,12,13,23} by the trigonometric formula, ordered by ord()
// (src/collapse_times.c:679-745).  Returns false for the -10 sentinel branch (q^3 < r^2 or q < 0).
// 3LPT(b) source accumulation of one cell (src/LPT.c:134-137): s -= 2 phi2_ab h_ab over the six components in the
// reference's order 11,12,13,22,23,33 (storage index 0,3,4,1,5,2), off-diagonal ones twice.  One definition for
// k_lpt_accum and for the z-pass that forms it on the fly (k_c2r_invariants, MODE 1); never contracted.
PF_HD double pf_lpt3b_accumulate(double s, const double phi2[6], const double h[6]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const int order[6] = {0, 3, 4, 1, 5, 2};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < 6; j++) {
    const int c = order[j];
    const double f = 2.0 * (c < 3 ? 1.0 : 2.0);
    s -= f * phi2[c] * h[c];
  }
  return s;
}

// ... in two pieces: the three invariants of the tensor, and everything after them.  The invariants are all the solve
// needs, so the z-pass of the sweep can store them (3 fields) instead of the six components (k_c2r_invariants); only an
// exactly isotropic tensor (q == 0) takes its eigenvalues from the diagonal itself, `diag`.
// (each invariant on its own too: the z-pass lets one wave form one of them for half a row, k_c2r_invariants; the same
//  operations in the same order as in pf_invariants below, whichever way they are called)
PF_HD double pf_invariant_mu1(const double d[6]) {
#if defined(__clang__)
#pragma clang fp contract(off)  // also when included from a translation unit built with contraction on (the z-pass)
#endif
  return d[0] + d[1] + d[2];
}
PF_HD double pf_invariant_mu2(const double d[6], double mu1) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double mu1_2 = mu1 * mu1;
  double mu2 = 0.5 * mu1_2;
  mu2 -= 0.5 * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double add0 = d[3] * d[3], add1 = d[4] * d[4], add2 = d[5] * d[5];
  mu2 -= add0 + add1 + add2;
  return mu2;
}
PF_HD double pf_invariant_mu3(const double d[6]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double add0 = d[3] * d[3], add1 = d[4] * d[4], add2 = d[5] * d[5];
  return d[0] * d[1] * d[2] + 2. * d[3] * d[4] * d[5] - d[0] * add2 - d[1] * add1 - d[2] * add0;
}
PF_HD void pf_invariants(const double d[6], double &mu1, double &mu2, double &mu3) {
  mu1 = pf_invariant_mu1(d);
  mu2 = pf_invariant_mu2(d, mu1);
  mu3 = pf_invariant_mu3(d);
}
// q of pf_eigen_from_invariants is zero (or so small that the division by nine flushes it) although the diagonal is not
// exactly (mu1/3, mu1/3, mu1/3) -- the one case in which the three invariants do not carry what the solve reads
PF_HD bool pf_invariants_lose_diagonal(const double d[6], double mu1, double mu2) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double x = mu1 * mu1 - 3.0 * mu2;
  if (!(x == 0.0 || (x < 1e-290 && x > -1e-290))) return false;
  const double third = mu1 * (1.0 / 3.0);
  return !(d[0] == third && d[1] == third && d[2] == third);
}
template <bool FAST = false> PF_HD bool pf_eigen_from_invariants(double mu1, double mu2, double mu3, const double diag[3], double lam[3], const double *c3tab = nullptr) {
  const double mu1_2 = mu1 * mu1;
  const double q = FAST ? pf_div_const<9>(mu1_2 - 3.0 * mu2) : (mu1_2 - 3.0 * mu2) / 9.0;
  double x1, x2, x3;
  if (q == 0.) {
    x1 = diag[0]; x2 = diag[1]; x3 = diag[2];
  } else {
    const double r = FAST ? pf_div_const<54>(-(2. * mu1_2 * mu1 - 9.0 * mu1 * mu2 + 27.0 * mu3))
                          : -(2. * mu1_2 * mu1 - 9.0 * mu1 * mu2 + 27.0 * mu3) / 54.;
    if (q * q * q < r * r || q < 0.0) {
      lam[0] = lam[1] = lam[2] = 0.0;
      return false;
    }
    const double sq = 2 * (FAST ? pf_sqrt_fast(q) : sqrt(q));
    const double inv_3 = 1.0 / 3.0;
    double c1, c2, c3;
    if (FAST) pf_cos3_fast(c3tab, pf_div_fast(2 * r, q * sq), c1, c2, c3);
    else pf_cos3_libm(acos(2 * r / q / sq), c1, c2, c3);
    x1 = -sq * c1 + mu1 * inv_3;
    x2 = -sq * c2 + mu1 * inv_3;
    x3 = -sq * c3 + mu1 * inv_3;
  }
  // ord(): hi, lo by comparisons, middle arithmetically
  // (fast flavour: v_max_f64 / v_min_f64 instead of compare + two selects each; the same values except for the sign of
  // a zero and for NaN, which is NaN in all three roots or in none)
  double hi = FAST ? fmax(fmax(x1, x2), x3) : (x1 > x2 ? x1 : x2); if (!FAST) hi = (hi > x3 ? hi : x3);
  double lo = FAST ? fmin(fmin(x1, x2), x3) : (x1 < x2 ? x1 : x2); if (!FAST) lo = (lo < x3 ? lo : x3);
  const double mid = x1 + x2 + x3 - lo - hi;
  lam[0] = hi; lam[1] = mid; lam[2] = lo;
  return true;
}
template <bool FAST = false> PF_HD bool pf_ordered_eigenvalues(const double d[6], double lam[3], const double *c3tab = nullptr) {
  double mu1, mu2, mu3;
  pf_invariants(d, mu1, mu2, mu3);
  return pf_eigen_from_invariants<FAST>(mu1, mu2, mu3, d, lam, c3tab);
}

// ell (src/collapse_times.c:404-427, ELL_CLASSIC): F = 1 + z_collapse, or 0 when the ellipsoid never collapses
template <bool FAST = false> PF_HD double pf_ell(const pf_spline_view &s, double l1, double l2, double l3) {
  const double bc = pf_ell_classic<FAST>(l1, l2, l3, s.c3tab);
  if (bc > 0.0) return 1. + pf_inverse_growing_mode<FAST>(s, bc);
  return 0.0;
}

// d = {11,22,33,12,13,23}.  Returns F = 1 + z_collapse (0: never collapses,
// -10: eigen-solver sentinel).  lam[3] receives the ordered eigenvalues.
template <bool FAST = false> PF_HD double pf_inverse_collapse_time(const double d[6], const pf_spline_view &s, double lam[3]) {
  if (!pf_ordered_eigenvalues<FAST>(d, lam, s.c3tab)) return -10.0;
  return pf_ell<FAST>(s, lam[0], lam[1], lam[2]);
}

// ---- TABULATED_CT (src/collapse_times.c:780-1231, BILINEAR_SPLINE flavour :40) --------------------------------
#define PF_CT_NBINS_XY 50
#define PF_CT_NBINS_D 100
#define PF_CT_RANGE_X 3.5
// the table of one radius: delta knots (shared by all splines), and per (ix, iy) node the values y and the cubic
// coefficients b, c, d of its natural spline in delta; node (ix, iy) starts at (ix + iy * 50) * 100
struct pf_ct_view {
  const double *delta;          // [100]
  const double *y, *b, *c, *d;  // [50*50*100]
  double ampl;                  // sqrt(Smoothing.Variance[ismooth])
};
// interpolate_collapse_time (:1110-1126, 1219-1231): bilinear in (x, y) of four my_spline_eval's in delta
PF_HD double pf_interpolate_collapse_time(const pf_ct_view &t, double l1, double l2, double l3) {
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const double d = (l1 + l2 + l3) / t.ampl;
  const double x = (l1 - l2) / t.ampl;
  const double y = (l2 - l3) / t.ampl;
  int ix = (int)(x / bin_x);
  int iy = (int)(y / bin_x);
  ix = (ix >= PF_CT_NBINS_XY - 1) ? PF_CT_NBINS_XY - 2 : (ix < 0) ? 0 : ix;
  iy = (iy >= PF_CT_NBINS_XY - 1) ? PF_CT_NBINS_XY - 2 : (iy < 0) ? 0 : iy;
  const double dx = x / bin_x - ix;
  const double dy = y / bin_x - iy;
  // the four splines share their knots: one search (GSL's accelerator lookup returns the same interval)
  const double *xa = t.delta;
  const int last = PF_CT_NBINS_D - 1;
  int mode = 0, ilo = 0;  // 0 inside, -1 below, +1 above the knot range (my_spline_eval extrapolates linearly)
  if (d < xa[0]) mode = -1;
  else if (d > xa[last]) mode = 1;
  else {
    int ihi = last;
    while (ihi > ilo + 1) {
      const int i = (ihi + ilo) >> 1;
      if (xa[i] > d) ihi = i; else ilo = i;
    }
  }
  double s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int node = ((ix + (k & 1)) + (iy + (k >> 1)) * PF_CT_NBINS_XY) * PF_CT_NBINS_D;
    const double *ya = t.y + node;
    if (mode < 0) s[k] = ya[0] + (d - xa[0]) * (ya[1] - ya[0]) / (xa[1] - xa[0]);
    else if (mode > 0) s[k] = ya[last] + (d - xa[last]) * (ya[last] - ya[last - 1]) / (xa[last] - xa[last - 1]);
    else {
      const double delx = d - xa[ilo];
      s[k] = ya[ilo] + delx * (t.b[node + ilo] + delx * (t.c[node + ilo] + delx * t.d[node + ilo]));
    }
  }
  return ((1. - dx) * (1. - dy) * s[0] + (dx) * (1. - dy) * s[1] + (1. - dx) * (dy) * s[2] + (dx) * (dy) * s[3]);
}

// ---- the other two interpolation flavours the reference offers (tests/Readme_Pinocchio_tests_V5_1.txt): -DTRILINEAR and
// -DALL_SPLINE put their own return in front of the BILINEAR_SPLINE one (src/collapse_times.c:1153-1216)
enum { PF_CT_BILINEAR_SPLINE = 0, PF_CT_TRILINEAR = 1, PF_CT_ALL_SPLINE = 2 };
// my_spline_eval of the node spline (ix, iy) at d
PF_HD double pf_ct_node_eval(const pf_ct_view &t, int ix, int iy, double d) {
  const double *xa = t.delta;
  const int last = PF_CT_NBINS_D - 1;
  const int node = (ix + iy * PF_CT_NBINS_XY) * PF_CT_NBINS_D;
  const double *ya = t.y + node;
  if (d < xa[0]) return ya[0] + (d - xa[0]) * (ya[1] - ya[0]) / (xa[1] - xa[0]);
  if (d > xa[last]) return ya[last] + (d - xa[last]) * (ya[last] - ya[last - 1]) / (xa[last] - xa[last - 1]);
  int ilo = 0, ihi = last;
  while (ihi > ilo + 1) {
    const int i = (ihi + ilo) >> 1;
    if (xa[i] > d) ihi = i; else ilo = i;
  }
  const double delx = d - xa[ilo];
  return ya[ilo] + delx * (t.b[node + ilo] + delx * (t.c[node + ilo] + delx * t.d[node + ilo]));
}
// TRILINEAR (:1189-1216): the eight table entries around (d, x, y), no splines
PF_HD double pf_interpolate_trilinear(const pf_ct_view &t, double d, double x, double y, int ix, int iy) {
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const double *dv = t.delta, *T = t.y;
  int id;
  if (d <= dv[0]) id = 0;
  else if (d >= dv[PF_CT_NBINS_D - 1]) id = PF_CT_NBINS_D - 2;
  else {  // bsearch with compare_search (:1129-1135): dv[id] <= d < dv[id + 1]
    int lo = 0, hi = PF_CT_NBINS_D - 1;
    while (hi > lo + 1) { const int m = (hi + lo) >> 1; if (dv[m] > d) hi = m; else lo = m; }
    id = lo;
  }
  const double dd = (d - dv[id]) / (dv[id + 1] - dv[id]);
  const double dx = x / bin_x - ix;
  const double dy = y / bin_x - iy;
  const int D = PF_CT_NBINS_D, DX = PF_CT_NBINS_D * PF_CT_NBINS_XY;
  return (((1. - dd) * (1. - dx) * (1. - dy) * T[id + (ix)*D + (iy)*DX]) + ((dd) * (1. - dx) * (1. - dy) * T[(id + 1) + (ix)*D + (iy)*DX]) +
          ((1. - dd) * (dx) * (1. - dy) * T[id + (ix + 1) * D + (iy)*DX]) + ((dd) * (dx) * (1. - dy) * T[(id + 1) + (ix + 1) * D + (iy)*DX]) +
          ((1. - dd) * (1. - dx) * (dy)*T[id + (ix)*D + (iy + 1) * DX]) + ((dd) * (1. - dx) * (dy)*T[(id + 1) + (ix)*D + (iy + 1) * DX]) +
          ((1. - dd) * (dx) * (dy)*T[id + (ix + 1) * D + (iy + 1) * DX]) + ((dd) * (dx) * (dy)*T[(id + 1) + (ix + 1) * D + (iy + 1) * DX]));
}
// derivative at its four knots of the natural cubic spline through (xa, ya): GSL's cspline_init for size 4 (a 2 x 2
// symmetric system, solve_tridiag's LDL^t with its order of operations) and cspline_eval_deriv at each knot
PF_HD void pf_cspline4_node_derivs(const double xa[4], const double ya[4], double dz[4]) {
  const double h0 = xa[1] - xa[0], h1 = xa[2] - xa[1], h2 = xa[3] - xa[2];
  const double y0 = ya[1] - ya[0], y1 = ya[2] - ya[1], y2 = ya[3] - ya[2];
  const double r0 = (h0 != 0.0) ? 1.0 / h0 : 0.0, r1 = (h1 != 0.0) ? 1.0 / h1 : 0.0, r2 = (h2 != 0.0) ? 1.0 / h2 : 0.0;
  const double diag0 = 2.0 * (h1 + h0), diag1 = 2.0 * (h2 + h1);
  const double g0 = 3.0 * (y1 * r1 - y0 * r0), g1 = 3.0 * (y2 * r2 - y1 * r1);
  const double alpha0 = diag0, gamma0 = h1 / alpha0, alpha1 = diag1 - h1 * gamma0;
  const double z1 = g1 - gamma0 * g0;
  const double cc0 = g0 / alpha0, cc1 = z1 / alpha1;
  const double c[4] = {0.0, cc0 - gamma0 * cc1, cc1, 0.0};
  const double h[3] = {h0, h1, h2}, dy[3] = {y0, y1, y2};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 3; i++) dz[i] = (dy[i] / h[i]) - h[i] * (c[i + 1] + 2.0 * c[i]) / 3.0;  // delx = 0: b_i itself
  const double b2 = dz[2], d2 = (c[3] - c[2]) / (3.0 * h2);
  dz[3] = b2 + h2 * (2.0 * c[2] + 3.0 * d2 * h2);  // the last knot belongs to the last interval (gsl_interp_bsearch)
}
// ALL_SPLINE (:1153-1185): the sixteen node splines around the cell evaluated at d, then gsl_spline2d's bicubic on that
// 4 x 4 grid (GSL 2.7.1 interp2d/bicubic.c restated: bicubic_init's zx, zy, zxy from natural splines along rows and
// columns, bicubic_eval's Hermite patch).  Beyond the grid gsl_spline2d_eval raises GSL_EDOM (the reference aborts);
// here the edge cell's patch is evaluated, as gsl_spline2d_eval_extrap does.
PF_HD double pf_interpolate_all_spline(const pf_ct_view &t, double d, double x, double y, int ix, int iy) {
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const int ixstart = (ix == 0) ? 0 : (ix >= PF_CT_NBINS_XY - 2) ? PF_CT_NBINS_XY - 4 : ix - 1;
  const int iystart = (iy == 0) ? 0 : (iy >= PF_CT_NBINS_XY - 2) ? PF_CT_NBINS_XY - 4 : iy - 1;
  double xa[4], ya[4], za[16], zx[16], zy[16], zxy[16], u[4], dz[4];
  for (int k = 0; k < 4; k++) { xa[k] = (k + ixstart) * bin_x; ya[k] = (k + iystart) * bin_x; }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) za[j * 4 + i] = pf_ct_node_eval(t, i + ixstart, j + iystart, d);
  for (int j = 0; j < 4; j++) {
    pf_cspline4_node_derivs(xa, za + 4 * j, dz);
    for (int i = 0; i < 4; i++) zx[j * 4 + i] = dz[i];
  }
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) u[j] = za[j * 4 + i];
    pf_cspline4_node_derivs(ya, u, dz);
    for (int j = 0; j < 4; j++) zy[j * 4 + i] = dz[j];
  }
  for (int j = 0; j < 4; j++) {
    pf_cspline4_node_derivs(xa, zy + 4 * j, dz);
    for (int i = 0; i < 4; i++) zxy[j * 4 + i] = dz[i];
  }
  int xi = 0, yi = 0;
  { int lo = 0, hi = 3; while (hi > lo + 1) { const int m = (hi + lo) >> 1; if (xa[m] > x) hi = m; else lo = m; } xi = lo; }
  { int lo = 0, hi = 3; while (hi > lo + 1) { const int m = (hi + lo) >> 1; if (ya[m] > y) hi = m; else lo = m; } yi = lo; }
  const int i00 = yi * 4 + xi, i01 = (yi + 1) * 4 + xi, i10 = yi * 4 + xi + 1, i11 = (yi + 1) * 4 + xi + 1;  // [x][y]: min/max
  const double dx = xa[xi + 1] - xa[xi], dy = ya[yi + 1] - ya[yi];
  const double tt = (x - xa[xi]) / dx, uu = (y - ya[yi]) / dy;
  const double dt = 1. / dx, du = 1. / dy;
  const double zminmin = za[i00], zminmax = za[i01], zmaxmin = za[i10], zmaxmax = za[i11];
  const double zxminmin = zx[i00] / dt, zxminmax = zx[i01] / dt, zxmaxmin = zx[i10] / dt, zxmaxmax = zx[i11] / dt;
  const double zyminmin = zy[i00] / du, zyminmax = zy[i01] / du, zymaxmin = zy[i10] / du, zymaxmax = zy[i11] / du;
  const double zxyminmin = zxy[i00] / (dt * du), zxyminmax = zxy[i01] / (dt * du), zxymaxmin = zxy[i10] / (dt * du), zxymaxmax = zxy[i11] / (dt * du);
  const double t0 = 1, t1 = tt, t2 = tt * tt, t3 = tt * t2, u0 = 1, u1 = uu, u2 = uu * uu, u3 = uu * u2;
  double z = 0, w;
  w = zminmin; z += w * t0 * u0;
  w = zyminmin; z += w * t0 * u1;
  w = -3 * zminmin + 3 * zminmax - 2 * zyminmin - zyminmax; z += w * t0 * u2;
  w = 2 * zminmin - 2 * zminmax + zyminmin + zyminmax; z += w * t0 * u3;
  w = zxminmin; z += w * t1 * u0;
  w = zxyminmin; z += w * t1 * u1;
  w = -3 * zxminmin + 3 * zxminmax - 2 * zxyminmin - zxyminmax; z += w * t1 * u2;
  w = 2 * zxminmin - 2 * zxminmax + zxyminmin + zxyminmax; z += w * t1 * u3;
  w = -3 * zminmin + 3 * zmaxmin - 2 * zxminmin - zxmaxmin; z += w * t2 * u0;
  w = -3 * zyminmin + 3 * zymaxmin - 2 * zxyminmin - zxymaxmin; z += w * t2 * u1;
  w = 9 * zminmin - 9 * zmaxmin + 9 * zmaxmax - 9 * zminmax + 6 * zxminmin + 3 * zxmaxmin - 3 * zxmaxmax - 6 * zxminmax + 6 * zyminmin - 6 * zymaxmin -
      3 * zymaxmax + 3 * zyminmax + 4 * zxyminmin + 2 * zxymaxmin + zxymaxmax + 2 * zxyminmax;
  z += w * t2 * u2;
  w = -6 * zminmin + 6 * zmaxmin - 6 * zmaxmax + 6 * zminmax - 4 * zxminmin - 2 * zxmaxmin + 2 * zxmaxmax + 4 * zxminmax - 3 * zyminmin + 3 * zymaxmin +
      3 * zymaxmax - 3 * zyminmax - 2 * zxyminmin - zxymaxmin - zxymaxmax - 2 * zxyminmax;
  z += w * t2 * u3;
  w = 2 * zminmin - 2 * zmaxmin + zxminmin + zxmaxmin; z += w * t3 * u0;
  w = 2 * zyminmin - 2 * zymaxmin + zxyminmin + zxymaxmin; z += w * t3 * u1;
  w = -6 * zminmin + 6 * zmaxmin - 6 * zmaxmax + 6 * zminmax - 3 * zxminmin - 3 * zxmaxmin + 3 * zxmaxmax + 3 * zxminmax - 4 * zyminmin + 4 * zymaxmin +
      2 * zymaxmax - 2 * zyminmax - 2 * zxyminmin - 2 * zxymaxmin - zxymaxmax - zxyminmax;
  z += w * t3 * u2;
  w = 4 * zminmin - 4 * zmaxmin + 4 * zmaxmax - 4 * zminmax + 2 * zxminmin + 2 * zxmaxmin - 2 * zxmaxmax - 2 * zxminmax + 2 * zyminmin - 2 * zymaxmin -
      2 * zymaxmax + 2 * zyminmax + zxyminmin + zxymaxmin + zxymaxmax + zxyminmax;
  z += w * t3 * u3;
  return z;
}
// interpolate_collapse_time of a build with flavour FLAV
template <int FLAV> PF_HD double pf_interpolate_collapse_time_as(const pf_ct_view &t, double l1, double l2, double l3) {
  if (FLAV == PF_CT_BILINEAR_SPLINE) return pf_interpolate_collapse_time(t, l1, l2, l3);
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const double d = (l1 + l2 + l3) / t.ampl;
  const double x = (l1 - l2) / t.ampl;
  const double y = (l2 - l3) / t.ampl;
  int ix = (int)(x / bin_x);
  int iy = (int)(y / bin_x);
  ix = (ix >= PF_CT_NBINS_XY - 1) ? PF_CT_NBINS_XY - 2 : (ix < 0) ? 0 : ix;
  iy = (iy >= PF_CT_NBINS_XY - 1) ? PF_CT_NBINS_XY - 2 : (iy < 0) ? 0 : iy;
  return FLAV == PF_CT_TRILINEAR ? pf_interpolate_trilinear(t, d, x, y, ix, iy) : pf_interpolate_all_spline(t, d, x, y, ix, iy);
}

// GSL's cspline_init for one (x, y) node of the collapse-time table (gsl_spline_init, src/collapse_times.c:1037-1041):
// right-hand side, forward and back substitution with the shared LDL^t factors (pf_ct_tridiag), then the b, d that
// cspline_eval derives from c.  c doubles as the work array of the substitutions.
PF_HD void pf_ct_node_spline(const double *xa, const double *alpha, const double *gamma, const double *ya, double *c, double *b, double *d) {
  const int n = PF_CT_NBINS_D, sys = n - 2;
  double *z = c + 1;  // the interior unknowns live in c[1..n-2]
  for (int i = 0; i < sys; i++) {
    const double h_i = xa[i + 1] - xa[i], h_ip1 = xa[i + 2] - xa[i + 1];
    const double ydiff_i = ya[i + 1] - ya[i], ydiff_ip1 = ya[i + 2] - ya[i + 1];
    const double g_i = (h_i != 0.0) ? 1.0 / h_i : 0.0, g_ip1 = (h_ip1 != 0.0) ? 1.0 / h_ip1 : 0.0;
    const double g = 3.0 * (ydiff_ip1 * g_ip1 - ydiff_i * g_i);
    z[i] = (i == 0) ? g : g - gamma[i - 1] * z[i - 1];
  }
  for (int i = 0; i < sys; i++) z[i] = z[i] / alpha[i];
  for (int i = sys - 2; i >= 0; i--) z[i] = z[i] - gamma[i] * z[i + 1];
  c[0] = 0.0; c[n - 1] = 0.0;
  for (int i = 0; i + 1 < n; i++) {
    const double dx = xa[i + 1] - xa[i], dy = ya[i + 1] - ya[i];
    b[i] = (dy / dx) - dx * (c[i + 1] + 2.0 * c[i]) / 3.0;
    d[i] = (c[i + 1] - c[i]) / (3.0 * dx);
  }
  b[n - 1] = d[n - 1] = 0.0;
}

// host side of the TABULATED_CT splines: every node's spline has the same knots, so the LDL^t factors of GSL's
// tridiagonal solve (alpha, gamma of solve_tridiag) are computed once here; k_ct_splines does the two substitutions
// per node with the same operations in the same order as cspline_init.  n knots -> n - 2 entries each.
inline void pf_ct_tridiag(const double *xa, int n, double *alpha, double *gamma) {
  const int sys = n - 2;
  for (int i = 0; i < sys; i++) {
    const double h_i = xa[i + 1] - xa[i], h_ip1 = xa[i + 2] - xa[i + 1];
    const double diag_i = 2.0 * (h_ip1 + h_i);
    if (i == 0) alpha[0] = diag_i;
    else alpha[i] = diag_i - (xa[i + 1] - xa[i]) * gamma[i - 1];  // offdiag[i-1] = h_i
    gamma[i] = h_ip1 / alpha[i];                                  // offdiag[i] = h_ip1 (unused for the last row)
  }
}
// the sampling in delta of initialize_collapse_times (src/collapse_times.c:836-876; CT_EXPO = 1.75, CT_SQUEEZE = 1.2,
// CT_RANGE_D = 7, CT_DELTA0 = -1): finest around CT_DELTA0.  Host libm, like the reference.
inline void pf_ct_delta_vector(double *delta_vector) {
  const double CT_SQUEEZE = 1.2, CT_EXPO = 1.75, CT_RANGE_D = 7.0, CT_DELTA0 = -1.0;
  const double deltaf = pow(CT_SQUEEZE / CT_EXPO, 1. / (CT_EXPO - 1.));
  const double ref_interval = ((pow(CT_RANGE_D - CT_DELTA0, 2. - CT_EXPO) + pow(CT_RANGE_D + CT_DELTA0, 2. - CT_EXPO)
                                - 2. * pow(deltaf, 2. - CT_EXPO)) / CT_EXPO / (2. - CT_EXPO) + 2. * deltaf / CT_SQUEEZE) / (PF_CT_NBINS_D - 2.0);
  double del = -CT_RANGE_D;
  for (int id = 0; id < PF_CT_NBINS_D; id++) {
    delta_vector[id] = del;
    double interval = CT_EXPO * ref_interval * pow(fabs(del - CT_DELTA0), CT_EXPO - 1.0);
    interval = (interval / ref_interval < CT_SQUEEZE ? ref_interval * CT_SQUEEZE : interval);
    del += interval;
  }
}

// host-side: natural cubic spline coefficients exactly as GSL's cspline_init
// (tridiagonal LDL^t, linalg/tridiag.c solve_tridiag).  c must hold n doubles.
inline int pf_spline_coeffs(const double *xa, const double *ya, int n, double *c) {
  if (n < 3) return 1;
  const int max_index = n - 1, sys = max_index - 1;
  c[0] = 0.0; c[max_index] = 0.0;
  double *g = new double[5 * sys];
  double *diag = g + sys, *off = g + 2 * sys, *alpha = g + 3 * sys, *gamma = g + 4 * sys;
  for (int i = 0; i < sys; i++) {
    const double h_i = xa[i + 1] - xa[i], h_ip1 = xa[i + 2] - xa[i + 1];
    const double yd_i = ya[i + 1] - ya[i], yd_ip1 = ya[i + 2] - ya[i + 1];
    const double g_i = (h_i != 0.0) ? 1.0 / h_i : 0.0, g_ip1 = (h_ip1 != 0.0) ? 1.0 / h_ip1 : 0.0;
    off[i] = h_ip1;
    diag[i] = 2.0 * (h_ip1 + h_i);
    g[i] = 3.0 * (yd_ip1 * g_ip1 - yd_i * g_i);
  }
  if (sys == 1) {
    c[1] = g[0] / diag[0];
  } else {
    alpha[0] = diag[0];
    gamma[0] = off[0] / alpha[0];
    for (int i = 1; i < sys - 1; i++) {
      alpha[i] = diag[i] - off[i - 1] * gamma[i - 1];
      gamma[i] = off[i] / alpha[i];
    }
    alpha[sys - 1] = diag[sys - 1] - off[sys - 2] * gamma[sys - 2];
    // forward substitution (z overwrites g), scaling, back substitution
    for (int i = 1; i < sys; i++) g[i] = g[i] - gamma[i - 1] * g[i - 1];
    for (int i = 0; i < sys; i++) g[i] = g[i] / alpha[i];
    double *x = c + 1;
    x[sys - 1] = g[sys - 1];
    for (int i = sys - 2; i >= 0; i--) x[i] = g[i] - gamma[i] * x[i + 1];
  }
  delete[] g;
  return 0;
}

// The translation units built with -ffp-contract=on (the transform kernels) get their default back for the code that
// follows this header, whatever the include order (the Makefile defines PF_FP_CONTRACT_ON for them).
#if defined(__clang__) && defined(PF_FP_CONTRACT_ON)
#pragma clang fp contract(on)
#endif
