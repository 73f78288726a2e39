// pf_gtab.h -- the inverse growing mode of the fast flavour as ONE table of polynomials in D.
//
// ell() ends with InverseGrowingMode(b_c) = 1 / pow(10., my_spline_eval(log10(b_c))) - 1 (src/collapse_times.c:404-427,
// src/cosmo.c:1822-1832, 2016-2027): a logarithm, a spline lookup and a power per cell -- ~90 of the ~500 fp64 instructions
// of the collapse solve even in their series forms.  The composite Y(D) = 10^(-S(log10 D)) is a smooth function of D on
// every interval between two knots of the spline S (where S is one cubic in log10 D) and a power law beyond the last knot
// (my_spline_eval extrapolates linearly): on each such interval it is approximated here by its degree-7 Chebyshev
// interpolant in D itself, fitted on the host in long double arithmetic from the very knots and cspline coefficients the
// reference evaluates.  Intervals: the knot intervals [10^x_j, 10^x_{j+1}) (split once where the fit asks for it), then,
// beyond the last knot and up to 2^10, sixteen or more per octave in geometric progression.  The truncation error of the interpolant
// is ~1e-15 relative on the knot intervals and, on the power law D^-p beyond them, grows with p (p = 3.8 for a LCDM table
// that ends at a = 1.5: twenty per octave for 9e-15); the table is checked against the long double composite at build time and refused beyond
// 2e-14 -- the reference's own chain log10 -> spline -> pow carries ~4e-15 of rounding -- in which case, as for D outside
// the table, the solve takes the series forms of pf_collapse_core.h.  The interval of D comes from the bits of D: 64 bins per octave, a start table
// and two comparisons with the next intervals' lower edges (no bin holds more than two edges: checked at build time).
// A cell on the "wrong" side of a knot by one rounding of log10 changes nothing visible: S is C^2 at its knots.
//
// Also compiled for the host by tests/cpu_emul (unit test against mpmath / long double); not a CPU path of the library.
#pragma once
#include <math.h>
#include <string.h>

#ifndef PF_HD
#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif
#endif

#define PF_GT_DEG 7
#define PF_GT_REC 10        /* doubles per interval: lower edge, 2 / width, c0 .. c7 (powers of u = (D - lo) * scale - 1) */
#define PF_GT_MAX_INT 432   /* intervals (a sentinel record with the table's upper end follows the last) */
#define PF_GT_LUT_BITS 6    /* start table: 64 bins per octave */
#define PF_GT_MAX_BINS 2048
#define PF_GT_ZONE_K 16     /* intervals per octave beyond the last knot, at least (more where the power law there is steep) */
#define PF_GT_HI_EXP 10     /* the table ends at the first zone edge at or above 2^10 */
#define PF_GT_ACCEPT 2e-14L /* largest relative error of an accepted table (a stored fp32 Fmax then differs on ~2 err / 2^-24 = 7e-7 of the cells at most) */
#define PF_GT_HEADER 8      /* doubles in front of the records: nint, nbins, bin0, lo_all, hi_all, max_rel_err, valid, 0 */

struct pf_gtab_view {
  const double *rec = nullptr;          // (nint + 1) * PF_GT_REC
  const unsigned short *lut = nullptr;  // nbins
  unsigned bin0 = 0;
  double lo_all = 0.0, hi_all = 0.0;
};

PF_HD unsigned pf_gtab_bits(double D) {
  unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = (unsigned long long)__double_as_longlong(D);
#else
  memcpy(&b, &D, sizeof(b));
#endif
  return (unsigned)(b >> (52 - PF_GT_LUT_BITS));
}

// Y = 10^(-S(log10 D)) for lo_all <= D < hi_all; false outside (and for NaN)
PF_HD bool pf_gtab_eval(const pf_gtab_view &g, double D, double &Y) {
  if (!(D >= g.lo_all && D < g.hi_all)) return false;
  const unsigned bin = pf_gtab_bits(D) - g.bin0;
  int j = g.lut[bin];
  if (D >= g.rec[(j + 1) * PF_GT_REC]) j++;
  if (D >= g.rec[(j + 1) * PF_GT_REC]) j++;  // (a knot interval that needed splitting: a bin may hold two edges)
  const double *r = g.rec + j * PF_GT_REC;
  const double u = fma(D - r[0], r[1], -1.0);
  double p = r[9];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 8; k >= 2; k--) p = fma(p, u, r[k]);
  Y = p;
  return true;
}

// ---- host side (plain host functions; parsed but not emitted in the device pass): the table of one spline (knots xa, ya, cspline c and the b, d of pf_spline_bd; n knots) ----
// out: PF_GT_HEADER + (PF_GT_MAX_INT + 1) * PF_GT_REC doubles; lut: PF_GT_MAX_BINS entries.  Returns 0 and sets out[6] = 1
// when the table is usable; otherwise out[6] = 0 (too many knots, knots too dense for the start table, or the check failed).
static inline long double pf_gt_truth(const double *xa, const double *ya, const double *ca, const double *ba, const double *da, int n, int j, long double D) {
  const long double x = log10l(D);
  long double S;
  if (j >= n - 1) {  // beyond the last knot: my_spline_eval's linear extrapolation
    const long double slope = ((long double)ya[n - 1] - (long double)ya[n - 2]) / ((long double)xa[n - 1] - (long double)xa[n - 2]);
    S = (long double)ya[n - 1] + (x - (long double)xa[n - 1]) * slope;
  } else {
    const long double dx = x - (long double)xa[j];
    S = (long double)ya[j] + dx * ((long double)ba[j] + dx * ((long double)ca[j] + dx * (long double)da[j]));
  }
  return powl(10.0L, -S);
}
static inline int pf_gtab_build(const double *xa, const double *ya, const double *ca, const double *ba, const double *da, int n, double *out, unsigned short *lut) {
  const int NC = PF_GT_DEG + 1;
  memset(out, 0, sizeof(double) * (PF_GT_HEADER + (PF_GT_MAX_INT + 1) * PF_GT_REC));
  memset(lut, 0, sizeof(unsigned short) * PF_GT_MAX_BINS);
  if (n < 3) return 1;
  // interval edges: the knots in D, then the zone beyond the last knot
  static thread_local double edge[PF_GT_MAX_INT + 2];
  static thread_local int piece[PF_GT_MAX_INT + 2];  // which piece of S an interval lies on (n - 1: the extrapolation)
  int ne = 0;
  for (int j = 0; j < n; j++) {
    if (ne > PF_GT_MAX_INT) return 1;
    edge[ne] = (double)powl(10.0L, (long double)xa[j]); piece[ne] = j; ne++;
    if (j && !(edge[ne - 1] > edge[ne - 2])) return 1;
  }
  double *rec = out + PF_GT_HEADER;
  long double worst = 0.0L;
  // fit interval i of the current edge list; returns its largest relative error against the long double composite
  auto fit = [&](int i, double *r) -> long double {
    const int j = piece[i];
    const long double lo = edge[i], hi = edge[i + 1];
    const double scale = (double)(2.0L / (hi - lo));
    // Chebyshev interpolation at NC nodes in u (the u the device forms: (D - lo) * scale - 1 with the rounded scale)
    long double f[PF_GT_DEG + 1], a[PF_GT_DEG + 1], mono[PF_GT_DEG + 1];
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < NC; k++) {
      const long double u = cosl(pi * (k + 0.5L) / NC);
      const long double D = lo + (u + 1.0L) / (long double)scale;
      f[k] = pf_gt_truth(xa, ya, ca, ba, da, n, j, D);
    }
    for (int m = 0; m < NC; m++) {
      long double sm = 0.0L;
      for (int k = 0; k < NC; k++) sm += f[k] * cosl(pi * m * (k + 0.5L) / NC);
      a[m] = sm * 2.0L / NC;
    }
    a[0] *= 0.5L;
    // Chebyshev -> monomials: T_0 = 1, T_1 = u, T_{m+1} = 2 u T_m - T_{m-1}
    long double T0[PF_GT_DEG + 1] = {1.0L}, T1[PF_GT_DEG + 1] = {0.0L, 1.0L}, T2[PF_GT_DEG + 1];
    for (int k = 0; k < NC; k++) mono[k] = 0.0L;
    mono[0] += a[0];
    for (int k = 0; k < NC; k++) mono[k] += a[1] * T1[k];
    for (int m = 2; m < NC; m++) {
      for (int k = 0; k < NC; k++) T2[k] = (k ? 2.0L * T1[k - 1] : 0.0L) - T0[k];
      for (int k = 0; k < NC; k++) mono[k] += a[m] * T2[k];
      for (int k = 0; k < NC; k++) { T0[k] = T1[k]; T1[k] = T2[k]; }
    }
    r[0] = (double)lo; r[1] = scale;
    for (int k = 0; k < NC; k++) r[2 + k] = (double)mono[k];
    // the check: the device's own evaluation against the long double composite
    long double w = 0.0L;
    for (int t = 0; t <= 32; t++) {
      double D = (double)(lo + (hi - lo) * (long double)t / 32.0L);
      if (t == 32) D = nextafter((double)hi, 0.0);
      const double u = fma(D - r[0], r[1], -1.0);
      double p = r[9];
      for (int k = 8; k >= 2; k--) p = fma(p, u, r[k]);
      const long double want = pf_gt_truth(xa, ya, ca, ba, da, n, j, (long double)D);
      const long double err = fabsl(((long double)p - want) / want);
      if (err > w) w = err;
    }
    return w;
  };
  {  // beyond the last knot: K intervals per octave in geometric progression.  On the power law D^-p there every interval has
     // the same relative error, growing with p: the smallest K of the list that brings it below 8e-15 and fits is taken
    const double last = edge[ne - 1], top = ldexp(1.0, PF_GT_HI_EXP);
    if (!(last < top) || ne + 2 > PF_GT_MAX_INT) return 1;
    const double octaves = log2(top / last);
    static const int ks[] = {PF_GT_ZONE_K, 20, 24, 28, 32, 40, 48};
    int K = PF_GT_ZONE_K;
    double tmp[PF_GT_REC];
    for (int q = 0; q < 7; q++) {
      if (ks[q] < K) continue;
      if (ne + (int)ceil(octaves * ks[q]) + 12 > PF_GT_MAX_INT) break;  // (12: room for split knot intervals)
      K = ks[q];
      edge[ne] = (double)((long double)last * powl(2.0L, 1.0L / K)); piece[ne] = n - 1;
      piece[ne - 1] = n - 1;
      const long double e = fit(ne - 1, tmp);
      if (e <= 8e-15L) break;
    }
    for (int k = 1;; k++) {
      if (ne > PF_GT_MAX_INT) return 1;
      const double v = (double)((long double)last * powl(2.0L, (long double)k / K));
      edge[ne] = v; piece[ne] = n - 1; ne++;
      if (v >= top) break;
    }
  }
  // a knot interval on which S wiggles (the ends of a natural spline) is split once, at its geometric middle
  {
    double tmp[PF_GT_REC];
    for (int i = 0; i + 1 < ne; i++) {
      if (piece[i] >= n - 1) break;  // the zone beyond the knots is a power law: never needed
      if (fit(i, tmp) <= 3e-15L || ne > PF_GT_MAX_INT) continue;
      for (int k = ne; k > i + 1; k--) { edge[k] = edge[k - 1]; piece[k] = piece[k - 1]; }
      edge[i + 1] = sqrt(edge[i] * edge[i + 2]); piece[i + 1] = piece[i];
      ne++;
      i++;  // (both halves stay as they are)
    }
  }
  const int nint = ne - 1;
  if (nint < 1 || nint > PF_GT_MAX_INT) return 1;
  for (int i = 0; i < nint; i++) {
    const long double e = fit(i, rec + (size_t)i * PF_GT_REC);
    if (e > worst) worst = e;
  }
  rec[(size_t)nint * PF_GT_REC] = edge[nint];  // sentinel: the upper end
  // start table: the interval that holds the lower edge of each bin; no bin may hold two interval edges
  const unsigned bin0 = pf_gtab_bits(edge[0]);
  const unsigned binl = pf_gtab_bits(nextafter(edge[nint], 0.0));
  const int nbins = (int)(binl - bin0) + 1;
  if (nbins < 1 || nbins > PF_GT_MAX_BINS) return 1;
  {
    int j = 0;
    for (int b = 0; b < nbins; b++) {
      unsigned long long bits = (unsigned long long)(bin0 + (unsigned)b) << (52 - PF_GT_LUT_BITS);
      double lowedge;
      memcpy(&lowedge, &bits, sizeof(lowedge));
      if (b == 0) lowedge = edge[0];  // (the first bin starts below the first edge; D >= edge[0] there)
      while (j + 1 < nint && edge[j + 1] <= lowedge) j++;
      lut[b] = (unsigned short)j;
      unsigned long long nb = (unsigned long long)(bin0 + (unsigned)b + 1) << (52 - PF_GT_LUT_BITS);
      double upedge;
      memcpy(&upedge, &nb, sizeof(upedge));
      int inside = 0;
      for (int q = j + 1; q <= nint && edge[q] < upedge; q++) inside++;
      if (inside > 2) return 1;  // knots denser than the bins
    }
  }
  out[0] = nint; out[1] = nbins; out[2] = (double)bin0; out[3] = edge[0]; out[4] = edge[nint]; out[5] = (double)worst;
  out[6] = worst <= PF_GT_ACCEPT ? 1.0 : 0.0;
  return out[6] != 0.0 ? 0 : 1;
}
