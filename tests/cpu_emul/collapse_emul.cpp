// Host-side unit-test driver for pinocchio_amd/csrc/pf_collapse_core.h.
// TEST ONLY -- not a CPU path of the library.
#include "../../pinocchio_amd/csrc/pf_collapse_core.h"
#include <vector>

extern "C" int emul_collapse(const double *sx, const double *sy, int nk, const double *d, long count,
                             double *F, double *lam) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  pf_spline_view s{sx, sy, c.data(), b.data(), dd.data(), nk};
  for (long i = 0; i < count; i++) F[i] = pf_inverse_collapse_time(d + 6 * i, s, lam + 3 * i);
  return 0;
}
// the fast flavour as the kernels run it: the inverse growing mode from the polynomial table of the spline (pf_gtab.h) when
// `with_table` and the table is accepted, the series forms otherwise.  Returns 0, or 2 when the table was built and used.
extern "C" int emul_collapse_fast(const double *sx, const double *sy, int nk, const double *d, long count,
                                  double *F, double *lam, int with_table) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  pf_spline_view s{sx, sy, c.data(), b.data(), dd.data(), nk};
  std::vector<double> g(PF_GT_HEADER + (PF_GT_MAX_INT + 1) * PF_GT_REC);
  std::vector<unsigned short> lut(PF_GT_MAX_BINS);
  const bool tab = with_table && pf_gtab_build(sx, sy, c.data(), b.data(), dd.data(), nk, g.data(), lut.data()) == 0;
  if (tab) { s.gt.rec = g.data() + PF_GT_HEADER; s.gt.lut = lut.data(); s.gt.bin0 = (unsigned)g[2]; s.gt.lo_all = g[3]; s.gt.hi_all = g[4]; }
  for (long i = 0; i < count; i++) F[i] = pf_inverse_collapse_time<true>(d + 6 * i, s, lam + 3 * i);
  return tab ? 2 : 0;
}
// the table alone: Y[i] = 10^(-S(log10 D[i])) from the table (NaN where D lies outside it); info = the table's header
// (nint, nbins, bin0, lo_all, hi_all, max_rel_err, valid, 0); bcd = the cspline's c, b, d (3 * nk) for an independent check
extern "C" int emul_gtab(const double *sx, const double *sy, int nk, const double *D, long count, double *Y, double *info, double *bcd) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  std::vector<double> g(PF_GT_HEADER + (PF_GT_MAX_INT + 1) * PF_GT_REC);
  std::vector<unsigned short> lut(PF_GT_MAX_BINS);
  const int rc = pf_gtab_build(sx, sy, c.data(), b.data(), dd.data(), nk, g.data(), lut.data());
  for (int i = 0; i < PF_GT_HEADER; i++) info[i] = g[i];
  for (int i = 0; i < nk; i++) { bcd[i] = c[i]; bcd[nk + i] = b[i]; bcd[2 * nk + i] = dd[i]; }
  pf_gtab_view v;
  v.rec = g.data() + PF_GT_HEADER; v.lut = lut.data(); v.bin0 = (unsigned)g[2]; v.lo_all = g[3]; v.hi_all = g[4];
  for (long i = 0; i < count; i++) {
    double y;
    Y[i] = (rc == 0 && pf_gtab_eval(v, D[i], y)) ? y : NAN;
  }
  return rc;
}
extern "C" double emul_ell_classic(double a, double b, double c) { return pf_ell_classic(a, b, c); }
extern "C" int emul_spline(const double *sx, const double *sy, int nk, const double *v, long count, double *out) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  pf_spline_view s{sx, sy, c.data(), b.data(), dd.data(), nk};
  for (long i = 0; i < count; i++) out[i] = pf_spline_eval(s, v[i]);
  return 0;
}

// TABULATED_CT pieces of the header: delta sampling, shared tridiagonal factors, node splines, interpolation
extern "C" int emul_ct(const double *table /*[50*50*100]*/, double ampl, const double *lam /*[3*count]*/, long count,
                       double *delta_out /*[100]*/, double *F, int flavour /* 0 BILINEAR_SPLINE, 1 TRILINEAR, 2 ALL_SPLINE */) {
  const int nd = PF_CT_NBINS_D, nn = PF_CT_NBINS_XY * PF_CT_NBINS_XY;
  std::vector<double> delta(nd), alpha(nd), gamma(nd), c((size_t)nn * nd), b((size_t)nn * nd), d((size_t)nn * nd);
  pf_ct_delta_vector(delta.data());
  pf_ct_tridiag(delta.data(), nd, alpha.data(), gamma.data());
  for (int node = 0; node < nn; node++)
    pf_ct_node_spline(delta.data(), alpha.data(), gamma.data(), table + (size_t)node * nd, c.data() + (size_t)node * nd,
                      b.data() + (size_t)node * nd, d.data() + (size_t)node * nd);
  pf_ct_view t{delta.data(), table, b.data(), c.data(), d.data(), ampl};
  for (long i = 0; i < count; i++) {
    const double *l = lam + 3 * i;
    F[i] = flavour == 1   ? pf_interpolate_collapse_time_as<PF_CT_TRILINEAR>(t, l[0], l[1], l[2])
           : flavour == 2 ? pf_interpolate_collapse_time_as<PF_CT_ALL_SPLINE>(t, l[0], l[1], l[2])
                          : pf_interpolate_collapse_time_as<PF_CT_BILINEAR_SPLINE>(t, l[0], l[1], l[2]);
  }
  for (int i = 0; i < nd; i++) delta_out[i] = delta[i];
  return 0;
}

// ELL_SNG model of the device header (pf_sng_core.h)
#include "../../pinocchio_amd/csrc/pf_sng_core.h"
extern "C" void emul_ell_sng(const double *lam, long count, double D_in, const double *cosmo, double *bc) {
  pf_sng_cosmo c{cosmo[0], cosmo[1], cosmo[2], cosmo[3], cosmo[4], cosmo[5], cosmo[6]};
  for (long i = 0; i < count; i++) bc[i] = pf_ell_sng(lam[3 * i], lam[3 * i + 1], lam[3 * i + 2], D_in, c);
}

// spline evaluation through the interval-search start table (what k_collapse uses) -- must find the same interval
extern "C" int emul_spline_lut(const double *sx, const double *sy, int nk, const double *v, long count, double *out) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  // the table exactly as the kernel prologue builds it (pf_collapse_body): direct form unless a bin holds two knots
  std::vector<unsigned short> lut(PF_SPLINE_LUT_BINS);
  double x0, inv_w;
  bool direct = true;
  pf_spline_lut_geometry(sx, nk, true, x0, inv_w);
  for (int i = 0; i < PF_SPLINE_LUT_BINS; i++) lut[i] = pf_spline_lut_entry(sx, nk, i, x0, inv_w, true);
  for (int i = 0; i + 1 < PF_SPLINE_LUT_BINS; i++) if ((int)lut[i + 1] - (int)lut[i] > 1) direct = false;
  if (!direct) {
    pf_spline_lut_geometry(sx, nk, false, x0, inv_w);
    for (int i = 0; i < PF_SPLINE_LUT_BINS; i++) lut[i] = pf_spline_lut_entry(sx, nk, i, x0, inv_w, false);
  }
  pf_spline_view s{sx, sy, c.data(), b.data(), dd.data(), nk};
  s.lut = lut.data();
  s.lut_inv_w = inv_w; s.lut_x0 = x0; s.lut_direct = direct ? 1 : 0; s.x_first = sx[0]; s.x_last = sx[nk - 1];
  for (long i = 0; i < count; i++) out[i] = pf_spline_eval(s, v[i]);
  return direct ? 2 : 0;  // (0 / 2: which form of the table was used; 1 is an error)
}

extern "C" void emul_div_const(const double *x, long count, double *q9, double *q54) {
  for (long i = 0; i < count; i++) { q9[i] = pf_div_const<9>(x[i]); q54[i] = pf_div_const<54>(x[i]); }
}

extern "C" void emul_log10(const double *x, long count, double *out) {
  for (long i = 0; i < count; i++) out[i] = pf_log10_pos(x[i]);
}

// the per-cell reductions the z-pass kernels share with the cell kernels
extern "C" void emul_invariants(const double *d6, long count, double *mu3, double *lam3, int *ok) {
  for (long i = 0; i < count; i++) {
    pf_invariants(d6 + 6 * i, mu3[3 * i], mu3[3 * i + 1], mu3[3 * i + 2]);
    const double third = mu3[3 * i] * (1.0 / 3.0), diag[3] = {third, third, third};
    ok[i] = pf_eigen_from_invariants<false>(mu3[3 * i], mu3[3 * i + 1], mu3[3 * i + 2], diag, lam3 + 3 * i) ? 1 : 0;
  }
}
extern "C" void emul_lpt3b(const double *s, const double *phi2, const double *h, long count, double *out) {
  for (long i = 0; i < count; i++) out[i] = pf_lpt3b_accumulate(s[i], phi2 + 6 * i, h + 6 * i);
}

extern "C" void emul_pow_third(const double *x, long count, double *out) {
  for (long i = 0; i < count; i++) out[i] = pf_pow_third<true>(x[i]);
}

extern "C" void emul_exp(const double *x, long count, double *e, double *e10) {
  for (long i = 0; i < count; i++) { e[i] = pf_exp_series(x[i]); e10[i] = pf_exp10_series(x[i]); }
}

// the cosine triple of the trigonometric root formula (fast flavour: no acos, no sincos)
extern "C" void emul_cos3(const double *x, long count, double *c) {
  for (long i = 0; i < count; i++) pf_cos3_of_acos(x[i], c[3 * i], c[3 * i + 1], c[3 * i + 2]);
}
// ... and its table form (pf_c3tab.h), as the cell kernels call it
extern "C" void emul_cos3_tab(const double *x, long count, double *c) {
  for (long i = 0; i < count; i++) pf_cos3_fast(pf_c3_tab, x[i], c[3 * i], c[3 * i + 1], c[3 * i + 2]);
}
