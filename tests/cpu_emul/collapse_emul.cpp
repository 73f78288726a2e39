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
extern "C" int emul_collapse_fast(const double *sx, const double *sy, int nk, const double *d, long count,
                                  double *F, double *lam) {
  std::vector<double> c(nk), b(nk), dd(nk);
  if (pf_spline_coeffs(sx, sy, nk, c.data())) return 1;
  pf_spline_bd(sx, sy, c.data(), nk, b.data(), dd.data());
  pf_spline_view s{sx, sy, c.data(), b.data(), dd.data(), nk};
  for (long i = 0; i < count; i++) F[i] = pf_inverse_collapse_time<true>(d + 6 * i, s, lam + 3 * i);
  return 0;
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
