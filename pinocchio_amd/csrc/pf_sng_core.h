// pf_sng_core.h -- ELL_SNG collapse model (src/collapse_times.c:222-400): the nine-equation system of Nadkarni-Ghosh &
// Singhal (2016) for the eigenvalues of the deformation, velocity-derivative and gravity tensors of a homogeneous
// ellipsoid, integrated in the scale factor from a = 1e-5 with an adaptive Runge-Kutta-Fehlberg (4,5) scheme until the
// first axis collapses.  The reference drives GSL's gsl_odeiv2_step_rkf45 / control_standard_new(1e-6, 1e-6, 1, 1) /
// evolve_apply; the same step, error control and accept/reject logic are written out here (GSL 2.7.1
// ode-initval2/rkf45.c, cstd.c, evolve.c), one integration per thread -- a batch of 250 000 independent ODE
// problems per smoothing radius when it fills the TABULATED_CT table, which is how the reference uses this model.
//
// Also compiled for the host by tests/cpu_emul/collapse_emul.cpp (unit test against the oracle).
#pragma once
#include <float.h>
#include <math.h>

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

struct pf_sng_cosmo {
  double Omega0, OmegaLambda, OmegaRad, OmegaK;  // Hubble(z), src/cosmo.c:1691-1711 with params.simpleLambda
  // MOD_GRAV_FR (src/collapse_times.c:295-312): |f_R0| of the build (0 = standard gravity), 100 / c (src/cosmo.c:109)
  // and the smoothing radius the reference hands to the ODE system as its parameter (:378-388)
  double FR0 = 0.0, H_over_c = 0.0, size = 0.0;
};

// Thin-shell enhancement of the force on a top-hat of radius c.size in Hu-Sawicki f(R) gravity, in [0, 1/3]: the shell
// fraction x from the contrast of the background and interior scalaron, then (1 - (1 - x)^3) / 3 clipped to the fully
// unscreened 1/3.  (What ForceModification does, src/collapse_times.c:295-312; the operation order is the reference's,
// because the table built from it is compared node by node.)
PF_HD double pf_sng_fifth_force(const pf_sng_cosmo &c, double a, double overdensity) {
  const double lam_over_m = 4. * c.OmegaLambda / c.Omega0;
  const double background = pow((1.0 + lam_over_m) / (1.0 + lam_over_m * pow(a, 3.)), 2.0);
  const double interior = pow((1.0 + lam_over_m) / (1.0 + overdensity + lam_over_m * pow(a, 3.)), 2.0);
  const double shell = c.FR0 / c.Omega0 / pow(c.H_over_c * c.size, 2.0) * pow(a, 7.) * pow((1. + overdensity), -1. / 3.) * (background - interior);
  const double unscreened = shell * (3. + shell * (-3. + shell));
  if (unscreened < 0.) return 0.;
  return unscreened < 1. ? unscreened / 3. : 1. / 3;
}

PF_HD double pf_sng_Esq(const pf_sng_cosmo &c, double z) {
  return c.OmegaRad * pow(1. + z, 4.) + c.Omega0 * pow(1. + z, 3.) + c.OmegaK * pow(1. + z, 2.) + c.OmegaLambda;
}
// OmegaMatter(z), OmegaLambda(z) (src/cosmo.c:1675-1689) through Ez = Hubble(z) / Hubble(0) (:1713-1718)
PF_HD void pf_sng_omegas(const pf_sng_cosmo &c, double z, double &omegam, double &omegal) {
  const double H0 = 100. * sqrt(pf_sng_Esq(c, 0.0));
  const double Ezv = 100. * sqrt(pf_sng_Esq(c, z)) / H0;
  omegam = c.Omega0 * pow(1. + z, 3.) / (Ezv * Ezv);
  omegal = c.OmegaLambda / (Ezv * Ezv);
}

// Right-hand side of the ellipsoid's nine equations in the scale factor a (what sng_system evaluates,
// src/collapse_times.c:241-293): state s = {axis deformations lambda_a[3], velocity eigenvalues lambda_v[3], density
// (gravity) eigenvalues lambda_d[3]}.  Each derivative is assembled from the shared pieces below with the reference's
// own association of every sum and product (the integration is compared bit for bit with the oracle's).
PF_HD void pf_sng_rhs(double a, const double s[9], double ds[9], const pf_sng_cosmo &c) {
  double om, ol;
  pf_sng_omegas(c, 1. / a - 1., om, ol);
  const double *ax = s, *vel = s + 3, *den = s + 6;
  const double overdensity = den[0] + den[1] + den[2];
  const double trv = vel[0] + vel[1] + vel[2];
  const double expansion = (3. + vel[0] + vel[1] + vel[2]) - (1. + overdensity) / (2.5 + overdensity) * trv;
  const double drag = om - 2.0 * ol - 2.0;
  // gravity source: 3 Omega_m lambda_d, times (1 + fifth force) in an f(R) build (x 1.0 leaves the value as it is)
  const double boost = c.FR0 != 0.0 ? (1. + pf_sng_fifth_force(c, a, overdensity)) : 1.0;
  double sq[3];  // (1 - lambda_a)^2: the squared axis ratios
  for (int k = 0; k < 3; k++) sq[k] = (1. - ax[k]) * (1. - ax[k]);
  for (int k = 0; k < 3; k++) {
    double tide = 0.;  // coupling of axis k to the two others; an equal pair contributes nothing
    for (int m = 0; m < 3; m++)
      if (m != k && ax[k] != ax[m]) tide += (den[m] - den[k]) * (sq[k] * (1. + vel[k]) - sq[m] * (1. + vel[m])) / (sq[k] - sq[m]);
    ds[k] = (vel[k] * (ax[k] - 1.0)) / a;
    ds[k + 3] = (0.5 * (vel[k] * drag - 3.0 * om * den[k] * boost - 2.0 * vel[k] * vel[k])) / a;
    ds[k + 6] = ((5. / 6. + den[k]) * expansion - (2.5 + overdensity) * (1. + vel[k]) + tide) / a;
  }
}

// one Fehlberg (4,5) step: y advanced by the fifth-order weights, yerr = difference to the embedded fourth-order
// solution, dydt_out = derivative at the new point (rkf45_apply)
PF_HD void pf_rkf45_apply(double t, double h, double y[9], double yerr[9], const double dydt_in[9], double dydt_out[9],
                          const pf_sng_cosmo &c) {
  const double ah[] = {1.0 / 4.0, 3.0 / 8.0, 12.0 / 13.0, 1.0, 1.0 / 2.0};
  const double b3[] = {3.0 / 32.0, 9.0 / 32.0};
  const double b4[] = {1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0};
  const double b5[] = {8341.0 / 4104.0, -32832.0 / 4104.0, 29440.0 / 4104.0, -845.0 / 4104.0};
  const double b6[] = {-6080.0 / 20520.0, 41040.0 / 20520.0, -28352.0 / 20520.0, 9295.0 / 20520.0, -5643.0 / 20520.0};
  const double c1 = 902880.0 / 7618050.0, c3 = 3953664.0 / 7618050.0, c4 = 3855735.0 / 7618050.0, c5 = -1371249.0 / 7618050.0,
               c6 = 277020.0 / 7618050.0;
  const double ec[] = {0.0, 1.0 / 360.0, 0.0, -128.0 / 4275.0, -2197.0 / 75240.0, 1.0 / 50.0, 2.0 / 55.0};
  double k2[9], k3[9], k4[9], k5[9], k6[9], ytmp[9];
  const double *k1 = dydt_in;
  for (int i = 0; i < 9; i++) ytmp[i] = y[i] + ah[0] * h * k1[i];
  pf_sng_rhs(t + ah[0] * h, ytmp, k2, c);
  for (int i = 0; i < 9; i++) ytmp[i] = y[i] + h * (b3[0] * k1[i] + b3[1] * k2[i]);
  pf_sng_rhs(t + ah[1] * h, ytmp, k3, c);
  for (int i = 0; i < 9; i++) ytmp[i] = y[i] + h * (b4[0] * k1[i] + b4[1] * k2[i] + b4[2] * k3[i]);
  pf_sng_rhs(t + ah[2] * h, ytmp, k4, c);
  for (int i = 0; i < 9; i++) ytmp[i] = y[i] + h * (b5[0] * k1[i] + b5[1] * k2[i] + b5[2] * k3[i] + b5[3] * k4[i]);
  pf_sng_rhs(t + ah[3] * h, ytmp, k5, c);
  for (int i = 0; i < 9; i++) ytmp[i] = y[i] + h * (b6[0] * k1[i] + b6[1] * k2[i] + b6[2] * k3[i] + b6[3] * k4[i] + b6[4] * k5[i]);
  pf_sng_rhs(t + ah[4] * h, ytmp, k6, c);
  for (int i = 0; i < 9; i++) {
    const double d_i = c1 * k1[i] + c3 * k3[i] + c4 * k4[i] + c5 * k5[i] + c6 * k6[i];
    yerr[i] = h * (ec[1] * k1[i] + ec[3] * k3[i] + ec[4] * k4[i] + ec[5] * k5[i] + ec[6] * k6[i]);
    y[i] += h * d_i;
  }
  pf_sng_rhs(t + h, y, dydt_out, c);
}

// gsl_odeiv2_control_standard_new(1e-6, 1e-6, 1, 1), method order 5: -1 decrease, +1 increase, 0 keep (std_control_hadjust)
PF_HD int pf_std_hadjust(const double y[9], const double yerr[9], const double yp[9], double &h) {
  const double eps_abs = 1.0e-6, eps_rel = 1.0e-6, a_y = 1.0, a_dydt = 1.0, S = 0.9;
  const unsigned int ord = 5;
  const double h_old = h;
  double rmax = DBL_MIN;
  for (int i = 0; i < 9; i++) {
    const double D0 = eps_rel * (a_y * fabs(y[i]) + a_dydt * fabs(h_old * yp[i])) + eps_abs;
    const double r = fabs(yerr[i]) / fabs(D0);
    rmax = (r > rmax ? r : rmax);
  }
  if (rmax > 1.1) {
    double r = S / pow(rmax, 1.0 / ord);
    if (r < 0.2) r = 0.2;
    h = r * h_old;
    return -1;
  } else if (rmax < 0.5) {
    double r = S / pow(rmax, 1.0 / (ord + 1.0));
    if (r > 5.0) r = 5.0;
    if (r < 1.0) r = 1.0;
    h = r * h_old;
    return 1;
  }
  return 0;
}

// ell_sng (src/collapse_times.c:319-400): scale factor at which lambda_a of the first axis reaches 0.99999 (0: no
// collapse before a = 5; -1: the step size cannot be reduced any further).  The loop is gsl_odeiv2_evolve_apply:
// take a step, let the control object judge it, undo and retry with the smaller step when it asks for a decrease.
// olda / oldlam are set once before the loop and never updated, as in the reference.
PF_HD double pf_ell_sng(double l1, double l2, double l3, double D_in, const pf_sng_cosmo &c) {
  const double amin = 1.e-5, amax = 5.0;
  double hh = 1.e-6, mya = amin;
  double y[9] = {l1 * D_in, l2 * D_in, l3 * D_in,
                 l1 * D_in / (l1 * D_in - 1.), l2 * D_in / (l2 * D_in - 1.), l3 * D_in / (l3 * D_in - 1.),
                 l1 * D_in, l2 * D_in, l3 * D_in};
  const double olda = mya, oldlam = y[0];
  double dydt_in[9], dydt_out[9], y0[9], yerr[9];
  bool first = true;
  while (mya < amax) {
    // ---- gsl_odeiv2_evolve_apply(e, c, s, sys, &mya, amax, &hh, y)
    const double t0 = mya, dt = amax - t0;
    double h0 = hh;
    bool final_step = false;
    for (int i = 0; i < 9; i++) y0[i] = y[i];
    if (first) { pf_sng_rhs(t0, y, dydt_in, c); first = false; }
    else for (int i = 0; i < 9; i++) dydt_in[i] = dydt_out[i];
    for (;;) {
      if (dt >= 0.0 && h0 > dt) { h0 = dt; final_step = true; } else final_step = false;
      pf_rkf45_apply(t0, h0, y, yerr, dydt_in, dydt_out, c);
      mya = final_step ? amax : t0 + h0;
      const double h_old = h0;
      if (pf_std_hadjust(y, yerr, dydt_out, h0) < 0) {
        const double t_next = mya + h0;
        if (fabs(h0) < fabs(h_old) && t_next != mya) {
          for (int i = 0; i < 9; i++) y[i] = y0[i];
          continue;
        }
        return -1;
      }
      break;
    }
    if (!final_step) hh = h0;
    // ---- back in ell_sng
    if (y[0] >= 0.99999) return olda + (1. - oldlam) * (mya - olda) / (y[0] - oldlam);
  }
  return 0;
}

// ell(), ELL_SNG branch (src/collapse_times.c:416-426)
PF_HD double pf_ell_sng_F(double l1, double l2, double l3, double D_in, const pf_sng_cosmo &c) {
  const double bc = pf_ell_sng(l1, l2, l3, D_in, c);
  return bc > 0.0 ? 1. / bc : 0.0;
}
