/*
 * pf_sng.c -- ORACLE (test infrastructure only): CPU restatement of the ELL_SNG collapse model of
 * src/collapse_times.c:222-400 -- the nine-equation system of Nadkarni-Ghosh & Singhal (2016) integrated with
 * GSL's adaptive Runge-Kutta-Fehlberg (4,5) stepper.
 *
 * GSL (2.7.1 in the reference's validation build, HMF_Validation/VALIDATION_log.txt:3) is not vendored in the
 * reference and not present in this image.  The three pieces the call site uses are restated from GSL's published
 * algorithm: ode-initval2/rkf45.c (step), ode-initval2/cstd.c (gsl_odeiv2_control_standard_new, hadjust) and
 * ode-initval2/evolve.c (gsl_odeiv2_evolve_apply).  Pinned end to end by the one ELL_SNG run the reference commits with its
 * outputs (tests/only_HMF_tests/MOD_GRAV_and_SCALE_DEP: 256^3, TABULATED_CT + MOD_GRAV_FR; collapsed count to 8 cells of
 * 16.7 M and the Fmax histogram, tests/test_hmf256_kat.py) and checked against scipy's integrators at the integrator's own
 * tolerance (tests/test_oracle.py).
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "pf_oracle.h"

/* src/cosmo.c:1675-1718 with params.simpleLambda: Ez = Hubble(z) / Hubble(0) */
static double sng_Esq(const double *cosmo, double z) {
  return cosmo[2] * pow(1. + z, 4.) + cosmo[0] * pow(1. + z, 3.) + cosmo[3] * pow(1. + z, 2.) + cosmo[1];
}
static double sng_Ez(const double *cosmo, double z) {
  double H0 = 100. * sqrt(sng_Esq(cosmo, 0.0));
  return 100. * sqrt(sng_Esq(cosmo, z)) / H0;
}
static double OmegaMatter(const double *cosmo, double z) {
  double Ezv = sng_Ez(cosmo, z);
  return cosmo[0] * pow(1. + z, 3.) / (Ezv * Ezv);
}
static double OmegaLambda(const double *cosmo, double z) {
  double Ezv = sng_Ez(cosmo, z);
  return cosmo[1] / (Ezv * Ezv);
}

/* ForceModification, src/collapse_times.c:295-312 (MOD_GRAV_FR): Hu-Sawicki f(R) with |f_R0| = cosmo[4] (the FR0 of the
   build), H_over_c = cosmo[5] (100 / c, src/cosmo.c:109), size = cosmo[6] (the smoothing radius handed to the ODE as
   its parameter, :378-388) */
static double ForceModification(const double *cosmo, double size, double a, double delta) {
  const double FR0 = cosmo[4], H_over_c = cosmo[5];
  double ff = 4. * cosmo[1] / cosmo[0];
  double thickness = FR0 / cosmo[0] / pow(H_over_c * size, 2.0) *
                     pow(a, 7.) * pow((1. + delta), -1. / 3.) *
                     (pow((1.0 + ff) / (1.0 + ff * pow(a, 3.)), 2.0) -
                      pow((1.0 + ff) / (1.0 + delta + ff * pow(a, 3.)), 2.0));
  double F3 = (thickness * (3. + thickness * (-3. + thickness)));
  if (F3 < 0.) F3 = 0.;
  return (F3 < 1. ? F3 / 3. : 1. / 3);
}

/* sng_system, src/collapse_times.c:241-293; cosmo[4] != 0 selects the MOD_GRAV_FR branch (:271-273) */
static int sng_system(double t, const double y[], double f[], const double *cosmo) {
  int i, j;
  double sum;
  double omegam = OmegaMatter(cosmo, 1. / t - 1.);
  double omegal = OmegaLambda(cosmo, 1. / t - 1.);
  double delta = y[6] + y[7] + y[8];
  for (i = 0; i < 3; i++) {
    sum = 0.;
    for (j = 0; j < 3; j++) {
      if (i == j || y[i] == y[j]) {
        continue;
      } else {
        sum += (y[j + 6] - y[i + 6]) * ((1. - y[i]) * (1. - y[i]) * (1. + y[i + 3]) -
               (1. - y[j]) * (1. - y[j]) * (1. + y[j + 3])) /
               ((1. - y[i]) * (1. - y[i]) - (1. - y[j]) * (1. - y[j]));
      }
    }
    f[i] = (y[i + 3] * (y[i] - 1.0)) / t;
    if (cosmo[4] != 0.0)
      f[i + 3] = (0.5 * (y[i + 3] * (omegam - 2.0 * omegal - 2.0)
                         - 3.0 * omegam * y[i + 6] * (1. + ForceModification(cosmo, cosmo[6], t, delta))
                         - 2.0 * y[i + 3] * y[i + 3])) / t;
    else
      f[i + 3] = (0.5 * (y[i + 3] * (omegam - 2.0 * omegal - 2.0)
                         - 3.0 * omegam * y[i + 6]
                         - 2.0 * y[i + 3] * y[i + 3])) / t;
    f[i + 6] = ((5. / 6. + y[i + 6]) *
                ((3. + y[3] + y[4] + y[5]) - (1. + delta) / (2.5 + delta) * (y[3] + y[4] + y[5])) -
                (2.5 + delta) * (1. + y[i + 3]) + sum) / t;
  }
  return 0;
}

/* GSL ode-initval2/rkf45.c: Fehlberg coefficients and rkf45_apply (dim 9) */
#define DIM 9
static const double ah[] = {1.0 / 4.0, 3.0 / 8.0, 12.0 / 13.0, 1.0, 1.0 / 2.0};
static const double b3[] = {3.0 / 32.0, 9.0 / 32.0};
static const double b4[] = {1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0};
static const double b5[] = {8341.0 / 4104.0, -32832.0 / 4104.0, 29440.0 / 4104.0, -845.0 / 4104.0};
static const double b6[] = {-6080.0 / 20520.0, 41040.0 / 20520.0, -28352.0 / 20520.0, 9295.0 / 20520.0, -5643.0 / 20520.0};
static const double c1 = 902880.0 / 7618050.0;
static const double c3 = 3953664.0 / 7618050.0;
static const double c4 = 3855735.0 / 7618050.0;
static const double c5 = -1371249.0 / 7618050.0;
static const double c6 = 277020.0 / 7618050.0;
static const double ec[] = {0.0, 1.0 / 360.0, 0.0, -128.0 / 4275.0, -2197.0 / 75240.0, 1.0 / 50.0, 2.0 / 55.0};

static void rkf45_apply(double t, double h, double y[], double yerr[], const double dydt_in[], double dydt_out[],
                        const double *cosmo) {
  double k1[DIM], k2[DIM], k3[DIM], k4[DIM], k5[DIM], k6[DIM], ytmp[DIM];
  int i;
  memcpy(k1, dydt_in, sizeof(k1));                       /* k1 step (derivative supplied by the evolver) */
  for (i = 0; i < DIM; i++) ytmp[i] = y[i] + ah[0] * h * k1[i];
  sng_system(t + ah[0] * h, ytmp, k2, cosmo);            /* k2 */
  for (i = 0; i < DIM; i++) ytmp[i] = y[i] + h * (b3[0] * k1[i] + b3[1] * k2[i]);
  sng_system(t + ah[1] * h, ytmp, k3, cosmo);            /* k3 */
  for (i = 0; i < DIM; i++) ytmp[i] = y[i] + h * (b4[0] * k1[i] + b4[1] * k2[i] + b4[2] * k3[i]);
  sng_system(t + ah[2] * h, ytmp, k4, cosmo);            /* k4 */
  for (i = 0; i < DIM; i++) ytmp[i] = y[i] + h * (b5[0] * k1[i] + b5[1] * k2[i] + b5[2] * k3[i] + b5[3] * k4[i]);
  sng_system(t + ah[3] * h, ytmp, k5, cosmo);            /* k5 */
  for (i = 0; i < DIM; i++) ytmp[i] = y[i] + h * (b6[0] * k1[i] + b6[1] * k2[i] + b6[2] * k3[i] + b6[3] * k4[i] + b6[4] * k5[i]);
  sng_system(t + ah[4] * h, ytmp, k6, cosmo);            /* k6 and final sum */
  for (i = 0; i < DIM; i++) {
    const double d_i = c1 * k1[i] + c3 * k3[i] + c4 * k4[i] + c5 * k5[i] + c6 * k6[i];
    y[i] += h * d_i;
  }
  sng_system(t + h, y, dydt_out, cosmo);                 /* derivatives at output */
  for (i = 0; i < DIM; i++) yerr[i] = h * (ec[1] * k1[i] + ec[3] * k3[i] + ec[4] * k4[i] + ec[5] * k5[i] + ec[6] * k6[i]);
}

/* GSL ode-initval2/cstd.c std_control_hadjust with (eps_abs, eps_rel, a_y, a_dydt) = (1e-6, 1e-6, 1, 1), ord = 5.
   returns -1 decrease, +1 increase, 0 unchanged */
static int std_control_hadjust(const double y[], const double yerr[], const double yp[], double *h) {
  const double eps_abs = 1.0e-6, eps_rel = 1.0e-6, a_y = 1.0, a_dydt = 1.0;
  const unsigned int ord = 5;
  const double S = 0.9;
  const double h_old = *h;
  double rmax = DBL_MIN;
  int i;
  for (i = 0; i < DIM; i++) {
    const double D0 = eps_rel * (a_y * fabs(y[i]) + a_dydt * fabs(h_old * yp[i])) + eps_abs;
    const double r = fabs(yerr[i]) / fabs(D0);
    rmax = (r > rmax ? r : rmax); /* GSL_MAX_DBL */
  }
  if (rmax > 1.1) {
    double r = S / pow(rmax, 1.0 / ord);
    if (r < 0.2) r = 0.2;
    *h = r * h_old;
    return -1;
  } else if (rmax < 0.5) {
    double r = S / pow(rmax, 1.0 / (ord + 1.0));
    if (r > 5.0) r = 5.0;
    if (r < 1.0) r = 1.0;
    *h = r * h_old;
    return 1;
  }
  return 0;
}

/* state of gsl_odeiv2_evolve across calls */
typedef struct { double dydt_in[DIM], dydt_out[DIM]; unsigned long count; } evolve_state;

/* GSL ode-initval2/evolve.c gsl_odeiv2_evolve_apply (forward integration, no driver) */
static int evolve_apply(evolve_state *e, double *t, double t1, double *h, double y[], const double *cosmo) {
  const double t0 = *t;
  double h0 = *h;
  int final_step = 0;
  const double dt = t1 - t0;
  double y0[DIM], yerr[DIM];
  memcpy(y0, y, sizeof(y0));
  if (e->count == 0) sng_system(t0, y, e->dydt_in, cosmo);
  else memcpy(e->dydt_in, e->dydt_out, sizeof(e->dydt_in));
  for (;;) { /* try_step */
    if (dt >= 0.0 && h0 > dt) { h0 = dt; final_step = 1; } else final_step = 0;
    rkf45_apply(t0, h0, y, yerr, e->dydt_in, e->dydt_out, cosmo);
    e->count++;
    if (final_step) *t = t1; else *t = t0 + h0;
    {
      const double h_old = h0;
      const int hadj = std_control_hadjust(y, yerr, e->dydt_out, &h0);
      if (hadj < 0) {
        const double t_curr = *t, t_next = t_curr + h0;
        if (fabs(h0) < fabs(h_old) && t_next != t_curr) { /* undo the step and try again with the smaller h0 */
          memcpy(y, y0, sizeof(y0));
          continue;
        } else {
          *h = h0;
          return 1; /* GSL_FAILURE: step size cannot be decreased */
        }
      }
    }
    break;
  }
  if (final_step == 0) *h = h0;
  return 0;
}

/* ell_sng, src/collapse_times.c:319-400.  D_in = GrowingMode(1/amin - 1, k of this radius), supplied by the caller.
   olda / oldlam are set once before the loop and never updated (kept as in the reference). */
double orc_ell_sng(double l1, double l2, double l3, double D_in, const double cosmo[7]) {
  double hh = 1.e-6;
  double amin = 1.e-5, amax = 5.0;
  double mya = amin;
  evolve_state e;
  memset(&e, 0, sizeof(e));
  double y[9] = {l1 * D_in, l2 * D_in, l3 * D_in,
                 l1 * D_in / (l1 * D_in - 1.), l2 * D_in / (l2 * D_in - 1.), l3 * D_in / (l3 * D_in - 1.),
                 l1 * D_in, l2 * D_in, l3 * D_in};
  double olda = mya;
  double oldlam = y[0];
  while (mya < amax) {
    int status = evolve_apply(&e, &mya, amax, &hh, y, cosmo);
    if (status != 0) return -1;
    if (y[0] >= 0.99999) return olda + (1. - oldlam) * (mya - olda) / (y[0] - oldlam);
  }
  return 0;
}

/* ell(), ELL_SNG branch (src/collapse_times.c:416-426) */
double orc_ell_sng_F(double l1, double l2, double l3, double D_in, const double cosmo[7]) {
  double bc = orc_ell_sng(l1, l2, l3, D_in, cosmo);
  if (bc > 0.0) return 1. / bc;
  return 0.0;
}
