/*
 * hmf_validation.c -- the reference's committed validation run (HMF_Validation/: 128^3, seed 486604, Eisenstein & Hu
 * spectrum, sigma8 = 0.8, nine smoothing radii) from a plain C host through the C ABI of libpinfmax_hip.so:
 * initial conditions on the device from seed + cosmology, the collapse-time sweep, the Fmax histogram.
 * Prints the reference's own log lines ("Completed, R=..., computed sigma: ...", "Number of collapsed particles").
 *
 *     make -C examples && ./examples/hmf_validation
 *
 * Expected (HMF_Validation/log_RUN.txt:135-335, 407): computed sigma 0.2032 0.3258 0.5051 0.7505 1.0850 1.5527 2.1897
 * 2.6563 2.7733; 1230386 collapsed particles (this build: within one particle).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/pinfmax.h"

#define NKNOTS 210 /* NBINS, src/pinocchio.h:65 */

/* growing mode of flat LCDM without radiation, D(a) = 2.5 Om H(a) int_0^a da' / (a' H(a'))^3, normalised to D(1) = 1
   (the reference integrates the equivalent ODE, src/cosmo.c:229-401); Gauss-Legendre in t with a' = a t^2 */
static double growth(double a, double om) {
  static const double xg[8] = {0.0950125098376374, 0.2816035507792589, 0.4580167776572274, 0.6178762444026438,
                               0.7554044083550030, 0.8656312023878318, 0.9445750230732326, 0.9894009349916499};
  static const double wg[8] = {0.1894506104550685, 0.1826034150449236, 0.1691565193950025, 0.1495959888165767,
                               0.1246289712555339, 0.0951585116824928, 0.0622535239386479, 0.0271524594117541};
  const double ol = 1.0 - om;
  double s = 0.0;
  const int panels = 64;
  for (int p = 0; p < panels; p++) {
    const double t0 = (double)p / panels, t1 = (double)(p + 1) / panels, c = 0.5 * (t0 + t1), h = 0.5 * (t1 - t0);
    for (int i = 0; i < 16; i++) {
      const double t = c + (i < 8 ? -xg[i] : xg[i - 8]) * h, w = wg[i & 7] * h;
      const double ap = a * t * t, H = sqrt(om / (ap * ap * ap) + ol);
      s += w * 2.0 * a * t / pow(ap * H, 3.0);
    }
  }
  return 2.5 * om * sqrt(om / (a * a * a) + ol) * s;
}

int main(void) {
  const int n = 128, ns = 9;
  const double h100 = 0.7, box = 128.0 / h100, cell = box / n; /* BoxSize 128 Mpc/h in true Mpc */
  const double radius[9] = {20.635922, 13.996056, 9.026099, 5.465945, 3.058354, 1.548258, 0.689079, 0.258729, 0.0};
  pf_config cfg = {n, 0, 1, 0, 8, 0};
  pf_ctx *ctx = NULL;
  pf_genic_params ic = {0.25, 0.044, h100, 0.96, box, 2.03146e7 /* PkNorm as logged */, 486604u, 0, 0, 0, NULL, NULL /* Eisenstein & Hu, no table */,
                        0 /* spectrum: by pk_n */, 0.0 /* no warm-dark-matter cut-off */, 0.0 /* UnitLength_in_cm: the default */};
  double x[NKNOTS], y[NKNOTS], rs[9], tv[9], d1;
  unsigned long long pdf[PF_NBINS], coll = 0;

  if (pf_create(&ctx, &cfg)) return 1; /* prints "ERROR on task 0: ..." itself */
  if (pf_genic_density(ctx, &ic)) return 1;
  d1 = growth(1.0, ic.Omega0);
  for (int i = 0; i < NKNOTS; i++) { /* SPLINE[SP_INVGROW]: x = log10 D(a), y = log10 a on log10 a = -4 + 0.02 i */
    y[i] = -4.0 + 0.02 * i;
    x[i] = log10(growth(pow(10.0, y[i]), ic.Omega0) / d1);
  }
  if (pf_set_invgrow(ctx, -1, x, y, NKNOTS)) return 1;
  for (int i = 0; i < ns; i++) rs[i] = radius[i] / cell; /* Rsmooth = R / CellSize, src/fmax.c:233 */
  if (pf_sweep(ctx, ns, rs, tv)) return 1;
  for (int i = 0; i < ns; i++) printf("Completed, R=%6.3f, computed sigma: %7.4f\n", radius[i], sqrt(tv[i]));
  if (pf_fmax_pdf(ctx, pdf)) return 1;
  for (int i = 10; i < PF_NBINS; i++) coll += pdf[i];
  printf("Number of collapsed particles to z=0: %llu\n", coll);
  pf_destroy(ctx);
  return 0;
}
