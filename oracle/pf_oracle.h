/*
 * pf_oracle.h -- CPU ORACLE for the PINOCCHIO collapse-time hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (pigimonaco/Pinocchio V5.1: src/fmax.c, src/fmax-pfft.c,
 * src/collapse_times.c, src/LPT.c, the spline evaluation in src/cosmo.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the shipped library (libpinfmax_hip.so) never does.
 *
 * Parity pin: (1) END TO END against the five runs the reference commits with their outputs, each reproduced with the
 * restated IC generator (pf_genic.c + tests/ic_oracle.py) from seed and cosmology -- logged sigma of every radius to the 4
 * printed decimals, collapsed-cell count, 210-bin Fmax histogram:
 *   HMF_Validation/ (128^3, E&H, 9 radii): 1 230 387 vs 1 230 386 collapsed, histogram L1 70 of 2 097 152 (tests/test_hmf_validation_kat.py);
 *   example/log (128^3, V5.1 default flags, 7 radii): 687 252 vs 687 249 (count only: the log holds no histogram);
 *   tests/only_HMF_tests/RECOMPUTE_DISPLACEMENTS_LCDM = SCALE_DEP_LCDM (256^3, FixedIC, 9 radii; also with per-radius splines):
 *     10 989 577 vs 10 989 578, L1 118 of 16 777 216;
 *   tests/only_HMF_tests/READ_PK_TABLE_and_SCALE_DEP (256^3, tabulated CAMB spectrum, FixedIC, 10 radii): 12 822 323 vs 12 822 323, L1 444;
 *   tests/only_HMF_tests/MOD_GRAV_and_SCALE_DEP (256^3, TABULATED_CT + ELL_SNG + MOD_GRAV_FR): 10 935 586 vs 10 935 578, L1 198
 *   (tests/test_hmf256_kat.py; data in the tests/golden/ fixtures named ..._kat.json, made by the scripts beside them).
 * (2) The per-cell solver against the known answers of the reference's ell_classic / inverse_collapse_time (SURVEY.md
 * Appendix D -> tests/golden/collapse_kat.json).  (3) The displacement half, for which the reference commits no output, against
 * closed-form plane-wave answers derived in exact arithmetic from the reference's formulas (tests/golden/lpt_analytic.json) and
 * an independent numpy/pocketfft restatement (tests/np_restatement.py).  The reference itself is UNBUILDABLE in this image
 * (needs GSL, FFTW3-MPI and PFFT, all absent; no stand-ins are written), so no oracle/_ref exists; see DESIGN.md "Oracle".
 *
 * Flags mirrored: -DTWO_LPT -DTHREE_LPT, ELL_CLASSIC or ELL_SNG (+ MOD_GRAV_FR), TABULATED_CT (three interpolations),
 * SCALE_DEPENDENT pieces (per-radius splines, k-binned growth), float products (double ones: libpf_oracle_dp.so).
 */
#ifndef PF_ORACLE_H
#define PF_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NBINS 210 /* src/pinocchio.h:65 */

/* src/pinocchio.h:219-225: PRODFLOAT is float, or double in a -DDOUBLE_PRECISION_PRODUCTS build -- the same switch here
   (oracle/Makefile builds both: libpf_oracle.so and libpf_oracle_dp.so) */
#ifdef DOUBLE_PRECISION_PRODUCTS
typedef double ORC_PRODFLOAT;
#else
typedef float ORC_PRODFLOAT;
#endif
/* src/pinocchio.h:233-259 with -DTWO_LPT -DTHREE_LPT: 56 B with float products, 112 B with double ones */
typedef struct {
  int   Rmax;
  ORC_PRODFLOAT Fmax, Vel[3];
  ORC_PRODFLOAT Vel_2LPT[3];
  ORC_PRODFLOAT Vel_3LPT_1[3], Vel_3LPT_2[3];
} orc_product;

typedef struct orc_ctx orc_ctx;

/* single rank, cubic grid n^3 (n even >= 4; not a power of two: O(n^2) transforms, test sizes only); nthreads<=0 -> omp default */
orc_ctx *orc_create(int n, int nthreads);
void     orc_destroy(orc_ctx *c);

/* kdensity[0]: half-spectrum [x][y][n/2+1][2] (src/fmax-pfft.c:366) */
int orc_set_density(orc_ctx *c, const double *dk);
/* SPLINE[SP_INVGROW] knots (src/cosmo.c:401): x = log10 D, y = log10 a */
int orc_set_invgrow(orc_ctx *c, const double *x, const double *y, int nk);
/* scale-independent growth multipliers at the target redshift:
   g[0]=GrowingMode, g[1]=GrowingMode_2LPT, g[2]=GrowingMode_3LPT_1 (already
   carrying its minus sign, src/cosmo.c:1810), g[3]=GrowingMode_3LPT_2 */
int orc_set_growth(orc_ctx *c, const double g[4]);
/* SCALE_DEPENDENT build (rows f-3): SPLINE_INVGROW[ismooth] (src/initialization.c:1704-1708), and the k-binned
   growth of InterpolateGrowth (src/cosmo.c:1728-1755) for ScaleDep.order = 1..4: T[j] = log10 growth in k-bin j at the
   target redshift, k bins at 10^(logkmin + j dlogk), sign = -1 for GrowingMode_3LPT_1.  nk = 0: back to the scalar. */
int orc_set_invgrow_radius(orc_ctx *c, int ismooth, const double *x, const double *y, int nk);
int orc_set_growth_table(orc_ctx *c, int order, const double *T, int nk, double logkmin, double dlogk, double sign);

/* src/fmax.c:36-190: Ns radii (in CELL units: Rsmooth = R/CellSize, :233),
   then compute_displacements(1,0,z) when do_lpt != 0.  true_var[Ns] out. */
int orc_compute_fmax(orc_ctx *c, int ns, const double *rs_cells, int do_lpt,
                     double *true_var);

/* individual steps (same names as the reference, for unit tests) */
int orc_compute_second_derivatives(orc_ctx *c, double rs_cells);  /* fmax.c:225 */
int orc_compute_collapse_times(orc_ctx *c, int ismooth, double *true_var); /* collapse_times.c:431 */
int orc_compute_displacements(orc_ctx *c, int compute_sources);   /* fmax.c:292 */
int orc_fmax_pdf(orc_ctx *c, unsigned long long hist[ORC_NBINS]); /* fmax.c:509 */

/* accessors */
const orc_product *orc_products(orc_ctx *c);
const double *orc_second_derivative(orc_ctx *c, int i); /* i=0..5, order 11,22,33,12,13,23 */
const double *orc_kvector(orc_ctx *c, int which);       /* 0:2LPT 1:3LPT_1 2:3LPT_2 */

/* stand-alone FFTs on caller buffers (for unit tests of the oracle itself) */
int orc_c2r(orc_ctx *c, const double *spec, double *real_out); /* unnormalised */
int orc_r2c(orc_ctx *c, const double *real_in, double *spec_out);

/* per-cell functions (KAT entry points) */
double orc_ell_classic(double l1, double l2, double l3);             /* collapse_times.c:114 */
double orc_inverse_collapse_time(orc_ctx *c, const double *d,
                                 double *x1, double *x2, double *x3, int *fail); /* :679 */
double orc_inverse_growing_mode(orc_ctx *c, double D);               /* cosmo.c:1822 */
double orc_spline_eval(orc_ctx *c, double x);                        /* cosmo.c:2016 */

/* wall-clock breakdown of the last orc_compute_fmax, seconds:
   [0] total [1] deriv (k-loop+fft+copies) [2] fft [3] collapse [4] lpt */
void orc_timers(orc_ctx *c, double t[5]);

#ifdef __cplusplus
}
#endif
/* The build options below (SCALE_DEPENDENT additions above, TABULATED_CT, ELL_SNG, MOD_GRAV_FR) are pinned end to end by the
   runs the reference commits under tests/only_HMF_tests (SCALE_DEP_LCDM, MOD_GRAV_and_SCALE_DEP: sigma per radius, collapsed
   count, Fmax histogram; tests/test_hmf256_kat.py, fixtures tests/golden/{hmf256,mg256}_kat.json); TABULATED_CT tables of
   ELL_CLASSIC and ELL_SNG without f(R) have no run of their own and share that code.
   TABULATED_CT build (row f-4; src/collapse_times.c:780-1231 with the BILINEAR_SPLINE interpolation of :40):
   orc_set_tabulated_ct(ns, Smoothing.Variance[]) makes every following collapse-time pass build the table of ell()
   for its radius (100 x 50 x 50 nodes in (delta, x, y) / sqrt(variance)) and interpolate in it; ns = 0 returns to the
   direct solve.  orc_ct_build does the table alone; orc_ct_table / orc_ct_delta expose it ([iy][ix][id] / [id]). */
int orc_set_tabulated_ct(orc_ctx *c, int ns, const double *variance);
int orc_ct_build(orc_ctx *c, int ismooth, double variance);
const double *orc_ct_table(orc_ctx *c);
const double *orc_ct_delta(orc_ctx *c);
double orc_interpolate_collapse_time(orc_ctx *c, double l1, double l2, double l3);
/* the interpolation of a -DTRILINEAR (1) or -DALL_SPLINE (2) build (src/collapse_times.c:1153-1216; gsl_spline2d's bicubic
   restated from GSL 2.7.1 interp2d/bicubic.c); 0 = BILINEAR_SPLINE, the source's own define.  No run of the reference with
   either is committed: PARITY UNPINNED for these two (checked against an independent numpy construction). */
int orc_set_ct_interpolation(orc_ctx *c, int flavour);

/* ELL_SNG collapse model (oracle/pf_sng.c; src/collapse_times.c:222-400): scale factor of collapse of the ellipsoid
   (0: none, -1: integrator failure) and the F = 1/b_c of ell().  cosmo = {Omega0, OmegaLambda, OmegaRad, OmegaK,
   FR0, H_over_c, size}: the last three are the MOD_GRAV_FR force modification (src/collapse_times.c:295-312), FR0 = 0
   is standard gravity; orc_set_collapse_model takes the first four, orc_set_modified_gravity the rest (size per radius:
   Smoothing.Radius[ismooth], the previous radius for the last one, :378-388);
   D_in = GrowingMode(z(a = 1e-5), k of the radius).  orc_set_collapse_model(1, ...) makes ell() use it (the table of a
   TABULATED_CT build is then filled with it); model 0 = ELL_CLASSIC. */
double orc_ell_sng(double l1, double l2, double l3, double D_in, const double cosmo[7]);
double orc_ell_sng_F(double l1, double l2, double l3, double D_in, const double cosmo[7]);
int orc_set_modified_gravity(orc_ctx *c, double fr0, double h_over_c, int ns, const double *size);
int orc_set_collapse_model(orc_ctx *c, int model, const double cosmo[4], int ns, const double *D_in);

/* GSL's natural cubic spline (coefficients c) and my_spline_eval (src/cosmo.c:2016-2027) on explicit arrays */
int orc_cspline_coeffs(const double *xa, const double *ya, int size, double *sc);
double orc_my_spline_eval(const double *sx, const double *sy, const double *sc, int size, double x);

/* Fmax >= flast (src/distribute.c:695), indices by descending Fmax (src/fragment.c:484-503, 118-126; ties by index).
   indices / fmax hold n^3 entries; returns the number selected. */
size_t orc_select_sorted(orc_ctx *c, float flast, unsigned int *indices, float *fmax);

/* Sampled x-planes of a box too large for the whole oracle (the 1024^3 of the metric; tests/test_lpt_analytic.py).
   orc_create_planes: a context without field arrays (twiddles, splines, scalars); only the orc_plane_* entry points and the
   per-cell functions may be used with it.  orc_plane_derivatives: compute_derivative + reverse_transform
   (src/fmax-pfft.c:255-441, 203-228) evaluated on the x-planes xs[] only -- the x-transform as the plain sum over kx for
   those planes, the same filter expression per mode; components (ia, ib) as compute_derivative takes them ((-1,-1): plain
   transform); out [ncomp][nplanes][n][n].  orc_plane_collapse_times: compute_collapse_times (src/collapse_times.c:431-673)
   on a list of cells, d6 = [6][ncells], running maximum in fmax / rmax (initialised at ismooth 0). */
orc_ctx *orc_create_planes(int n, int nthreads);
int orc_plane_derivatives(orc_ctx *c, const double *spec, double rs_cells, int ncomp, const int *ia, const int *ib,
                          int nplanes, const int *xs, double *out);
int orc_plane_collapse_times(orc_ctx *c, int ismooth, size_t ncells, const double *d6, ORC_PRODFLOAT *fmax, int *rmax);
/* orc_plane_derivatives for a spectrum that does not fit the host (BASELINE config 5, 2048^3: 69 GB in fp64): the rows of the
   spectrum are added in pieces of consecutive kx (ascending, all of them), several radii at once; then the planes of each radius
   are finished.  The per-mode expressions and the order of the sum over kx are those of orc_plane_derivatives. */
typedef struct orc_plane_acc orc_plane_acc;
orc_plane_acc *orc_plane_acc_create(orc_ctx *c, int nrad, const double *rs_cells, int ncomp, const int *ia, const int *ib, int nplanes, const int *xs);
int orc_plane_acc_add(orc_plane_acc *a, const double *rows, int kx0, int nkx);
int orc_plane_acc_finish(orc_plane_acc *a, int irad, double *out);
double orc_plane_acc_power(orc_plane_acc *a, int irad);   /* sum of |spec|^2 smoothing^2 over the modes of the rows added (Parseval) */
void orc_plane_acc_destroy(orc_plane_acc *a);

#endif
