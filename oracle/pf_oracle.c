/*
 * pf_oracle.c -- CPU ORACLE (test infrastructure, see pf_oracle.h).
 *
 * Restates, function by function, the reference hot path.  Every routine
 * cites the reference file:line it follows (paths relative to the reference
 * tree).  Structure is deliberately the reference's own (one k-loop + one
 * c2r per derivative, flat copies between user arrays and the FFT buffers,
 * AoS float products), NOT the fused design of the HIP library, so that the
 * two implementations are independent.
 *
 * The FFT is a textbook iterative radix-2 (power-of-two sizes only); the
 * reference uses PFFT/FFTW, any correct FFT gives the same fields to fp64
 * round-off.  3-D c2r semantics = complex inverse
 * transforms along x and y on the half-spectrum, then a 1-D c2r along z that
 * ignores the imaginary parts of the z-DC and z-Nyquist inputs (what FFTW /
 * pocketfft compute).
 */
#define _POSIX_C_SOURCE 200809L
#include "pf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846 /* src/pinocchio.h:56 */
#define ORC_MAX_SMOOTH 64
#define ORC_MAX_KBINS 32
#define ORC_SMALL ((double)1.e-20)     /* src/collapse_times.c:38 */

struct orc_ctx {
  int n, nzh, nthreads;
  size_t n_r;   /* total_local_size      = n^3            (fmax-pfft.c:111) */
  size_t n_fft; /* total_local_size_fft  = 2*n*n*(n/2+1)  (fmax-pfft.c:110) */
  double norm;  /* 1/Ntotal (fmax-pfft.c:85) */
  double *tw;   /* twiddles exp(+2 pi i j/n), interleaved */
  int    *brev; /* bit-reversal permutation */

  /* the reference's globals (src/variables.c) */
  double *kdensity;                 /* kdensity[0] */
  double *cvector, *rvector;        /* cvector_fft[0], rvector_fft[0] */
  double *second_derivatives[6];    /* second_derivatives[0][0..5] */
  double *kvector_2LPT, *kvector_3LPT_1, *kvector_3LPT_2;
  double *source_2LPT, *source_3LPT_1, *source_3LPT_2; /* alias kvector_* (allocations.c:349,357) */
  orc_product *products;
  double Rsmooth;                   /* in cells */
  int    sd_order;                  /* ScaleDep.order */
  double growth[4];

  /* inverse-growth natural cubic spline (GSL cspline restated) */
  int nk;
  double *sx, *sy, *sc;
  /* SCALE_DEPENDENT build: one inverse-growth spline per smoothing radius, SPLINE_INVGROW[ismooth]
     (src/initialization.c:1551-1553, 1704-1708; src/cosmo.c:1828), and the k-binned growth tables behind
     InterpolateGrowth (src/cosmo.c:1728-1755): T[j] = spline_j(-log10(1+z)), j = 0..NkBINS-1 */
  int rnk[ORC_MAX_SMOOTH];
  double *rsx[ORC_MAX_SMOOTH], *rsy[ORC_MAX_SMOOTH], *rsc[ORC_MAX_SMOOTH];
  int gt_n[4];
  double gt_T[4][ORC_MAX_KBINS], gt_logkmin[4], gt_dlogk[4], gt_sign[4];
  /* TABULATED_CT build (src/collapse_times.c:780-1231, BILINEAR_SPLINE flavour :40): per radius a table of ell() on
     a (delta, x, y) grid and one natural cubic spline in delta per (x, y) node */
  int model;                      /* 0 ELL_CLASSIC, 1 ELL_SNG (oracle/pf_sng.c) */
  double sng_cosmo[7], sng_Din[ORC_MAX_SMOOTH], sng_size[ORC_MAX_SMOOTH];
  int cur_ismooth;
  int tab_ns;
  int ct_flavour;                 /* 0 BILINEAR_SPLINE (the source's define), 1 TRILINEAR, 2 ALL_SPLINE */
  double tab_var[ORC_MAX_SMOOTH]; /* Smoothing.Variance[] */
  double tab_ampl;                /* sqrt(Smoothing.Variance[ismooth]) of the table in place */
  double *ct_table, *ct_c, *ct_delta;

  double t_total, t_deriv, t_fft, t_coll, t_lpt;
};

/* InterpolateGrowth (src/cosmo.c:1728-1755, SCALE_DEPENDENT branch) followed by the +-pow(10., .) of
   GrowingMode* (src/cosmo.c:1789-1819).  k is whatever compute_derivative passes: |k| in rad/cell (quirk Q3). */
static double growth_of_k(const orc_ctx *c, int o, double k) {
  const double *T = c->gt_T[o];
  const int nk = c->gt_n[o];
  const double LOGKMIN = c->gt_logkmin[o], DELTALOGK = c->gt_dlogk[o];
  const double kmin = pow(10., LOGKMIN), kmax = pow(10., LOGKMIN + (nk - 1) * DELTALOGK);
  double v;
  if (k < kmin) v = T[0];
  else if (k > kmax) v = T[nk - 1];
  else {
    double dk = (log10(k) - LOGKMIN) / DELTALOGK;
    int kk = (int)dk;
    dk -= kk;
    v = (kk >= nk - 1) ? T[nk - 1] : dk * T[kk + 1] + (1 - dk) * T[kk];
  }
  return c->gt_sign[o] * pow(10., v);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ FFT -- */

/* in-place radix-2 DIT on n interleaved complex values; sign=+1 -> e^{+i..}.  Grid sizes that are not a power of two
   (the reference takes any GridSize through FFTW; e.g. the 200^3 of INSTALLATION:101) go through the plain O(n^2) sum
   with the same twiddle table -- test sizes only. */
static void fft1d(double *a, int n, int sign, const double *tw, const int *brev) {
  if (n & (n - 1)) {
    double tmp[2 * n];
    for (int k = 0; k < n; k++) {
      double sr = 0.0, si = 0.0;
      int idx = 0; /* (j * k) mod n */
      for (int j = 0; j < n; j++) {
        const double wr = tw[2 * idx], wi = sign * tw[2 * idx + 1];
        sr += a[2 * j] * wr - a[2 * j + 1] * wi;
        si += a[2 * j] * wi + a[2 * j + 1] * wr;
        idx += k; if (idx >= n) idx -= n;
      }
      tmp[2 * k] = sr; tmp[2 * k + 1] = si;
    }
    memcpy(a, tmp, sizeof(double) * 2 * n);
    return;
  }
  for (int i = 0; i < n; i++) {
    int j = brev[i];
    if (j > i) {
      double tr = a[2 * i], ti = a[2 * i + 1];
      a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
      a[2 * j] = tr; a[2 * j + 1] = ti;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int s = 0; s < n; s += len) {
      for (int k = 0; k < half; k++) {
        double wr = tw[2 * (k * step)], wi = sign * tw[2 * (k * step) + 1];
        double *u = a + 2 * (s + k), *v = a + 2 * (s + k + half);
        double xr = v[0] * wr - v[1] * wi, xi = v[0] * wi + v[1] * wr;
        v[0] = u[0] - xr; v[1] = u[1] - xi;
        u[0] += xr; u[1] += xi;
      }
    }
  }
}

/* (The loops over planes of the transforms and of the k-space filter are scheduled dynamically: every iteration writes its own
   plane, so the results do not depend on who runs it, and on a host shared with other jobs a thread that loses its core no
   longer holds up a static team -- the oracle got SLOWER from 16 to 128 threads on such a host with static schedules.) */
/* complex transforms along x and y of a half-spectrum array [n][n][nzh].  The lines are strided (nzh, n*nzh complex);
   ORC_FFT_BLOCK neighbouring kz columns are gathered together so that every cache line fetched is used whole.  Each
   line still goes through the same fft1d: results do not depend on the blocking. */
#define ORC_FFT_BLOCK 8
static void fft_strided_lines(const orc_ctx *c, double *base, size_t stride, int ncols, int sign, double *lines) {
  const int n = c->n;
  for (int e = 0; e < n; e++) {
    const double *src = base + 2 * (size_t)e * stride;
    for (int j = 0; j < ncols; j++) { lines[2 * ((size_t)j * n + e)] = src[2 * j]; lines[2 * ((size_t)j * n + e) + 1] = src[2 * j + 1]; }
  }
  for (int j = 0; j < ncols; j++) fft1d(lines + 2 * (size_t)j * n, n, sign, c->tw, c->brev);
  for (int e = 0; e < n; e++) {
    double *dst = base + 2 * (size_t)e * stride;
    for (int j = 0; j < ncols; j++) { dst[2 * j] = lines[2 * ((size_t)j * n + e)]; dst[2 * j + 1] = lines[2 * ((size_t)j * n + e) + 1]; }
  }
}
static void fft_xy(orc_ctx *c, double *spec, int sign) {
  const int n = c->n, nzh = c->nzh;
#pragma omp parallel num_threads(c->nthreads)
  {
    double *lines = (double *)malloc(sizeof(double) * 2 * n * ORC_FFT_BLOCK);
    /* y lines: fixed (x,kz), stride nzh */
#pragma omp for schedule(dynamic, 1)
    for (int x = 0; x < n; x++)
      for (int kz = 0; kz < nzh; kz += ORC_FFT_BLOCK)
        fft_strided_lines(c, spec + 2 * ((size_t)x * n * nzh + kz), (size_t)nzh, nzh - kz < ORC_FFT_BLOCK ? nzh - kz : ORC_FFT_BLOCK, sign, lines);
    /* x lines: fixed (y,kz), stride n*nzh */
#pragma omp for schedule(dynamic, 1)
    for (int y = 0; y < n; y++)
      for (int kz = 0; kz < nzh; kz += ORC_FFT_BLOCK)
        fft_strided_lines(c, spec + 2 * ((size_t)y * nzh + kz), (size_t)n * nzh, nzh - kz < ORC_FFT_BLOCK ? nzh - kz : ORC_FFT_BLOCK, sign, lines);
    free(lines);
  }
}

/* pfft_execute(reverse_plan): unnormalised c2r, cvector -> rvector.
   (destroys the spectrum, as FFTW's c2r may) */
static void c2r_3d(orc_ctx *c, double *spec, double *real_out) {
  const int n = c->n, nzh = c->nzh;
  fft_xy(c, spec, +1);
#pragma omp parallel num_threads(c->nthreads)
  {
    double *line = (double *)malloc(sizeof(double) * 2 * n);
#pragma omp for schedule(dynamic, 1)
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++) {
        const double *h = spec + 2 * ((size_t)x * n + y) * nzh;
        /* Hermitian extension; Im of DC and Nyquist do not reach the real part */
        for (int k = 0; k < nzh; k++) { line[2 * k] = h[2 * k]; line[2 * k + 1] = h[2 * k + 1]; }
        for (int k = nzh; k < n; k++) { line[2 * k] = h[2 * (n - k)]; line[2 * k + 1] = -h[2 * (n - k) + 1]; }
        line[1] = 0.0; line[2 * (n / 2) + 1] = 0.0;
        fft1d(line, n, +1, c->tw, c->brev);
        double *r = real_out + ((size_t)x * n + y) * n;
        for (int z = 0; z < n; z++) r[z] = line[2 * z];
      }
    free(line);
  }
}

/* pfft_execute(forward_plan): unnormalised r2c, rvector -> cvector */
static void r2c_3d(orc_ctx *c, const double *real_in, double *spec) {
  const int n = c->n, nzh = c->nzh;
#pragma omp parallel num_threads(c->nthreads)
  {
    double *line = (double *)malloc(sizeof(double) * 2 * n);
#pragma omp for schedule(dynamic, 1)
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++) {
        const double *r = real_in + ((size_t)x * n + y) * n;
        for (int z = 0; z < n; z++) { line[2 * z] = r[z]; line[2 * z + 1] = 0.0; }
        fft1d(line, n, -1, c->tw, c->brev);
        double *h = spec + 2 * ((size_t)x * n + y) * nzh;
        for (int k = 0; k < nzh; k++) { h[2 * k] = line[2 * k]; h[2 * k + 1] = line[2 * k + 1]; }
      }
    free(line);
  }
  fft_xy(c, spec, -1);
}

/* -------------------------------------------- src/fmax-pfft.c restated -- */

/* fmax-pfft.c:191-200 */
static double forward_transform(orc_ctx *c) {
  double t = now_s();
  r2c_3d(c, c->rvector, c->cvector);
  return now_s() - t;
}

/* fmax-pfft.c:203-228.  The double application of norm on the last
   total_local_size_fft%4 elements (quirk Q1) is inert: n even -> n_fft%4==0 */
static double reverse_transform(orc_ctx *c) {
  double t = now_s();
  c2r_3d(c, c->cvector, c->rvector);
  for (size_t i = c->n_fft - c->n_fft % 4; i < c->n_fft; i++) c->rvector[i] *= c->norm;
  const size_t nr = c->n_r;
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
  for (size_t i = 0; i < nr; i++) c->rvector[i] *= c->norm;
  return now_s() - t;
}

/* fmax-pfft.c:444-456 */
static double greens_function(const double *diff_comp, double k_squared, int first_derivative,
                              int second_derivative) {
  if (first_derivative == -1 && second_derivative == -1) return 1.0;
  if (first_derivative == 0 && second_derivative == 0)
    return -diff_comp[first_derivative] * diff_comp[second_derivative] / k_squared;
  else
    return diff_comp[first_derivative] * diff_comp[second_derivative] / k_squared;
}

/* fmax-pfft.c:255-441, non-transposed layout, single rank (start=0, local=global).
   The reference's k-loop is serial; the oracle threads it over idx so that the
   CPU baseline is not handicapped (results are identical, each mode is independent). */
static int compute_derivative(orc_ctx *c, int first_derivative, int second_derivative) {
  const int n = c->n, nzh = c->nzh;
  const int Nhalf = n / 2;
  const double knorm = 2. * ORC_PI / (double)n;
  const int swap = ((first_derivative == 0 && second_derivative > 0) ||
                    (first_derivative > 0 && second_derivative == 0));
  const double Rsmooth = c->Rsmooth;
  double growth_rate_order;
  switch (c->sd_order) { /* fmax-pfft.c:344-364, scale-independent growth */
    case 1: growth_rate_order = c->growth[0]; break;
    case 2: growth_rate_order = c->growth[1]; break;
    case 3: growth_rate_order = c->growth[2]; break;
    case 4: growth_rate_order = c->growth[3]; break;
    default: growth_rate_order = 1.0; break;
  }
  double *cv = c->cvector;

#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic, 1)
  for (int idx = 0; idx < n; idx++) {
    int ii[3];
    ii[0] = idx;
    if (ii[0] > Nhalf) ii[0] -= n;
    double k_x = knorm * ii[0];
    double k2_0 = k_x * k_x;
    for (int idy = 0; idy < n; idy++) {
      ii[1] = idy;
      if (ii[1] > Nhalf) ii[1] -= n;
      double k_y = knorm * ii[1];
      double k2_1 = k2_0 + k_y * k_y;
      for (int idz = 0; idz < nzh; idz++) {
        ii[2] = idz;
        if (ii[2] > Nhalf) ii[2] -= n;
        double k_z = knorm * ii[2];
        double k_squared = k2_1 + k_z * k_z;
        double growth_rate = growth_rate_order;
        if (c->sd_order >= 1 && c->sd_order <= 4 && c->gt_n[c->sd_order - 1] > 0)
          growth_rate = growth_of_k(c, c->sd_order - 1, sqrt(k_squared)); /* fmax-pfft.c:339-364 */
        size_t index = 2 * (((size_t)idx * n + idy) * nzh + idz);
        if (k_squared != 0.) {
          double smoothing = exp(-0.5 * k_squared * Rsmooth * Rsmooth);
          double diff_comp[4];
          diff_comp[0] = 1.0; diff_comp[1] = k_x; diff_comp[2] = k_y; diff_comp[3] = k_z;
          double green = greens_function(diff_comp, k_squared, first_derivative, second_derivative);
          cv[index] *= green * smoothing * growth_rate;
          cv[index + 1] *= green * smoothing * growth_rate;
        }
        if (swap) {
          double tmp = cv[index + 1];
          cv[index + 1] = cv[index];
          cv[index] = -tmp;
        }
      }
    }
  }
  double time = reverse_transform(c);
  c->t_fft += time;
  return 0;
}

/* a flat copy by all threads (same bytes as memcpy; the reference's copies are serial loops -- threaded here, like the k-loop,
   so that the CPU baseline is not handicapped); chunked by x-planes, as every other loop of the oracle */
static void par_copy(const orc_ctx *c, double *dst, const double *src, size_t count) {
  const size_t chunk = (count + (size_t)c->n - 1) / (size_t)c->n;
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
  for (int i = 0; i < c->n; i++) {
    const size_t lo = (size_t)i * chunk, hi = lo + chunk < count ? lo + chunk : count;
    if (lo < hi) memcpy(dst + lo, src + lo, sizeof(double) * (hi - lo));
  }
}
/* fmax-pfft.c:459-560: flat copies */
static void write_in_cvector(orc_ctx *c, const double *v) { par_copy(c, c->cvector, v, c->n_fft); }
static void write_from_cvector(orc_ctx *c, double *v) { par_copy(c, v, c->cvector, c->n_fft); }
static void write_in_rvector(orc_ctx *c, const double *v) { par_copy(c, c->rvector, v, c->n_r); }
static void write_from_rvector(orc_ctx *c, double *v) { par_copy(c, v, c->rvector, c->n_r); }

/* fmax-pfft.c:563-631 */
static void write_from_rvector_to_products(orc_ctx *c, int ia, int order) {
  const size_t nr = c->n_r;
  orc_product *p = c->products;
  const double *rv = c->rvector;
  switch (order) {
    case 1:
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
      for (size_t i = 0; i < nr; i++) p[i].Vel[ia] = rv[i];
      break;
    case 2:
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
      for (size_t i = 0; i < nr; i++) p[i].Vel_2LPT[ia] = rv[i];
      break;
    case 3:
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
      for (size_t i = 0; i < nr; i++) p[i].Vel_3LPT_1[ia] = rv[i];
      break;
    case 4:
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
      for (size_t i = 0; i < nr; i++) p[i].Vel_3LPT_2[ia] = rv[i];
      break;
    default: break;
  }
}

/* ------------------------------------------------- src/cosmo.c restated -- */

/* GSL 2.7.1 (not vendored in the reference; pinned by HMF_Validation/
   VALIDATION_log.txt:3) interpolation/cspline.c cspline_init: natural cubic
   spline, c[0]=c[n-1]=0, interior c from the symmetric tridiagonal system
   solved by linalg/tridiag.c solve_tridiag (LDL^t). */
static int natural_cspline(const double *xa, const double *ya, int size, double *sc) {
  for (int i = 0; i < size; i++) sc[i] = 0.0;
  if (size < 3) return 1;
  int max_index = size - 1, sys_size = max_index - 1;
  double *g = (double *)malloc(sizeof(double) * sys_size);
  double *diag = (double *)malloc(sizeof(double) * sys_size);
  double *offdiag = (double *)malloc(sizeof(double) * sys_size);
  for (int i = 0; i < sys_size; i++) {
    const double h_i = xa[i + 1] - xa[i];
    const double h_ip1 = xa[i + 2] - xa[i + 1];
    const double ydiff_i = ya[i + 1] - ya[i];
    const double ydiff_ip1 = ya[i + 2] - ya[i + 1];
    const double g_i = (h_i != 0.0) ? 1.0 / h_i : 0.0;
    const double g_ip1 = (h_ip1 != 0.0) ? 1.0 / h_ip1 : 0.0;
    offdiag[i] = h_ip1;
    diag[i] = 2.0 * (h_ip1 + h_i);
    g[i] = 3.0 * (ydiff_ip1 * g_ip1 - ydiff_i * g_i);
  }
  if (sys_size == 1) {
    sc[1] = g[0] / diag[0];
  } else {
    const int N = sys_size;
    double *gamma = (double *)malloc(sizeof(double) * N);
    double *alpha = (double *)malloc(sizeof(double) * N);
    double *cc = (double *)malloc(sizeof(double) * N);
    double *z = (double *)malloc(sizeof(double) * N);
    double *x = sc + 1;
    alpha[0] = diag[0];
    gamma[0] = offdiag[0] / alpha[0];
    for (int i = 1; i < N - 1; i++) {
      alpha[i] = diag[i] - offdiag[i - 1] * gamma[i - 1];
      gamma[i] = offdiag[i] / alpha[i];
    }
    if (N > 1) alpha[N - 1] = diag[N - 1] - offdiag[N - 2] * gamma[N - 2];
    z[0] = g[0];
    for (int i = 1; i < N; i++) z[i] = g[i] - gamma[i - 1] * z[i - 1];
    for (int i = 0; i < N; i++) cc[i] = z[i] / alpha[i];
    x[N - 1] = cc[N - 1];
    if (N >= 2)
      for (int i = N - 2, j = 0; j <= N - 2; j++, i--) x[i] = cc[i] - gamma[i] * x[i + 1];
    free(gamma); free(alpha); free(cc); free(z);
  }
  free(g); free(diag); free(offdiag);
  return 0;
}
static int spline_init(orc_ctx *c, const double *xa, const double *ya, int size) {
  free(c->sx); free(c->sy); free(c->sc);
  c->nk = size;
  c->sx = (double *)malloc(sizeof(double) * size);
  c->sy = (double *)malloc(sizeof(double) * size);
  c->sc = (double *)calloc(size, sizeof(double));
  memcpy(c->sx, xa, sizeof(double) * size);
  memcpy(c->sy, ya, sizeof(double) * size);
  return natural_cspline(xa, ya, size, c->sc);
}

/* GSL interpolation/bsearch.c gsl_interp_bsearch + cspline.c cspline_eval */
static double gsl_spline_eval_arrays(const double *xa, const double *ya, const double *ca, int nk, double x) {
  size_t ilo = 0, ihi = (size_t)nk - 1;
  while (ihi > ilo + 1) {
    size_t i = (ihi + ilo) / 2;
    if (xa[i] > x) ihi = i; else ilo = i;
  }
  const size_t index = ilo;
  const double x_hi = xa[index + 1], x_lo = xa[index];
  const double dx = x_hi - x_lo;
  const double y_lo = ya[index], y_hi = ya[index + 1];
  const double dy = y_hi - y_lo;
  const double delx = x - x_lo;
  const double c_i = ca[index], c_ip1 = ca[index + 1];
  const double b_i = (dy / dx) - dx * (c_ip1 + 2.0 * c_i) / 3.0;
  const double d_i = (c_ip1 - c_i) / (3.0 * dx);
  return y_lo + delx * (b_i + delx * (c_i + delx * d_i));
}

static double gsl_spline_eval_restated(const orc_ctx *c, double x) { return gsl_spline_eval_arrays(c->sx, c->sy, c->sc, c->nk, x); }
/* my_spline_eval (src/cosmo.c:2016-2027) on explicit arrays */
static double my_spline_eval_arrays(const double *sx, const double *sy, const double *sc, int size, double x) {
  if (x < sx[0])
    return sy[0] + (x - sx[0]) * (sy[1] - sy[0]) / (sx[1] - sx[0]);
  else if (x > sx[size - 1])
    return sy[size - 1] + (x - sx[size - 1]) * (sy[size - 1] - sy[size - 2]) / (sx[size - 1] - sx[size - 2]);
  else
    return gsl_spline_eval_arrays(sx, sy, sc, size, x);
}

/* the same two for other files of the oracle (pf_genic.c: the tabulated power spectrum) */
int orc_cspline_coeffs(const double *xa, const double *ya, int size, double *sc) { return natural_cspline(xa, ya, size, sc); }
double orc_my_spline_eval(const double *sx, const double *sy, const double *sc, int size, double x) { return my_spline_eval_arrays(sx, sy, sc, size, x); }

/* cosmo.c:2016-2027 my_spline_eval: linear extrapolation beyond the knots */
double orc_spline_eval(orc_ctx *c, double x) {
  const double *sx = c->sx, *sy = c->sy;
  const int size = c->nk;
  if (x < sx[0])
    return sy[0] + (x - sx[0]) * (sy[1] - sy[0]) / (sx[1] - sx[0]);
  else if (x > sx[size - 1])
    return sy[size - 1] + (x - sx[size - 1]) * (sy[size - 1] - sy[size - 2]) / (sx[size - 1] - sx[size - 2]);
  else
    return gsl_spline_eval_restated(c, x);
}

/* cosmo.c:1822-1832 (non SCALE_DEPENDENT branch) */
double orc_inverse_growing_mode(orc_ctx *c, double D) {
  return 1. / pow(10., orc_spline_eval(c, log10(D))) - 1.;
}

/* ---------------------------------------- src/collapse_times.c restated -- */

/* collapse_times.c:114-221 */
double orc_ell_classic(double l1, double l2, double l3) {
  double ell;
  double del = l1 + l2 + l3;
  double det = l1 * l2 * l3;

  if (fabs(l1) < ORC_SMALL) {
    ell = -0.1;
  } else {
    double den = det / 126. + 5. * l1 * del * (del - l1) / 84.;
    if (fabs(den) < ORC_SMALL) {
      if (fabs(del - l1) < ORC_SMALL) {
        if (l1 > 0.0) ell = 1. / l1; else ell = -.1;
      } else {
        double dis = 7. * l1 * (l1 + 6. * del);
        if (dis < 0.0) {
          ell = -.1;
        } else {
          ell = (7. * l1 - sqrt(dis)) / (3. * l1 * (l1 - del));
          if (ell < 0.) ell = -.1;
        }
      }
    } else {
      double rden = 1.0 / den;
      double a1 = 3. * l1 * (del - l1) / 14. * rden;
      double a1_2 = a1 * a1;
      double a2 = l1 * rden;
      double a3 = -1.0 * rden;
      double q = (a1_2 - 3. * a2) / 9.;
      double r = (2. * a1_2 * a1 - 9. * a1 * a2 + 27. * a3) / 54.;
      double r_2_q_3 = r * r - q * q * q;
      if (r_2_q_3 > 0) {
        double fabs_r = fabs(r);
        double sq = pow(sqrt(r_2_q_3) + fabs_r, 0.333333333333333);
        ell = -fabs_r / r * (sq + q / sq) - a1 / 3.;
        if (ell < 0.) ell = -.1;
      } else {
        double sq = 2 * sqrt(q);
        double inv_3 = 1.0 / 3;
        double t = acos(2 * r / q / sq);
        double s1 = -sq * cos(t * inv_3) - a1 * inv_3;
        double s2 = -sq * cos((t + 2. * ORC_PI) * inv_3) - a1 * inv_3;
        double s3 = -sq * cos((t + 4. * ORC_PI) * inv_3) - a1 * inv_3;
        if (s1 < 0.) s1 = 1.e10;
        if (s2 < 0.) s2 = 1.e10;
        if (s3 < 0.) s3 = 1.e10;
        ell = (s1 < s2 ? s1 : s2);
        ell = (s3 < ell ? s3 : ell);
        if (ell == 1.e10) ell = -.1;
      }
    }
  }
  if (del > 0. && ell > 0.) {
    double inv_del = 1.0 / del;
    ell += -.364 * inv_del * exp(-6.5 * (l1 - l2) * inv_del - 2.8 * (l2 - l3) * inv_del);
  }
  return ell;
}

/* collapse_times.c:404-415 (ELL_CLASSIC) */
static double ell_fn(orc_ctx *c, double l1, double l2, double l3) {
  if (c->model == 1) { /* #ifdef ELL_SNG, :416 */
    double cosmo[7];
    memcpy(cosmo, c->sng_cosmo, sizeof(cosmo));
    cosmo[6] = c->sng_size[c->cur_ismooth];
    return orc_ell_sng_F(l1, l2, l3, c->sng_Din[c->cur_ismooth], cosmo);
  }
  double bc = orc_ell_classic(l1, l2, l3);
  if (bc > 0.0) return 1. + orc_inverse_growing_mode(c, bc);
  else return 0.0;
}

/* ------------------------------------------------ TABULATED_CT (src/collapse_times.c:780-1231), restated ---- */
#define CT_NBINS_XY (50)
#define CT_NBINS_D (100)
#define CT_SQUEEZE (1.2)
#define CT_EXPO (1.75)
#define CT_RANGE_D (7.0)
#define CT_RANGE_X (3.5)
#define CT_DELTA0 (-1.0)

/* the sampling in delta, finer around CT_DELTA0 (collapse_times.c:836-876; CT_EXPO is neither 1 nor 2) */
static void ct_delta_vector(double *delta_vector) {
  double deltaf = pow(CT_SQUEEZE / CT_EXPO, 1. / (CT_EXPO - 1.));
  double ref_interval = ((pow(CT_RANGE_D - CT_DELTA0, 2. - CT_EXPO) + pow(CT_RANGE_D + CT_DELTA0, 2. - CT_EXPO)
                          - 2. * pow(deltaf, 2. - CT_EXPO)) / CT_EXPO / (2. - CT_EXPO) + 2. * deltaf / CT_SQUEEZE) / (CT_NBINS_D - 2.0);
  double del = -CT_RANGE_D, interval;
  int id = 0;
  do {
    delta_vector[id] = del;
    interval = CT_EXPO * ref_interval * pow(fabs(del - CT_DELTA0), CT_EXPO - 1.0);
    interval = (interval / ref_interval < CT_SQUEEZE ? ref_interval * CT_SQUEEZE : interval);
    del += interval;
    id++;
  } while (id < CT_NBINS_D);
}

/* initialize_collapse_times(ismooth, 0) with params.CTtableFile == "none" (collapse_times.c:820-1043): the table of
   ell() and the CT_NBINS_XY^2 splines in delta; single task, so the MPI split of the computations is not restated */
static int ct_initialize(orc_ctx *c, int ismooth) {
  c->cur_ismooth = ismooth;
  const int Ncomputations = CT_NBINS_D * CT_NBINS_XY * CT_NBINS_XY;
  if (!c->ct_table) {
    c->ct_table = (double *)calloc(Ncomputations, sizeof(double));
    c->ct_c = (double *)calloc(Ncomputations, sizeof(double));
    c->ct_delta = (double *)malloc(CT_NBINS_D * sizeof(double));
    ct_delta_vector(c->ct_delta);
  }
  const double bin_x = CT_RANGE_X / (double)(CT_NBINS_XY);
  const double ampl = sqrt(c->tab_var[ismooth]);
  c->tab_ampl = ampl;
  const int was = c->tab_ns;
  c->tab_ns = 0; /* the table itself is made of true ell() values */
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
  for (int i = 0; i < Ncomputations; ++i) {
    int id = i % CT_NBINS_D;
    int ix = (i / CT_NBINS_D) % CT_NBINS_XY;
    int iy = i / CT_NBINS_D / CT_NBINS_XY;
    double x = ix * bin_x;
    double y = iy * bin_x;
    double l1 = (c->ct_delta[id] + 2. * x + y) / 3.0 * ampl;
    double l2 = (c->ct_delta[id] - x + y) / 3.0 * ampl;
    double l3 = (c->ct_delta[id] - x - 2. * y) / 3.0 * ampl;
    c->ct_table[i] = ell_fn(c, l1, l2, l3);
  }
  c->tab_ns = was;
  /* gsl_spline_init(CT_Spline[i][j], delta_vector, &CT_table[i*CT_NBINS_D + j*CT_NBINS_D*CT_NBINS_XY], CT_NBINS_D) */
  for (int i = 0; i < CT_NBINS_XY; ++i)
    for (int j = 0; j < CT_NBINS_XY; ++j) {
      const size_t off = (size_t)i * CT_NBINS_D + (size_t)j * CT_NBINS_D * CT_NBINS_XY;
      natural_cspline(c->ct_delta, c->ct_table + off, CT_NBINS_D, c->ct_c + off);
    }
  return 0;
}

/* GSL interpolation/cspline.c cspline_eval_deriv (what gsl_spline_eval_deriv calls) on explicit arrays */
static double gsl_spline_eval_deriv_arrays(const double *xa, const double *ya, const double *ca, int nk, double x) {
  size_t ilo = 0, ihi = (size_t)nk - 1;
  while (ihi > ilo + 1) {
    size_t i = (ihi + ilo) / 2;
    if (xa[i] > x) ihi = i; else ilo = i;
  }
  const size_t index = ilo;
  const double dx = xa[index + 1] - xa[index];
  const double dy = ya[index + 1] - ya[index];
  const double delx = x - xa[index];
  const double c_i = ca[index], c_ip1 = ca[index + 1];
  const double b_i = (dy / dx) - dx * (c_ip1 + 2.0 * c_i) / 3.0;
  const double d_i = (c_ip1 - c_i) / (3.0 * dx);
  return b_i + delx * (2.0 * c_i + 3.0 * d_i * delx);
}

/* GSL 2.7.1 interp2d/bicubic.c (not vendored; restated): bicubic_init -- the partial derivatives zx, zy, zxy at the
   nodes from natural cubic splines along the rows and columns -- and bicubic_eval -- the bicubic Hermite patch of the
   cell that holds (x, y).  za[j * 4 + i] = z(xa[i], ya[j]) as in IDX2D.  gsl_spline2d_eval refuses points outside the
   grid (GSL_EDOM, which aborts under the default handler); here the edge cell's patch is evaluated, as
   gsl_spline2d_eval_extrap does. */
static double gsl_bicubic_4x4(const double xa[4], const double ya[4], const double za[16], double x, double y) {
  double zx[16], zy[16], zxy[16], u[4], v[4], sc[4];
  for (int j = 0; j < 4; j++) { /* zx: a spline in x through every row */
    for (int i = 0; i < 4; i++) { u[i] = xa[i]; v[i] = za[j * 4 + i]; }
    natural_cspline(u, v, 4, sc);
    for (int i = 0; i < 4; i++) zx[j * 4 + i] = gsl_spline_eval_deriv_arrays(u, v, sc, 4, xa[i]);
  }
  for (int i = 0; i < 4; i++) { /* zy: a spline in y through every column */
    for (int j = 0; j < 4; j++) { u[j] = ya[j]; v[j] = za[j * 4 + i]; }
    natural_cspline(u, v, 4, sc);
    for (int j = 0; j < 4; j++) zy[j * 4 + i] = gsl_spline_eval_deriv_arrays(u, v, sc, 4, ya[j]);
  }
  for (int j = 0; j < 4; j++) { /* zxy: a spline in x through every row of zy */
    for (int i = 0; i < 4; i++) { u[i] = xa[i]; v[i] = zy[j * 4 + i]; }
    natural_cspline(u, v, 4, sc);
    for (int i = 0; i < 4; i++) zxy[j * 4 + i] = gsl_spline_eval_deriv_arrays(u, v, sc, 4, xa[i]);
  }
  size_t xi = 0, yi = 0; /* gsl_interp_bsearch over the whole grid */
  { size_t lo = 0, hi = 3; while (hi > lo + 1) { size_t i = (hi + lo) / 2; if (xa[i] > x) hi = i; else lo = i; } xi = lo; }
  { size_t lo = 0, hi = 3; while (hi > lo + 1) { size_t i = (hi + lo) / 2; if (ya[i] > y) hi = i; else lo = i; } yi = lo; }
#define IDX(i, j) ((j) * 4 + (i))
  const double xmin = xa[xi], xmax = xa[xi + 1], ymin = ya[yi], ymax = ya[yi + 1];
  const double zminmin = za[IDX(xi, yi)], zminmax = za[IDX(xi, yi + 1)], zmaxmin = za[IDX(xi + 1, yi)], zmaxmax = za[IDX(xi + 1, yi + 1)];
  const double dx = xmax - xmin, dy = ymax - ymin;
  const double t = (x - xmin) / dx, uu = (y - ymin) / dy;
  const double dt = 1. / dx, du = 1. / dy;
  const double zxminmin = zx[IDX(xi, yi)] / dt, zxminmax = zx[IDX(xi, yi + 1)] / dt, zxmaxmin = zx[IDX(xi + 1, yi)] / dt, zxmaxmax = zx[IDX(xi + 1, yi + 1)] / dt;
  const double zyminmin = zy[IDX(xi, yi)] / du, zyminmax = zy[IDX(xi, yi + 1)] / du, zymaxmin = zy[IDX(xi + 1, yi)] / du, zymaxmax = zy[IDX(xi + 1, yi + 1)] / du;
  const double zxyminmin = zxy[IDX(xi, yi)] / (dt * du), zxyminmax = zxy[IDX(xi, yi + 1)] / (dt * du), zxymaxmin = zxy[IDX(xi + 1, yi)] / (dt * du),
               zxymaxmax = zxy[IDX(xi + 1, yi + 1)] / (dt * du);
#undef IDX
  const double t0 = 1, t1 = t, t2 = t * t, t3 = t * t2, u0 = 1, u1 = uu, u2 = uu * uu, u3 = uu * u2;
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

/* interpolate_collapse_time (collapse_times.c:1139-1231).  The source defines BILINEAR_SPLINE; -DTRILINEAR or -DALL_SPLINE in
   the Makefile's OPTIONS (tests/Readme_Pinocchio_tests_V5_1.txt) put their own return in front of it. */
int orc_set_ct_interpolation(orc_ctx *c, int flavour) {
  if (flavour < 0 || flavour > 2) return 1;
  c->ct_flavour = flavour;
  return 0;
}
double orc_interpolate_collapse_time(orc_ctx *c, double l1, double l2, double l3) {
  const double bin_x = CT_RANGE_X / (double)(CT_NBINS_XY);
  double ampl = c->tab_ampl;
  double d = (l1 + l2 + l3) / ampl;
  double x = (l1 - l2) / ampl;
  double y = (l2 - l3) / ampl;
  int ix = (int)(x / bin_x);
  int iy = (int)(y / bin_x);
  ix = (ix >= CT_NBINS_XY - 1) ? CT_NBINS_XY - 2 : (ix < 0) ? 0 : ix;
  iy = (iy >= CT_NBINS_XY - 1) ? CT_NBINS_XY - 2 : (iy < 0) ? 0 : iy;
#define CT_SPL(I, J) my_spline_eval_arrays(c->ct_delta, c->ct_table + (size_t)(I) * CT_NBINS_D + (size_t)(J) * CT_NBINS_D * CT_NBINS_XY, \
                                           c->ct_c + (size_t)(I) * CT_NBINS_D + (size_t)(J) * CT_NBINS_D * CT_NBINS_XY, CT_NBINS_D, d)
  if (c->ct_flavour == 2) { /* #ifdef ALL_SPLINE, :1153-1185: bicubic on the 4x4 nodes around the cell */
    double xls[4], yls[4], zls[16];
    int ixstart = (ix == 0) ? 0 : (ix >= CT_NBINS_XY - 2) ? CT_NBINS_XY - 4 : ix - 1;
    int iystart = (iy == 0) ? 0 : (iy >= CT_NBINS_XY - 2) ? CT_NBINS_XY - 4 : iy - 1;
    for (int ixx = 0; ixx < 4; ixx++) {
      xls[ixx] = (ixx + ixstart) * bin_x;
      yls[ixx] = (ixx + iystart) * bin_x;
    }
    for (int ixx = 0; ixx < 4; ixx++)
      for (int iyy = 0; iyy < 4; iyy++) zls[ixx + iyy * 4] = CT_SPL(ixx + ixstart, iyy + iystart);
    return gsl_bicubic_4x4(xls, yls, zls, x, y);
  }
  if (c->ct_flavour == 1) { /* #ifdef TRILINEAR, :1189-1216 */
    const double *dv = c->ct_delta, *T = c->ct_table;
    int id;
    if (d <= dv[0]) id = 0;
    else if (d >= dv[CT_NBINS_D - 1]) id = CT_NBINS_D - 2;
    else { /* bsearch with compare_search (:1129-1135): dv[id] <= d < dv[id + 1] */
      int lo = 0, hi = CT_NBINS_D - 1;
      while (hi > lo + 1) { int m = (hi + lo) / 2; if (dv[m] > d) hi = m; else lo = m; }
      id = lo;
    }
    double dd = (d - dv[id]) / (dv[id + 1] - dv[id]);
    double dx = x / bin_x - ix;
    double dy = y / bin_x - iy;
    return (((1. - dd) * (1. - dx) * (1. - dy) * T[id + (ix) * CT_NBINS_D + (iy) * CT_NBINS_D * CT_NBINS_XY]) +
            ((dd) * (1. - dx) * (1. - dy) * T[(id + 1) + (ix) * CT_NBINS_D + (iy) * CT_NBINS_D * CT_NBINS_XY]) +
            ((1. - dd) * (dx) * (1. - dy) * T[id + (ix + 1) * CT_NBINS_D + (iy) * CT_NBINS_D * CT_NBINS_XY]) +
            ((dd) * (dx) * (1. - dy) * T[(id + 1) + (ix + 1) * CT_NBINS_D + (iy) * CT_NBINS_D * CT_NBINS_XY]) +
            ((1. - dd) * (1. - dx) * (dy) * T[id + (ix) * CT_NBINS_D + (iy + 1) * CT_NBINS_D * CT_NBINS_XY]) +
            ((dd) * (1. - dx) * (dy) * T[(id + 1) + (ix) * CT_NBINS_D + (iy + 1) * CT_NBINS_D * CT_NBINS_XY]) +
            ((1. - dd) * (dx) * (dy) * T[id + (ix + 1) * CT_NBINS_D + (iy + 1) * CT_NBINS_D * CT_NBINS_XY]) +
            ((dd) * (dx) * (dy) * T[(id + 1) + (ix + 1) * CT_NBINS_D + (iy + 1) * CT_NBINS_D * CT_NBINS_XY]));
  }
  double dx = x / bin_x - ix;
  double dy = y / bin_x - iy;
  return ((1. - dx) * (1. - dy) * CT_SPL(ix, iy) +
          (dx) * (1. - dy) * CT_SPL(ix + 1, iy) +
          (1. - dx) * (dy) * CT_SPL(ix, iy + 1) +
          (dx) * (dy) * CT_SPL(ix + 1, iy + 1));
#undef CT_SPL
}

/* collapse_times.c:1354-1362 */
static void ord(double *a, double *b, double *c) {
  double lo, hi;
  hi = (*a > *b ? *a : *b); hi = (hi > *c ? hi : *c);
  lo = (*a < *b ? *a : *b); lo = (lo < *c ? lo : *c);
  *b = *a + *b + *c - lo - hi;
  *a = hi;
  *c = lo;
}

/* collapse_times.c:679-776 */
double orc_inverse_collapse_time(orc_ctx *c, const double *deformation_tensor, double *x1, double *x2,
                                 double *x3, int *fail) {
  double mu1, mu2, mu3;
  double dtensor[6] = {deformation_tensor[0], deformation_tensor[1], deformation_tensor[2],
                       deformation_tensor[3], deformation_tensor[4], deformation_tensor[5]};
  *fail = 0;
  mu1 = dtensor[0] + dtensor[1] + dtensor[2];
  double mu1_2 = mu1 * mu1;
  mu2 = 0.5 * mu1_2;
  {
    double add[3];
    add[0] = dtensor[0] * dtensor[0];
    add[1] = dtensor[1] * dtensor[1];
    add[2] = dtensor[2] * dtensor[2];
    mu2 -= 0.5 * (add[0] + add[1] + add[2]);
  }
  double add[3];
  add[0] = dtensor[3] * dtensor[3];
  add[1] = dtensor[4] * dtensor[4];
  add[2] = dtensor[5] * dtensor[5];
  mu2 -= add[0] + add[1] + add[2];
  mu3 = dtensor[0] * dtensor[1] * dtensor[2] + 2. * dtensor[3] * dtensor[4] * dtensor[5] -
        dtensor[0] * add[2] - dtensor[1] * add[1] - dtensor[2] * add[0];
  double q;
  q = (mu1_2 - 3.0 * mu2) / 9.0;
  if (q == 0.) {
    *x1 = dtensor[0]; *x2 = dtensor[1]; *x3 = dtensor[2];
  } else {
    double r = -(2. * mu1_2 * mu1 - 9.0 * mu1 * mu2 + 27.0 * mu3) / 54.;
    if (q * q * q < r * r || q < 0.0) return -10.0;
    double sq = 2 * sqrt(q);
    double t = acos(2 * r / q / sq);
    double inv_3 = 1.0 / 3.0;
    *x1 = -sq * cos(t * inv_3) + mu1 * inv_3;
    *x2 = -sq * cos((t + 2. * ORC_PI) * inv_3) + mu1 * inv_3;
    *x3 = -sq * cos((t + 4. * ORC_PI) * inv_3) + mu1 * inv_3;
  }
  ord(x1, x2, x3);
  if (c->tab_ns > 0) return orc_interpolate_collapse_time(c, *x1, *x2, *x3); /* #ifdef TABULATED_CT, :749 */
  return ell_fn(c, *x1, *x2, *x3);
}

/* collapse_times.c:431-673 */
static int compute_collapse_times_with_current_spline(orc_ctx *c, int ismooth, double *true_var) {
  const size_t nr = c->n_r;
  c->cur_ismooth = ismooth;
  if (c->tab_ns > 0 && ct_initialize(c, ismooth)) return 1; /* src/fmax.c:103-106 */
  orc_product *products = c->products;
  double local_variance = 0.0, local_average = 0.0;

  if (!ismooth) {
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
    for (size_t i = 0; i < nr; i++) {
      products[i].Fmax = -10.0;
      products[i].Rmax = -1;
      for (int k = 0; k < 3; k++) {
        products[i].Vel[k] = 0.0; products[i].Vel_2LPT[k] = 0.0;
        products[i].Vel_3LPT_1[k] = 0.0; products[i].Vel_3LPT_2[k] = 0.0;
      }
    }
  }
  int all_fails = 0;
  /* per-thread partial sums combined in thread order (deterministic for a
     fixed thread count; the reference uses omp atomic, :602-606) */
  int nt = c->nthreads;
  double *pv = (double *)calloc(nt, sizeof(double)), *pa = (double *)calloc(nt, sizeof(double));
#pragma omp parallel num_threads(nt)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double mylocal_average = 0, mylocal_variance = 0;
    int fails = 0;
#pragma omp for schedule(static) nowait
    for (size_t index = 0; index < nr; index++) {
      double diff_ten[6];
      for (int i = 0; i < 6; i++) diff_ten[i] = c->second_derivatives[i][index];
      double delta = diff_ten[0] + diff_ten[1] + diff_ten[2];
      mylocal_average += delta;
      mylocal_variance += delta * delta;
      double lambda1, lambda2, lambda3;
      int fail;
      double Fnew = orc_inverse_collapse_time(c, diff_ten, &lambda1, &lambda2, &lambda3, &fail);
      if (fail) fails = 1;
      if (products[index].Fmax < Fnew) { /* float promoted to double (quirk Q2) */
        products[index].Fmax = Fnew;
        products[index].Rmax = ismooth;
      }
    }
    pv[tid] = mylocal_variance; pa[tid] = mylocal_average;
#pragma omp atomic
    all_fails += fails;
  }
  for (int t = 0; t < nt; t++) { local_variance += pv[t]; local_average += pa[t]; }
  free(pv); free(pa);
  if (all_fails) {
    printf("ERROR on task 0: failure in inverse_collapse_time\n");
    return 1;
  }
  double global_variance = local_variance / (double)nr; /* Ntotal, :662 */
  if (true_var) *true_var = global_variance;
  return 0;
}

/* collapse_times.c:431; in a SCALE_DEPENDENT build InverseGrowingMode(D, ismooth) reads SPLINE_INVGROW[ismooth] */
int orc_compute_collapse_times(orc_ctx *c, int ismooth, double *true_var) {
  if (ismooth < 0 || ismooth >= ORC_MAX_SMOOTH || !c->rnk[ismooth]) return compute_collapse_times_with_current_spline(c, ismooth, true_var);
  double *dx = c->sx, *dy = c->sy, *dc = c->sc;
  const int dn = c->nk;
  c->sx = c->rsx[ismooth]; c->sy = c->rsy[ismooth]; c->sc = c->rsc[ismooth]; c->nk = c->rnk[ismooth];
  const int rc = compute_collapse_times_with_current_spline(c, ismooth, true_var);
  c->sx = dx; c->sy = dy; c->sc = dc; c->nk = dn;
  return rc;
}

/* ----------------------------------------------- src/fmax.c restated ---- */

/* fmax.c:225-258 */
int orc_compute_second_derivatives(orc_ctx *c, double rs_cells) {
  c->Rsmooth = rs_cells;
  for (int ia = 1; ia <= 3; ia++)
    for (int ib = ia; ib <= 3; ib++) {
      int ider = (ia == ib ? ia : ia + ib + 1);
      write_in_cvector(c, c->kdensity);
      if (compute_derivative(c, ia, ib)) return 1;
      write_from_rvector(c, c->second_derivatives[ider - 1]);
    }
  return 0;
}

/* fmax.c:193-222 */
static int compute_first_derivatives(orc_ctx *c, double rs_cells, int order, const double *vector) {
  c->Rsmooth = rs_cells;
  for (int ia = 1; ia <= 3; ia++) {
    write_in_cvector(c, vector);
    if (compute_derivative(c, ia, 0)) return 1;
    write_from_rvector_to_products(c, ia - 1, order);
  }
  return 0;
}

/* src/LPT.c:32-235 */
static int compute_LPT_displacements(orc_ctx *c, int compute_sources) {
  const size_t nr = c->n_r;
  double **sd = c->second_derivatives;
  if (compute_sources) {
    c->sd_order = 0;
    double *source_2LPT = c->source_2LPT, *source_3LPT_1 = c->source_3LPT_1, *source_3LPT_2 = c->source_3LPT_2;
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
    for (size_t index = 0; index < nr; index++) {
      source_2LPT[index] = sd[0][index] * sd[1][index] + sd[0][index] * sd[2][index] +
                           sd[1][index] * sd[2][index] - sd[3][index] * sd[3][index] -
                           sd[4][index] * sd[4][index] - sd[5][index] * sd[5][index];
      source_3LPT_1[index] =
          3.0 * (sd[0][index] * (sd[1][index] * sd[2][index] - sd[5][index] * sd[5][index]) -
                 sd[3][index] * (sd[3][index] * sd[2][index] - sd[4][index] * sd[5][index]) +
                 sd[4][index] * (sd[3][index] * sd[5][index] - sd[4][index] * sd[1][index]));
      source_3LPT_2[index] = 2.0 * (sd[0][index] + sd[1][index] + sd[2][index]) * source_2LPT[index];
    }
    write_in_rvector(c, source_2LPT);
    c->t_fft += forward_transform(c);
    write_from_cvector(c, c->kvector_2LPT);

    for (int ia = 1; ia <= 3; ia++)
      for (int ib = ia; ib <= 3; ib++) {
        int ider = (ia == ib ? ia : ia + ib + 1);
        write_in_cvector(c, c->kvector_2LPT);
        if (compute_derivative(c, ia, ib)) return 1;
        const double *rv = c->rvector;
        const double *sdi = sd[ider - 1];
        const double f = 2.0 * (ider <= 3 ? 1.0 : 2.0);
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
        for (size_t index = 0; index < nr; index++) source_3LPT_2[index] -= f * rv[index] * sdi[index];
      }
    write_in_rvector(c, source_3LPT_1);
    c->t_fft += forward_transform(c);
    write_from_cvector(c, c->kvector_3LPT_1);

    write_in_rvector(c, source_3LPT_2);
    c->t_fft += forward_transform(c);
    write_from_cvector(c, c->kvector_3LPT_2);
  }
  c->sd_order = 2;
  compute_first_derivatives(c, 0., 2, c->kvector_2LPT);
  c->sd_order = 3;
  compute_first_derivatives(c, 0., 3, c->kvector_3LPT_1);
  c->sd_order = 4;
  compute_first_derivatives(c, 0., 4, c->kvector_3LPT_2);
  return 0;
}

/* fmax.c:292-367 with recompute_sd = 0 */
int orc_compute_displacements(orc_ctx *c, int compute_sources) {
  double t = now_s();
  if (compute_LPT_displacements(c, compute_sources)) return 1;
  c->t_lpt += now_s() - t;
  t = now_s();
  c->sd_order = 1;
  if (compute_first_derivatives(c, 0.0, 1, c->kdensity)) return 1;
  c->t_deriv += now_s() - t;
  return 0;
}

/* fmax.c:36-190 */
int orc_compute_fmax(orc_ctx *c, int ns, const double *rs_cells, int do_lpt, double *true_var) {
  c->t_total = now_s();
  c->t_deriv = c->t_fft = c->t_coll = c->t_lpt = 0.0;
  c->sd_order = 0;
  for (int ismooth = 0; ismooth < ns; ismooth++) {
    double t = now_s();
    if (orc_compute_second_derivatives(c, rs_cells[ismooth])) return 1;
    c->t_deriv += now_s() - t;
    t = now_s();
    if (orc_compute_collapse_times(c, ismooth, true_var ? &true_var[ismooth] : NULL)) return 1;
    c->t_coll += now_s() - t;
  }
  if (do_lpt)
    if (orc_compute_displacements(c, 1)) return 1;
  c->t_total = now_s() - c->t_total;
  return 0;
}

/* fmax.c:509-550 */
int orc_fmax_pdf(orc_ctx *c, unsigned long long hist[ORC_NBINS]) {
  for (int i = 0; i < ORC_NBINS; i++) hist[i] = 0;
  for (size_t i = 0; i < c->n_r; i++) {
    int xF = (int)(c->products[i].Fmax * 10.);
    if (xF < 0) xF = 0;
    if (xF >= ORC_NBINS) xF = ORC_NBINS - 1;
    hist[xF]++;
  }
  return 0;
}

/* ----------------------------------------------------------- plumbing --- */

orc_ctx *orc_create(int n, int nthreads) {
  if (n < 4 || (n & 1)) return NULL; /* even sizes; powers of two use the radix-2 transform */
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->n = n; c->nzh = n / 2 + 1;
#ifdef _OPENMP
  c->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  c->nthreads = 1;
#endif
  c->n_r = (size_t)n * n * n;
  c->n_fft = 2 * (size_t)n * n * c->nzh;
  c->norm = (double)1.0 / ((double)c->n_r);
  c->tw = (double *)malloc(sizeof(double) * 2 * n);
  for (int j = 0; j < n; j++) {
    c->tw[2 * j] = cos(2. * ORC_PI * j / n);
    c->tw[2 * j + 1] = sin(2. * ORC_PI * j / n);
  }
  c->brev = (int *)malloc(sizeof(int) * n);
  int lg = 0; while ((1 << lg) < n) lg++;
  for (int i = 0; i < n; i++) {
    int r = 0;
    for (int b = 0; b < lg; b++) if (i & (1 << b)) r |= 1 << (lg - 1 - b);
    c->brev[i] = r;
  }
  size_t bf = sizeof(double) * c->n_fft;
  c->kdensity = (double *)calloc(1, bf);
  c->cvector = (double *)calloc(1, bf);
  c->rvector = (double *)calloc(1, bf);
  for (int i = 0; i < 6; i++) c->second_derivatives[i] = (double *)calloc(c->n_r, sizeof(double));
  c->kvector_2LPT = (double *)calloc(1, bf);
  c->kvector_3LPT_1 = (double *)calloc(1, bf);
  c->kvector_3LPT_2 = (double *)calloc(1, bf);
  c->source_2LPT = c->kvector_2LPT; c->source_3LPT_1 = c->kvector_3LPT_1; c->source_3LPT_2 = c->kvector_3LPT_2;
  c->products = (orc_product *)calloc(c->n_r, sizeof(orc_product));
  /* first touch by the threads that will work on the x-planes (the pages of a calloc'ed array are placed where they are
     first written: by one thread they would all sit on one memory node of a two-socket host) */
  {
    double *big[12] = {c->kdensity, c->cvector, c->rvector, c->kvector_2LPT, c->kvector_3LPT_1, c->kvector_3LPT_2,
                       c->second_derivatives[0], c->second_derivatives[1], c->second_derivatives[2],
                       c->second_derivatives[3], c->second_derivatives[4], c->second_derivatives[5]};
    const size_t cnt[12] = {c->n_fft, c->n_fft, c->n_fft, c->n_fft, c->n_fft, c->n_fft, c->n_r, c->n_r, c->n_r, c->n_r, c->n_r, c->n_r};
    for (int b = 0; b < 12; b++) {
      const size_t chunk = (cnt[b] + (size_t)n - 1) / (size_t)n;
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
      for (int i = 0; i < n; i++) {
        const size_t lo = (size_t)i * chunk, hi = lo + chunk < cnt[b] ? lo + chunk : cnt[b];
        if (lo < hi) memset(big[b] + lo, 0, sizeof(double) * (hi - lo));
      }
    }
    const size_t pchunk = (c->n_r + (size_t)n - 1) / (size_t)n;
#pragma omp parallel for num_threads(c->nthreads) schedule(static)
    for (int i = 0; i < n; i++) {
      const size_t lo = (size_t)i * pchunk, hi = lo + pchunk < c->n_r ? lo + pchunk : c->n_r;
      if (lo < hi) memset(c->products + lo, 0, sizeof(orc_product) * (hi - lo));
    }
  }
  c->growth[0] = 1.0; c->growth[1] = 3. / 7.; c->growth[2] = -1. / 9.; c->growth[3] = 5. / 42.;
  return c;
}

/* A context WITHOUT the field arrays: twiddles, splines and scalars only.  For the sampled-plane functions below
   (orc_plane_*), which restate the path on a few x-planes of a box whose whole oracle would not fit the host (the 1024^3
   of the metric: ~450 GB with the test's own arrays).  Every whole-box entry point needs orc_create. */
orc_ctx *orc_create_planes(int n, int nthreads) {
  if (n < 4 || (n & 1)) return NULL;
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->n = n; c->nzh = n / 2 + 1;
#ifdef _OPENMP
  c->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  c->nthreads = 1;
#endif
  c->n_r = 0; c->n_fft = 0;  /* no fields */
  c->norm = (double)1.0 / ((double)n * (double)n * (double)n);
  c->tw = (double *)malloc(sizeof(double) * 2 * n);
  for (int j = 0; j < n; j++) {
    c->tw[2 * j] = cos(2. * ORC_PI * j / n);
    c->tw[2 * j + 1] = sin(2. * ORC_PI * j / n);
  }
  c->brev = (int *)malloc(sizeof(int) * n);
  int lg = 0; while ((1 << lg) < n) lg++;
  for (int i = 0; i < n; i++) {
    int r = 0;
    for (int b = 0; b < lg; b++) if (i & (1 << b)) r |= 1 << (lg - 1 - b);
    c->brev[i] = r;
  }
  c->growth[0] = 1.0; c->growth[1] = 3. / 7.; c->growth[2] = -1. / 9.; c->growth[3] = 5. / 42.;
  return c;
}

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  free(c->tw); free(c->brev); free(c->kdensity); free(c->cvector); free(c->rvector);
  for (int i = 0; i < 6; i++) free(c->second_derivatives[i]);
  free(c->kvector_2LPT); free(c->kvector_3LPT_1); free(c->kvector_3LPT_2);
  free(c->products); free(c->sx); free(c->sy); free(c->sc);
  for (int i = 0; i < ORC_MAX_SMOOTH; i++) { free(c->rsx[i]); free(c->rsy[i]); free(c->rsc[i]); }
  free(c->ct_table); free(c->ct_c); free(c->ct_delta);
  free(c);
}

int orc_set_density(orc_ctx *c, const double *dk) { par_copy(c, c->kdensity, dk, c->n_fft); return 0; }
int orc_set_invgrow(orc_ctx *c, const double *x, const double *y, int nk) { return spline_init(c, x, y, nk); }
int orc_set_invgrow_radius(orc_ctx *c, int ismooth, const double *x, const double *y, int nk) {
  if (ismooth < 0 || ismooth >= ORC_MAX_SMOOTH) return 1;
  double *dx = c->sx, *dy = c->sy, *dc = c->sc;
  const int dn = c->nk;
  c->sx = c->sy = c->sc = NULL;
  const int rc = spline_init(c, x, y, nk);
  free(c->rsx[ismooth]); free(c->rsy[ismooth]); free(c->rsc[ismooth]);
  c->rsx[ismooth] = c->sx; c->rsy[ismooth] = c->sy; c->rsc[ismooth] = c->sc; c->rnk[ismooth] = nk;
  c->sx = dx; c->sy = dy; c->sc = dc; c->nk = dn;
  return rc;
}
int orc_set_collapse_model(orc_ctx *c, int model, const double cosmo[4], int ns, const double *D_in) {
  if (model < 0 || model > 1 || ns < 0 || ns > ORC_MAX_SMOOTH) return 1;
  c->model = model;
  if (model == 1) {
    memcpy(c->sng_cosmo, cosmo, sizeof(double) * 4);
    memcpy(c->sng_Din, D_in, sizeof(double) * ns);
  }
  return 0;
}
int orc_set_modified_gravity(orc_ctx *c, double fr0, double h_over_c, int ns, const double *size) {
  if (ns < 0 || ns > ORC_MAX_SMOOTH) return 1;
  c->sng_cosmo[4] = fr0; c->sng_cosmo[5] = h_over_c;
  if (ns) memcpy(c->sng_size, size, sizeof(double) * ns);
  return 0;
}
int orc_set_tabulated_ct(orc_ctx *c, int ns, const double *variance) {
  if (ns < 0 || ns > ORC_MAX_SMOOTH) return 1;
  c->tab_ns = ns;
  if (ns) memcpy(c->tab_var, variance, sizeof(double) * ns);
  return 0;
}
int orc_ct_build(orc_ctx *c, int ismooth, double variance) {
  if (ismooth < 0 || ismooth >= ORC_MAX_SMOOTH) return 1;
  c->tab_var[ismooth] = variance;
  return ct_initialize(c, ismooth);
}
const double *orc_ct_table(orc_ctx *c) { return c->ct_table; }
const double *orc_ct_delta(orc_ctx *c) { return c->ct_delta; }
int orc_set_growth_table(orc_ctx *c, int order, const double *T, int nk, double logkmin, double dlogk, double sign) {
  if (order < 1 || order > 4 || nk < 0 || nk > ORC_MAX_KBINS) return 1;
  c->gt_n[order - 1] = nk;
  if (nk) memcpy(c->gt_T[order - 1], T, sizeof(double) * nk);
  c->gt_logkmin[order - 1] = logkmin; c->gt_dlogk[order - 1] = dlogk; c->gt_sign[order - 1] = sign;
  return 0;
}
int orc_set_growth(orc_ctx *c, const double g[4]) { memcpy(c->growth, g, sizeof(double) * 4); return 0; }
const orc_product *orc_products(orc_ctx *c) { return c->products; }
const double *orc_second_derivative(orc_ctx *c, int i) { return c->second_derivatives[i]; }
const double *orc_kvector(orc_ctx *c, int which) {
  return which == 0 ? c->kvector_2LPT : which == 1 ? c->kvector_3LPT_1 : c->kvector_3LPT_2;
}
int orc_c2r(orc_ctx *c, const double *spec, double *real_out) {
  memcpy(c->cvector, spec, sizeof(double) * c->n_fft);
  c2r_3d(c, c->cvector, real_out);
  return 0;
}
int orc_r2c(orc_ctx *c, const double *real_in, double *spec_out) { r2c_3d(c, real_in, spec_out); return 0; }
void orc_timers(orc_ctx *c, double t[5]) {
  t[0] = c->t_total; t[1] = c->t_deriv; t[2] = c->t_fft; t[3] = c->t_coll; t[4] = c->t_lpt;
}

/* ----------------------------------------------------------------------------------------------------------------
 * Row f-2: the first stage of fragmentation on the products -- selection Fmax >= Flast (update_distmap,
 * src/distribute.c:695) and sort_and_organize's first qsort (src/fragment.c:484-503) with index_compare_F (:118-126).
 * qsort leaves the order of equal keys unspecified; the comparator below breaks ties by ascending index so that the
 * result is unique (and is one of the orders the reference's qsort may return).
 * ---------------------------------------------------------------------------------------------------------------- */
static const orc_product *sort_products;
static int index_compare_F(const void *a, const void *b) {
  const unsigned int ia = *(const unsigned int *)a, ib = *(const unsigned int *)b;
  if (sort_products[ia].Fmax == sort_products[ib].Fmax) return (ia > ib) - (ia < ib);
  else if (sort_products[ia].Fmax > sort_products[ib].Fmax) return -1;
  else return 1;
}
size_t orc_select_sorted(orc_ctx *c, float flast, unsigned int *indices, float *fmax) {
  size_t m = 0;
  for (size_t i = 0; i < c->n_r; i++)
    if (c->products[i].Fmax >= flast) indices[m++] = (unsigned int)i;
  sort_products = c->products;
  qsort((void *)indices, m, sizeof(unsigned int), index_compare_F);
  for (size_t i = 0; i < m; i++) fmax[i] = c->products[indices[i]].Fmax;
  return m;
}

/* the y and z transforms of nfields planes G[nfields][n][nzh] (complex, [ky][kz]) into out[nfields][n][n] reals times 1/n^3:
   y lines of every plane (fixed kz, stride nzh), then the c2r rows along z -- what c2r_3d does for a whole box */
static void plane_finish(orc_ctx *c, double *G, int nfields, double *out) {
  const int n = c->n, nzh = c->nzh;
  const size_t plane_c = (size_t)n * nzh;
#pragma omp parallel num_threads(c->nthreads)
  {
    double *lines = (double *)malloc(sizeof(double) * 2 * n * ORC_FFT_BLOCK);
    const int nblk = (nzh + ORC_FFT_BLOCK - 1) / ORC_FFT_BLOCK;
#pragma omp for schedule(dynamic, 1)
    for (int w = 0; w < nfields * nblk; w++) {
      const int kz = (w % nblk) * ORC_FFT_BLOCK;
      double *g = G + 2 * ((size_t)(w / nblk) * plane_c);
      fft_strided_lines(c, g + 2 * kz, (size_t)nzh, nzh - kz < ORC_FFT_BLOCK ? nzh - kz : ORC_FFT_BLOCK, +1, lines);
    }
    free(lines);
    double *line = (double *)malloc(sizeof(double) * 2 * n);
#pragma omp for schedule(dynamic, 16)
    for (long long w = 0; w < (long long)nfields * n; w++) {
      const double *h = G + 2 * ((size_t)w * nzh);
      for (int k = 0; k < nzh; k++) { line[2 * k] = h[2 * k]; line[2 * k + 1] = h[2 * k + 1]; }
      for (int k = nzh; k < n; k++) { line[2 * k] = h[2 * (n - k)]; line[2 * k + 1] = -h[2 * (n - k) + 1]; }
      line[1] = 0.0; line[2 * (n / 2) + 1] = 0.0;
      fft1d(line, n, +1, c->tw, c->brev);
      double *r = out + (size_t)w * n;
      for (int z = 0; z < n; z++) r[z] = line[2 * z] * c->norm;
    }
    free(line);
  }
}

/* ----------------------------------------------------------------------------------------------------------------
 * Sampled x-planes of a box that is too large for the whole oracle.
 *
 * compute_derivative (src/fmax-pfft.c:255-441) multiplies every mode by green * smoothing * growth and transforms back
 * (reverse_transform, :203-228).  The value of that 3-D c2r on ONE x-plane needs only
 *     G(ky, kz) = sum_kx  cvector(kx, ky, kz) e^{+2 pi i kx x / n}
 * followed by the y and z transforms of that plane -- the very same operations per mode, the x-transform written as the
 * plain sum for the sampled x only (O(n) per mode and plane instead of the FFT's O(log n) per mode for all planes).
 * ncomp components (ia[c], ib[c]) as compute_derivative takes them: (a, b) with 1 <= a <= b <= 3 a second derivative,
 * (-1, -1) the plain transform.  First derivatives (the `swap` of :389-396) are not needed by the callers and refused.
 * ScaleDep.order = 0 (second derivatives: growth_rate = 1, :344-364).
 * spec: [n][n][n/2+1] complex (not modified); out: [ncomp][nplanes][n][n] reals, times 1/n^3 (:220-225).
 * ---------------------------------------------------------------------------------------------------------------- */
struct orc_plane_acc;
orc_plane_acc *orc_plane_acc_create(orc_ctx *c, int nrad, const double *rs_cells, int ncomp, const int *ia, const int *ib, int nplanes, const int *xs);
int orc_plane_acc_add(orc_plane_acc *a, const double *rows, int kx0, int nkx);
int orc_plane_acc_finish(orc_plane_acc *a, int irad, double *out);
void orc_plane_acc_destroy(orc_plane_acc *a);
int orc_plane_derivatives(orc_ctx *c, const double *spec, double rs_cells, int ncomp, const int *ia, const int *ib,
                          int nplanes, const int *xs, double *out) {
  /* (one radius, the whole spectrum in one piece, through the accumulators of the streaming form below: one implementation) */
  orc_plane_acc *a = orc_plane_acc_create(c, 1, &rs_cells, ncomp, ia, ib, nplanes, xs);
  if (!a) return 1;
  int rc = orc_plane_acc_add(a, spec, 0, c->n);
  if (!rc) rc = orc_plane_acc_finish(a, 0, out);
  orc_plane_acc_destroy(a);
  return rc;
}

/* The streaming form, for a spectrum that does not fit the host (BASELINE config 5: 2048^3): the rows of the spectrum arrive in pieces of
   consecutive kx, in ascending order, and several radii are accumulated in one go.  Per (ky, kz) the sum over kx runs in ascending order
   with compute_derivative's expressions per mode, whatever the pieces (tests/test_oracle.py: pieces of 1, 5, 2 ... rows against one piece, to the bit).  G: [nrad][ncomp][nplanes][n][nzh] complex. */
struct orc_plane_acc {
  orc_ctx *c;
  int nrad, ncomp, nplanes, next_kx;
  double rs[ORC_MAX_SMOOTH];
  int ia[6], ib[6];
  int *xs;
  double *E, *G;
  double *pw;   /* [n (ky)][nrad]: sum over the rows added so far of w |spec|^2 smoothing^2, w = 2 for 0 < kz < n/2 (the half-spectrum) */
};
orc_plane_acc *orc_plane_acc_create(orc_ctx *c, int nrad, const double *rs_cells, int ncomp, const int *ia, const int *ib, int nplanes, const int *xs) {
  if (!c || nrad < 1 || nrad > ORC_MAX_SMOOTH || ncomp < 1 || ncomp > 6 || nplanes < 1) return NULL;
  const int n = c->n, nzh = c->nzh;
  for (int k = 0; k < ncomp; k++) {
    const int plain = ia[k] == -1 && ib[k] == -1;
    if (!plain && !(ia[k] >= 1 && ia[k] <= 3 && ib[k] >= 1 && ib[k] <= 3)) return NULL;
  }
  for (int p = 0; p < nplanes; p++) if (xs[p] < 0 || xs[p] >= n) return NULL;
  orc_plane_acc *a = (orc_plane_acc *)calloc(1, sizeof(*a));
  a->c = c; a->nrad = nrad; a->ncomp = ncomp; a->nplanes = nplanes; a->next_kx = 0;
  memcpy(a->rs, rs_cells, sizeof(double) * nrad); memcpy(a->ia, ia, sizeof(int) * ncomp); memcpy(a->ib, ib, sizeof(int) * ncomp);
  a->xs = (int *)malloc(sizeof(int) * nplanes); memcpy(a->xs, xs, sizeof(int) * nplanes);
  a->E = (double *)malloc(sizeof(double) * 2 * (size_t)n * nplanes);
  for (int p = 0; p < nplanes; p++)
    for (int idx = 0; idx < n; idx++) {
      const int j = (int)(((long long)idx * xs[p]) % n);
      a->E[2 * ((size_t)p * n + idx)] = c->tw[2 * j]; a->E[2 * ((size_t)p * n + idx) + 1] = c->tw[2 * j + 1];
    }
  a->G = (double *)calloc((size_t)2 * n * nzh * nrad * ncomp * nplanes, sizeof(double));
  a->pw = (double *)calloc((size_t)n * nrad, sizeof(double));
  if (!a->G || !a->pw) { free(a->G); free(a->pw); free(a->E); free(a->xs); free(a); return NULL; }
  return a;
}
/* rows kx0 .. kx0 + nkx - 1: [nkx][n][nzh] complex.  Pieces must follow each other (kx0 = where the last one ended).
   Per (ky, kx) row of the spectrum: first the filtered modes of every (radius, component) -- the expressions of compute_derivative,
   mode by mode -- into a scratch row, then one multiply-add per plane and accumulator with the row's e^{2 pi i kx x / n}: loops over kz
   with unit stride.  The order of the sum over kx of every accumulator element is the order of the rows: ascending, as in the
   plain triple loop this replaces (round 6: 2800 cycles per mode there, the accumulators touched in 36 interleaved streams). */
int orc_plane_acc_add(orc_plane_acc *a, const double *rows, int kx0, int nkx) {
  orc_ctx *c = a->c;
  const int n = c->n, nzh = c->nzh, Nhalf = n / 2, nrad = a->nrad, ncomp = a->ncomp, nplanes = a->nplanes;
  const double knorm = 2. * ORC_PI / (double)n;
  if (kx0 != a->next_kx || nkx < 1 || kx0 + nkx > n) return 1;
  const size_t plane_c = (size_t)n * nzh;
#pragma omp parallel num_threads(c->nthreads)
  {
    double *fre = (double *)malloc(sizeof(double) * (size_t)nrad * ncomp * nzh), *fim = (double *)malloc(sizeof(double) * (size_t)nrad * ncomp * nzh);
    double *pwr = (double *)malloc(sizeof(double) * nrad);
#pragma omp for schedule(dynamic, 1)
    for (int idy = 0; idy < n; idy++) {
      int ii[3];
      ii[1] = idy; if (ii[1] > Nhalf) ii[1] -= n;
      const double k_y = knorm * ii[1];
      for (int r = 0; r < nrad; r++) pwr[r] = a->pw[(size_t)idy * nrad + r];
      for (int idx = kx0; idx < kx0 + nkx; idx++) {
        ii[0] = idx; if (ii[0] > Nhalf) ii[0] -= n;
        const double k_x = knorm * ii[0];
        const double k2_0 = k_x * k_x;
        const double k2_1 = k2_0 + k_y * k_y;
        const double *row = rows + 2 * (((size_t)(idx - kx0) * n + idy) * nzh);
        for (int idz = 0; idz < nzh; idz++) {
          ii[2] = idz; if (ii[2] > Nhalf) ii[2] -= n;
          const double k_z = knorm * ii[2];
          const double k_squared = k2_1 + k_z * k_z;
          double diff_comp[4];
          diff_comp[0] = 1.0; diff_comp[1] = k_x; diff_comp[2] = k_y; diff_comp[3] = k_z;
          double green[6];
          for (int k = 0; k < ncomp; k++) green[k] = k_squared != 0. ? greens_function(diff_comp, k_squared, a->ia[k], a->ib[k]) : 1.0;
          const double wgt = (idz == 0 || 2 * idz == n) ? 1.0 : 2.0;
          for (int r = 0; r < nrad; r++) {
            const double Rsmooth = a->rs[r];
            double smoothing = 1.0;
            if (k_squared != 0.) smoothing = exp(-0.5 * k_squared * Rsmooth * Rsmooth);
            {
              const double sre = row[2 * idz] * smoothing, sim = row[2 * idz + 1] * smoothing;
              pwr[r] += wgt * (sre * sre + sim * sim);
            }
            for (int k = 0; k < ncomp; k++) {
              double re = row[2 * idz], im = row[2 * idz + 1];
              if (k_squared != 0.) {  /* the k = 0 mode is left untouched (:368) */
                const double growth_rate = 1.0;
                re *= green[k] * smoothing * growth_rate;
                im *= green[k] * smoothing * growth_rate;
              }
              fre[((size_t)r * ncomp + k) * nzh + idz] = re;
              fim[((size_t)r * ncomp + k) * nzh + idz] = im;
            }
          }
        }
        for (int rk = 0; rk < nrad * ncomp; rk++) {
          const double *restrict xr = fre + (size_t)rk * nzh, *restrict xi = fim + (size_t)rk * nzh;
          for (int p = 0; p < nplanes; p++) {
            const double er = a->E[2 * ((size_t)p * n + idx)], ei = a->E[2 * ((size_t)p * n + idx) + 1];
            double *restrict g = a->G + 2 * (((size_t)rk * nplanes + p) * plane_c + (size_t)idy * nzh);
            for (int idz = 0; idz < nzh; idz++) {
              g[2 * idz] += xr[idz] * er - xi[idz] * ei;
              g[2 * idz + 1] += xr[idz] * ei + xi[idz] * er;
            }
          }
        }
      }
      for (int r = 0; r < nrad; r++) a->pw[(size_t)idy * nrad + r] = pwr[r];
    }
    free(fre); free(fim); free(pwr);
  }
  a->next_kx = kx0 + nkx;
  return 0;
}
/* every row added: the planes of radius irad, out [ncomp][nplanes][n][n].  (Transforms that radius' accumulators in place: once per radius.) */
int orc_plane_acc_finish(orc_plane_acc *a, int irad, double *out) {
  if (a->next_kx != a->c->n || irad < 0 || irad >= a->nrad) return 1;
  plane_finish(a->c, a->G + 2 * ((size_t)irad * a->ncomp * a->nplanes * a->c->n * a->c->nzh), a->ncomp * a->nplanes, out);
  return 0;
}
/* Parseval's side of the rows added so far: sum over their modes of |spec|^2 smoothing^2 (both halves of the spectrum counted) -- over
   all rows, n^6 times the variance of the smoothed density (the k = 0 mode counts unsmoothed, as compute_derivative leaves it) */
double orc_plane_acc_power(orc_plane_acc *a, int irad) {
  double s = 0.0;
  if (irad < 0 || irad >= a->nrad) return -1.0;
  for (int idy = 0; idy < a->c->n; idy++) s += a->pw[(size_t)idy * a->nrad + irad];
  return s;
}
void orc_plane_acc_destroy(orc_plane_acc *a) {
  if (!a) return;
  free(a->G); free(a->pw); free(a->E); free(a->xs); free(a);
}

/* compute_collapse_times (src/collapse_times.c:431-673) on a list of cells: d6 = [6][ncells] in the storage order
   11,22,33,12,13,23; fmax / rmax are the cells' products.Fmax (float, compared after promotion: quirk Q2) and Rmax,
   initialised here at ismooth 0 (:461-492) and updated by the running maximum. */
int orc_plane_collapse_times(orc_ctx *c, int ismooth, size_t ncells, const double *d6, ORC_PRODFLOAT *fmax, int *rmax) {
  double *dx = c->sx, *dy = c->sy, *dc = c->sc;
  const int dn = c->nk;
  const int per_radius = ismooth >= 0 && ismooth < ORC_MAX_SMOOTH && c->rnk[ismooth];
  if (per_radius) { c->sx = c->rsx[ismooth]; c->sy = c->rsy[ismooth]; c->sc = c->rsc[ismooth]; c->nk = c->rnk[ismooth]; }
  c->cur_ismooth = ismooth;
  int all_fails = 0;
#pragma omp parallel for num_threads(c->nthreads) schedule(static) reduction(+ : all_fails)
  for (size_t index = 0; index < ncells; index++) {
    if (!ismooth) { fmax[index] = -10.0; rmax[index] = -1; }
    double diff_ten[6];
    for (int i = 0; i < 6; i++) diff_ten[i] = d6[(size_t)i * ncells + index];
    double lambda1, lambda2, lambda3;
    int fail;
    const double Fnew = orc_inverse_collapse_time(c, diff_ten, &lambda1, &lambda2, &lambda3, &fail);
    if (fail) all_fails += 1;
    if (fmax[index] < Fnew) {
      fmax[index] = Fnew;
      rmax[index] = ismooth;
    }
  }
  if (per_radius) { c->sx = dx; c->sy = dy; c->sc = dc; c->nk = dn; }
  return all_fails ? 1 : 0;
}
