/*
 * pf_compat.c -- host side of the drop-in: the reference's own entry points
 * (same names, arguments, return convention and log lines) implemented over
 * the C ABI of libpinfmax_hip.so.  Linked INSTEAD of
 *     fmax.o  collapse_times.o  fmax-pfft.o  LPT.o        (src/Makefile:222-224)
 * it serves every caller of the path: main (src/pinocchio.c:146-150,186,229),
 * initialization (src/initialization.c:139,512) and fragment (src/fragment.c:409).
 *
 *   reference symbol                         file:line            here
 *   int  set_one_grid(int)                   fmax-pfft.c:80       geometry of this rank's x-slab
 *   int  compute_fft_plans(void)             fmax-pfft.c:139      pf_create (+ RCCL exchange with MPI)
 *   int  finalize_fft(void)                  fmax-pfft.c:231      pf_destroy
 *   int  compute_fmax(void)                  fmax.c:36            upload, pf_sweep, displacements, download, PDF
 *   int  compute_displacements(int,int,dbl)  fmax.c:292           pf_set_growth + pf_displacements
 *   int  compute_collapse_times(int)         collapse_times.c:431 pf_collapse_times
 *   int  compute_LPT_displacements(int,dbl)  LPT.c:32             (inside pf_displacements)
 *   int  Fmax_PDF(void)                      fmax.c:509           pf_fmax_pdf + the same ASCII file
 *   int  dump_products(void), read_dumps()   fmax.c:372,429       same files, same checks
 *   char *fdate(void)                        fmax.c:261           same string
 *   (optional) int pf_compat_genic(double)   GenIC.c:73           GenIC_large on the device instead of the host
 *
 * The FFT-module seam (src/pinocchio.h:551-562), which the density writer (src/pinocchio.c:146-150) and
 * ReadWhiteNoise.c:161-222 also call, works on the reference's host arrays cvector_fft / rvector_fft:
 *   double forward_transform(int), reverse_transform(int)   fmax-pfft.c:191-228    pf_forward/reverse_transform
 *   int  compute_derivative(int,int,int)       fmax-pfft.c:255      pf_derivative
 *   void write_in/from_cvector, _rvector       fmax-pfft.c:459-560  flat host copies (unchanged semantics)
 *   void write_from_rvector_to_products        fmax-pfft.c:563-631  host loop
 *   int  compute_first_derivatives(double,int,int,double*)  fmax.c:193   the three above, per axis
 *   int  compute_second_derivatives(double,int)             fmax.c:225   pf_second_derivatives + 6 downloads
 * Each transform makes a host round trip: a bring-up / tools seam, not what compute_fmax uses.
 *
 * With -DTABULATED_CT (stand-alone build: pf_compat_tabulated_ct != 0) also
 *   int initialize_collapse_times(int ismooth, int onlycompute)   collapse_times.c:820   pf_ct_build / pf_ct_load + the
 *                                                                                        CTtable file, same format
 *   int reset_collapse_times(int)                                 collapse_times.c:1046  nothing to reset
 *
 * Build modes: stand-alone (default; globals from pf_compat_types.h, defined in
 * pf_compat_globals.c) or -DPF_IN_PINOCCHIO_TREE inside the reference source
 * tree (globals and cosmology from the reference itself, MPI for the
 * ncclUniqueId broadcast).  Host code is plain C99, as in the reference.
 */
#ifdef PF_IN_PINOCCHIO_TREE
#include "pinocchio.h"
#define PF_KNOTS_X (SPLINE[SP_INVGROW]->x)
#define PF_KNOTS_Y (SPLINE[SP_INVGROW]->y)
#define PF_KNOTS_N ((int)SPLINE[SP_INVGROW]->size)
#define PF_GM(z) GrowingMode((z), params.k_for_GM)
#define PF_GM_K(z, k) GrowingMode((z), (k))
#define PF_GM2(z) GrowingMode_2LPT((z), params.k_for_GM)
#define PF_GM31(z) GrowingMode_3LPT_1((z), params.k_for_GM)
#define PF_GM32(z) GrowingMode_3LPT_2((z), params.k_for_GM)
#else
#include "pf_compat_types.h"
#define PF_KNOTS_X (pf_invgrow_knots.x)
#define PF_KNOTS_Y (pf_invgrow_knots.y)
#define PF_KNOTS_N ((int)pf_invgrow_knots.size)
#define PF_GM(z) pf_GrowingMode((z), 0.0)
#define PF_GM_K(z, k) pf_GrowingMode((z), (k))
#define PF_GM2(z) pf_GrowingMode_2LPT((z), 0.0)
#define PF_GM31(z) pf_GrowingMode_3LPT_1((z), 0.0)
#define PF_GM32(z) pf_GrowingMode_3LPT_2((z), 0.0)
#endif

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "../../include/pinfmax.h"

#if defined(PF_IN_PINOCCHIO_TREE)
#ifdef TABULATED_CT
#define PF_TABULATED 1
#else
#define PF_TABULATED 0
#endif
#else
#define PF_TABULATED pf_compat_tabulated_ct
#endif
#if defined(PF_IN_PINOCCHIO_TREE)
#ifdef ELL_SNG
#define PF_ELL_SNG 1
#else
#define PF_ELL_SNG 0
#endif
#define PF_HUBBLE(z) Hubble(z)
#else
#define PF_ELL_SNG pf_compat_ell_sng
#define PF_HUBBLE(z) pf_Hubble(z)
#endif
/* -DMOD_GRAV_FR -DFR0=...: the f(R) force modification inside the ELL_SNG system (src/collapse_times.c:295-312) */
#if defined(PF_IN_PINOCCHIO_TREE)
#ifdef MOD_GRAV_FR
#define PF_FR0 ((double)FR0)
#else
#define PF_FR0 0.0
#endif
#define PF_H_OVER_C H_over_c
#else
#define PF_FR0 pf_compat_fr0
#define PF_H_OVER_C (100. / 299792.458) /* src/cosmo.c:109 with SPEEDOFLIGHT in km/s */
#endif
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(TABULATED_CT)
#define PF_HAVE_CT 1
int initialize_collapse_times(int ismooth, int onlycompute);
int reset_collapse_times(int ismooth);
#else
#define initialize_collapse_times(a, b) 0
#define reset_collapse_times(a) 0
#endif

/* order of the displacements: the reference's -DTWO_LPT / -DTHREE_LPT (src/pinocchio.h:158-160: THREE_LPT implies TWO_LPT);
   the stand-alone build has the full record and takes the order from pf_compat_lpt_order */
#if defined(PF_IN_PINOCCHIO_TREE) && defined(THREE_LPT)
#define PF_LPT_ORDER 3
#elif defined(PF_IN_PINOCCHIO_TREE) && defined(TWO_LPT)
#define PF_LPT_ORDER 2
#elif defined(PF_IN_PINOCCHIO_TREE)
#define PF_LPT_ORDER 1
#else
#define PF_LPT_ORDER pf_compat_lpt_order
#endif

/* table interpolation of a TABULATED_CT build: -DTRILINEAR or -DALL_SPLINE in OPTIONS (tests/Readme_Pinocchio_tests_V5_1.txt)
   take precedence over the BILINEAR_SPLINE the source defines (src/collapse_times.c:39-41, 1153-1231) */
#if defined(PF_IN_PINOCCHIO_TREE) && defined(ALL_SPLINE)
#define PF_CT_FLAVOUR 2
#elif defined(PF_IN_PINOCCHIO_TREE) && defined(TRILINEAR)
#define PF_CT_FLAVOUR 1
#elif defined(PF_IN_PINOCCHIO_TREE)
#define PF_CT_FLAVOUR 0
#else
#define PF_CT_FLAVOUR pf_compat_ct_interpolation
#endif

static pf_ctx *pf_context = NULL;
static int pf_density_on_device = 0; /* set by pf_compat_genic: kdensity[0] is not uploaded */
static int pf_inputs_on_device = 0;  /* the context holds the density and the inverse-growth spline(s) (pf_upload_inputs) */
static int pf_device_of_rank = -1; /* -1: rank % visible devices */

/* lets a launcher pin ranks to devices before compute_fft_plans (e.g. from SLURM_LOCALID) */
void pf_compat_set_device(int device) { pf_device_of_rank = device; }
pf_ctx *pf_compat_context(void) { return pf_context; }

static double pf_wtime(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* The reference's time stamp (src/fmax.c:261-289 rearranges ctime()): "Www Mmm dd yyyy hh:mm:ss", 24 characters, day of
   the month blank padded, English names whatever the locale.  Built from the broken-down local time. */
char *fdate(void) {
  static const char day_name[7][4] = {"Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat"};
  static const char month_name[12][4] = {"Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"};
  const time_t now = time(NULL);
  struct tm t;
  char stamp[64]; /* room for any int the compiler can imagine in the fields; 24 characters are used */
  localtime_r(&now, &t);
  snprintf(stamp, sizeof(stamp), "%s %s %2d %4d %02d:%02d:%02d", day_name[t.tm_wday % 7], month_name[t.tm_mon % 12], t.tm_mday,
           1900 + t.tm_year, t.tm_hour, t.tm_min, t.tm_sec);
  memcpy(date_string, stamp, 24);
  date_string[24] = '\0';
  return date_string;
}

/* src/fmax-pfft.c:80-134: x-slab decomposition (1-D PFFT, src/initialization.c:1317-1325) */
int set_one_grid(int ThisGrid) {
  grid_data *G = &MyGrids[ThisGrid];
  const ptrdiff_t n = G->GSglobal[_x_];
  if (G->GSglobal[_y_] != n || G->GSglobal[_z_] != n) {
    printf("ERROR on task %d: the GPU path needs a cubic grid\n", ThisTask);
    return 1;
  }
  if (n % NTasks) { /* PFFT would hand out ragged slabs; the device transposes equal blocks */
    printf("ERROR on task %d: NTasks=%d must divide GridSize=%ld (slab decomposition)\n", ThisTask, NTasks, (long)n);
    return 1;
  }
#ifdef PF_IN_PINOCCHIO_TREE
  if (internal.tasks_subdivision_dim > 1) { /* src/initialization.c:1317-1325: pencils / volumes */
    printf("ERROR on task %d: the GPU path decomposes the box in slabs (one task per GPU): tasks_subdivision_dim = %d\n", ThisTask,
           internal.tasks_subdivision_dim);
    return 1;
  }
#endif
  G->norm = (double)1.0 / ((double)G->Ntotal);
  G->CellSize = (double)G->BoxSize / G->GSglobal[_x_];
  G->GSlocal[_x_] = n / NTasks; G->GSlocal[_y_] = n; G->GSlocal[_z_] = n;
  G->GSstart[_x_] = ThisTask * (n / NTasks); G->GSstart[_y_] = 0; G->GSstart[_z_] = 0;
  G->GSlocal_k[_x_] = n / NTasks; G->GSlocal_k[_y_] = n; G->GSlocal_k[_z_] = n / 2 + 1; /* non-transposed output */
  G->GSstart_k[_x_] = G->GSstart[_x_]; G->GSstart_k[_y_] = 0; G->GSstart_k[_z_] = 0;
  if (params.use_transposed_fft) { /* PFFT_TRANSPOSED_OUT on slabs (src/fmax-pfft.c:92): k-space is a ky-slab, memory order [y, x, z] (:271-281) */
    G->GSlocal_k[_x_] = n; G->GSlocal_k[_y_] = n / NTasks;
    G->GSstart_k[_x_] = 0; G->GSstart_k[_y_] = ThisTask * (n / NTasks);
  }
  G->total_local_size_fft = (unsigned int)(2 * (n / NTasks) * n * (n / 2 + 1));
  G->total_local_size = (unsigned int)(G->GSlocal[_x_] * n * n);
  G->off = 0;
  return 0;
}

/* src/fmax-pfft.c:139-188: the "plans" are the device context */
int compute_fft_plans(void) {
  pf_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n = MyGrids[0].GSglobal[_x_];
  cfg.rank = ThisTask;
  cfg.nranks = NTasks;
  cfg.device = pf_device_of_rank >= 0 ? pf_device_of_rank : 0;
  cfg.field_bytes = 8;
  cfg.flags = PF_FLAG_TIMING; /* cputime.fft like the reference (src/fmax-pfft.c:195-199) */
#if defined(PF_IN_PINOCCHIO_TREE) && defined(DOUBLE_PRECISION_PRODUCTS)
  cfg.flags |= PF_FLAG_DOUBLE_PRODUCTS; /* PRODFLOAT double (src/pinocchio.h:219-225): Fmax and Vel* stay fp64 on the device too */
#endif
  if (pf_context) return 0;
  if (pf_create(&pf_context, &cfg)) return 1;
  if (pf_set_lpt_order(pf_context, PF_LPT_ORDER)) return 1;
  if (pf_set_ct_interpolation(pf_context, PF_CT_FLAVOUR)) return 1;
  if (pf_set_transposed_spectra(pf_context, params.use_transposed_fft)) return 1;
#if defined(PF_IN_PINOCCHIO_TREE)
  if (NTasks > 1) { /* one rank per GPU: RCCL all-to-all replaces the MPI_Alltoall inside pfft_execute */
    char id[128];
    if (!ThisTask && pf_rccl_unique_id(id)) return 1;
    MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
    if (pf_init_rccl(pf_context, id)) return 1;
    if (pf_rccl_comm_count(pf_context) != NTasks) { /* the communicator RCCL built must be the one MPI started */
      printf("ERROR on task %d: the RCCL communicator has %d ranks, MPI has %d tasks\n", ThisTask, pf_rccl_comm_count(pf_context), NTasks);
      return 1;
    }
  }
#else
  if (NTasks > 1) {
    printf("ERROR on task %d: stand-alone pf_compat build is single rank (use pf_fabric_attach or the in-tree build)\n", ThisTask);
    return 1;
  }
#endif
  return 0;
}

/* src/fmax-pfft.c:231-252 */
int finalize_fft(void) {
#ifndef RECOMPUTE_DISPLACEMENTS
  if (pf_context) {
    pf_destroy(pf_context);
    pf_context = NULL;
    pf_density_on_device = 0;
    pf_inputs_on_device = 0;
  }
#endif
  return 0;
}

/* Optional replacement of GenIC_large(0) (src/GenIC.c:73, called from src/initialization.c:461): the same
   realisation (spiral seed table, ranlxd1 streams, E&H spectrum) generated in HBM; kdensity[0] is then neither
   filled nor uploaded.  PkNorm: the constant of src/cosmo.c:1058-1072 (static there; the value it prints), or
   <= 0 to have it integrated here from params.Sigma8. */
int pf_compat_genic(double PkNorm) {
  pf_genic_params g;
  if (!pf_context && compute_fft_plans()) return 1;
  memset(&g, 0, sizeof(g));
  g.Omega0 = params.Omega0; g.OmegaBaryon = params.OmegaBaryon; g.Hubble100 = params.Hubble100;
  g.PrimordialIndex = params.PrimordialIndex; g.BoxSize_true_Mpc = params.BoxSize_htrue;
  g.RandomSeed = (unsigned int)params.RandomSeed;
  g.FixedIC = params.FixedIC; g.PairedIC = params.PairedIC; /* src/GenIC.c:371-376 */
  g.PkNorm = PkNorm;
#ifdef PF_IN_PINOCCHIO_TREE
  /* which spectrum (initialize_PowerSpectrum, src/cosmo.c:1009-1046): "no" / "EH" = Eisenstein & Hu; "Efstathiou" and "PowerLaw" the two
     other analytic forms; a file name or CAMBTable = the table behind SPLINE[SP_PK] (PowerSpec_Tabulated :1432-1435); the
     warm-dark-matter cut-off of PowerSpectrum() (:987-1005) multiplies any of them */
  g.WDM_PartMass_in_kev = params.WDM_PartMass_in_kev;
  g.UnitLength_in_cm = 3.085678e24; /* UnitLength_in_cm, a constant private to src/cosmo.c:41 */
  if (!strcmp(params.FileWithInputSpectrum, "Efstathiou")) g.spectrum = 3;
  else if (!strcmp(params.FileWithInputSpectrum, "PowerLaw")) g.spectrum = 4;
  else if (strcmp(params.FileWithInputSpectrum, "no") && strcmp(params.FileWithInputSpectrum, "EH")) {
    g.pk_n = (int)SPLINE[SP_PK]->size; g.pk_logk = SPLINE[SP_PK]->x; g.pk_logk3p = SPLINE[SP_PK]->y;
    if (PkNorm <= 0.0) { /* normalize_PowerSpectrum (:1061-1081): a trusted table (Sigma8 0, or CAMBTable) has PkNorm 1 */
      if (params.Sigma8 != 0.0 && strcmp(params.FileWithInputSpectrum, "CAMBTable")) {
        printf("ERROR on task %d: a tabulated spectrum renormalised to Sigma8 needs the PkNorm the reference printed\n", ThisTask);
        return 1;
      }
      g.PkNorm = 1.0;
    }
  }
#endif
  if (g.PkNorm <= 0.0 && pf_pk_norm(&g, params.Sigma8, &g.PkNorm)) return 1;
  if (!ThisTask) printf("[%s] Generating the linear density field on the device, PkNorm=%g\n", fdate(), g.PkNorm);
  if (pf_genic_density(pf_context, &g)) return 1;
  pf_density_on_device = 1;
  return 0;
}

static int pf_upload_inputs(void) {
  if (!pf_context && compute_fft_plans()) return 1;
  if (!pf_density_on_device && pf_set_density(pf_context, kdensity[0])) return 1;
#if defined(PF_IN_PINOCCHIO_TREE) && defined(SCALE_DEPENDENT)
  { /* one inverse-growth spline per smoothing radius, src/initialization.c:1704-1708, src/cosmo.c:1828 */
    int ismooth;
    for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++)
      if (pf_set_invgrow(pf_context, ismooth, SPLINE_INVGROW[ismooth]->x, SPLINE_INVGROW[ismooth]->y, (int)SPLINE_INVGROW[ismooth]->size)) return 1;
  }
#elif defined(PF_IN_PINOCCHIO_TREE)
  if (pf_set_invgrow(pf_context, -1, PF_KNOTS_X, PF_KNOTS_Y, PF_KNOTS_N)) return 1;
#else
  if (pf_compat_scale_dependent) {
    int ismooth;
    for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++)
      if (pf_set_invgrow(pf_context, ismooth, pf_invgrow_knots_radius[ismooth].x, pf_invgrow_knots_radius[ismooth].y,
                         (int)pf_invgrow_knots_radius[ismooth].size)) return 1;
  } else if (pf_set_invgrow(pf_context, -1, PF_KNOTS_X, PF_KNOTS_Y, PF_KNOTS_N)) return 1;
#endif
  pf_inputs_on_device = 1;
  return 0;
}

/* -DELL_SNG: the model of the collapse-time table (src/collapse_times.c:222-400).  OmegaRad and OmegaK are static in
   src/cosmo.c; they are recovered from the public Hubble(z): E^2(z) - Omega0 (1+z)^3 - OmegaLambda =
   OmegaRad (1+z)^4 + OmegaK (1+z)^2 at two redshifts (params.simpleLambda). */
static int pf_upload_collapse_model(void) {
  double cosmo[4], D_in[64], h0, r1, r3;
  int ismooth;
  if (!PF_ELL_SNG) return pf_set_collapse_model(pf_context, 0, NULL, 0, NULL);
  h0 = PF_HUBBLE(0.0);
  r1 = pow(PF_HUBBLE(1.0) / h0, 2.) * 1.0 - params.Omega0 * 8. - params.OmegaLambda;   /* = 16 Or + 4 Ok, E^2 normalised to E(0) = 1 */
  r3 = pow(PF_HUBBLE(3.0) / h0, 2.) * 1.0 - params.Omega0 * 64. - params.OmegaLambda;  /* = 256 Or + 16 Ok */
  cosmo[0] = params.Omega0; cosmo[1] = params.OmegaLambda;
  cosmo[2] = (r3 - 4. * r1) / 192.;
  cosmo[3] = (r1 - 16. * cosmo[2]) / 4.;
  if (fabs(cosmo[2]) < 1e-12) cosmo[2] = 0.0;
  if (fabs(cosmo[3]) < 1e-12) cosmo[3] = 0.0;
  for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++) /* GrowingMode(1/amin - 1, 1/Radius), :353-361 */
    D_in[ismooth] = PF_GM_K(1. / 1.e-5 - 1., 1. / Smoothing.Radius[ismooth]);
  if (PF_FR0 != 0.0) { /* ode_param: the radius itself, the previous one for the last radius (:378-388) */
    double size[64];
    for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++)
      size[ismooth] = Smoothing.Radius[ismooth < Smoothing.Nsmooth - 1 ? ismooth : (ismooth > 0 ? ismooth - 1 : 0)];
    if (pf_set_modified_gravity(pf_context, PF_FR0, PF_H_OVER_C, Smoothing.Nsmooth, size)) return 1;
  } else if (pf_set_modified_gravity(pf_context, 0.0, 0.0, 0, NULL)) return 1;
  return pf_set_collapse_model(pf_context, 1, cosmo, Smoothing.Nsmooth, D_in);
}

/* growth multipliers of compute_derivative for ScaleDep.order = 1..4 (src/fmax-pfft.c:344-364) at `redshift`:
   scalars, or in a SCALE_DEPENDENT build the NkBINS-entry tables InterpolateGrowth interpolates in
   (src/cosmo.c:1728-1755), each entry one spline evaluation on the host */
#define PF_NKBINS 10
#define PF_LOGKMIN (-3.0)
#define PF_DELTALOGK 0.5
static int pf_upload_growth(double redshift) {
  double g[4];
#if defined(PF_IN_PINOCCHIO_TREE) && defined(SCALE_DEPENDENT)
  static const int first[4] = {SP_GROW1, SP_GROW2, SP_GROW31, SP_GROW32};
  int o, j;
  for (o = 0; o < 4; o++) {
    double T[NkBINS];
    for (j = 0; j < NkBINS; j++) T[j] = my_spline_eval(SPLINE[first[o] + j], -log10(1. + redshift), ACCEL[first[o] + j]);
    if (pf_set_growth_table(pf_context, o + 1, T, NkBINS, LOGKMIN, DELTALOGK, o == 2 ? -1.0 : 1.0)) return 1;
  }
#elif !defined(PF_IN_PINOCCHIO_TREE)
  if (pf_compat_scale_dependent) {
    double (*fn[4])(double, double);
    int o, j;
    fn[0] = pf_GrowingMode; fn[1] = pf_GrowingMode_2LPT; fn[2] = pf_GrowingMode_3LPT_1; fn[3] = pf_GrowingMode_3LPT_2;
    for (o = 0; o < 4; o++) {
      double T[PF_NKBINS], sign = fn[o](redshift, 1.0) < 0.0 ? -1.0 : 1.0;
      for (j = 0; j < PF_NKBINS; j++) T[j] = log10(sign * fn[o](redshift, pow(10., PF_LOGKMIN + j * PF_DELTALOGK)));
      if (pf_set_growth_table(pf_context, o + 1, T, PF_NKBINS, PF_LOGKMIN, PF_DELTALOGK, sign)) return 1;
    }
  }
#endif
  g[0] = PF_GM(redshift); g[1] = PF_GM2(redshift); g[2] = PF_GM31(redshift); g[3] = PF_GM32(redshift);
  return pf_set_growth(pf_context, g);
}

static void pf_collect_cputime(void) {
  pf_cputime t;
  if (!pf_get_cputime(pf_context, &t)) {
    cputime.fft += t.fft; cputime.coll += t.coll; cputime.lpt += t.lpt; cputime.deriv += t.deriv;
    cputime.mem_transf += t.mem_transf;
    pf_reset_cputime(pf_context);
  }
}

static int pf_inside_compute_fmax = 0;

/* the displacement fields this build's product_data has (src/pinocchio.h:237-243) */
static void pf_velocity_offsets(pf_product_layout *lay) {
  lay->off_Vel = (int)offsetof(product_data, Vel);
  lay->off_Vel_2LPT = lay->off_Vel_3LPT_1 = lay->off_Vel_3LPT_2 = -1;
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(TWO_LPT) || defined(THREE_LPT)
  if (PF_LPT_ORDER >= 2) lay->off_Vel_2LPT = (int)offsetof(product_data, Vel_2LPT);
#endif
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(THREE_LPT)
  if (PF_LPT_ORDER >= 3) {
    lay->off_Vel_3LPT_1 = (int)offsetof(product_data, Vel_3LPT_1);
    lay->off_Vel_3LPT_2 = (int)offsetof(product_data, Vel_3LPT_2);
  }
#endif
}

/* re-entrant compute_displacements (src/fragment.c:398-410): only the Vel* fields of the host records change */
static int pf_download_velocities(void) {
  pf_product_layout lay;
  lay.stride = sizeof(product_data);
  lay.off_Rmax = -1;
  lay.off_Fmax = -1;
  pf_velocity_offsets(&lay);
  return pf_update_products(pf_context, products, &lay);
}

static int pf_download_products(void) {
  pf_product_layout lay;
  lay.stride = sizeof(product_data);
  lay.off_Rmax = (int)offsetof(product_data, Rmax);
  lay.off_Fmax = (int)offsetof(product_data, Fmax);
  pf_velocity_offsets(&lay);
  return pf_get_products(pf_context, products, &lay);
}

int Fmax_PDF(void);

/* src/fmax.c:292-367.  recompute_sd: second derivatives at R=0 are recomputed on the device. */
int compute_displacements(int compute_sources, int recompute_sd, double redshift) {
  double cputmp = pf_wtime();
  if (!ThisTask) printf("\n[%s] Computing LPT displacements\n", fdate());
  /* "pinocchio.x parameterfile 3" (src/pinocchio.c:172-187) comes here straight from the initialisation: plans made, nothing
     uploaded yet */
  if ((!pf_context || !pf_inputs_on_device) && pf_upload_inputs()) return 1;
  ScaleDep.redshift = redshift;
  if (pf_upload_growth(redshift)) return 1;
  if (pf_displacements(pf_context, compute_sources, recompute_sd)) return 1;
  if (!pf_inside_compute_fmax) { /* called from fragment: the host records take the new displacements now */
    if (pf_download_velocities()) return 1;
    pf_collect_cputime();
  }
  ScaleDep.order = 1;
  cputmp = pf_wtime() - cputmp;
  if (!ThisTask) printf("[%s] Done LPT displacements and first derivatives, cpu time = %f s\n", fdate(), cputmp);
  return 0;
}

/* src/LPT.c:32 -- kept for callers that link it directly; the work is inside pf_displacements */
int compute_LPT_displacements(int compute_sources, double redshift) { return compute_displacements(compute_sources, 0, redshift); }

/* src/collapse_times.c:431: one radius, on the second derivatives resident on the device */
int compute_collapse_times(int ismooth) {
  double tv = 0.0;
  if (pf_collapse_times(pf_context, ismooth, &tv)) return 1;
  Smoothing.TrueVariance[ismooth] = tv;
  return 0;
}

/* src/fmax.c:36-190 */
int compute_fmax(void) {
  int ismooth;
  double *rs;
  cputime.fmax = pf_wtime();
  if (!ThisTask) printf("[%s] First part: computation of collapse times\n", fdate());
  ScaleDep.order = 0;
  ScaleDep.redshift = 0.0;
  if (pf_upload_inputs()) return 1;

  /* CYCLE ON SMOOTHING RADII: radii in grid units, Rsmooth = R / CellSize (src/fmax.c:233) */
  rs = (double *)malloc(sizeof(double) * Smoothing.Nsmooth);
  for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++) rs[ismooth] = Smoothing.Radius[ismooth] / MyGrids[0].CellSize;
  Rsmooth = rs[Smoothing.Nsmooth - 1];
  {
    double cpusm = pf_wtime();
    if (PF_TABULATED) { /* src/fmax.c:66-150 radius by radius: derivatives, table of this radius, collapse times */
      for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++) {
        if (pf_second_derivatives(pf_context, rs[ismooth]) || initialize_collapse_times(ismooth, 0) ||
            compute_collapse_times(ismooth) || reset_collapse_times(ismooth)) { free(rs); return 1; }
      }
    } else { /* compute_displacements(1, 0, z) follows at once (src/fmax.c:150-163): the last radius leaves the LPT sources */
      int rc = pf_set_sources_in_sweep(pf_context, PF_LPT_ORDER >= 2) || pf_sweep(pf_context, Smoothing.Nsmooth, rs, Smoothing.TrueVariance);
      pf_set_sources_in_sweep(pf_context, 0);
      if (rc) { free(rs); return 1; }
    }
    free(rs);
    /* the radii run back to back on the device: the log keeps the reference's two lines per radius (src/fmax.c:66-69,
       141-145), with the sweep's wall time shared evenly */
    cpusm = (pf_wtime() - cpusm) / Smoothing.Nsmooth;
    if (!ThisTask)
      for (ismooth = 0; ismooth < Smoothing.Nsmooth; ismooth++) {
        const double sig = Smoothing.Variance ? sqrt(Smoothing.Variance[ismooth]) : 0.0;
        printf("\n[%s] Starting smoothing radius %d of %d (R=%9.5f, sigma=%9.5f)\n", fdate(), ismooth + 1, Smoothing.Nsmooth,
               Smoothing.Radius[ismooth], sig);
        printf("[%s] Completed, R=%6.3f, expected sigma: %7.4f, computed sigma: %7.4f, cpu time = %f s\n", fdate(),
               Smoothing.Radius[ismooth], sig, sqrt(Smoothing.TrueVariance[ismooth]), cpusm);
      }
  }

  /* COMPUTATION OF DISPLACEMENTS for the first (or only) redshift segment (src/fmax.c:160-169) */
  if (!ThisTask) printf("\n[%s] Computing displacements  for redshift %f\n", fdate(), ScaleDep.z[0]);
  pf_inside_compute_fmax = 1;
  if (compute_displacements(1, 0, ScaleDep.z[0])) { pf_inside_compute_fmax = 0; return 1; }
  pf_inside_compute_fmax = 0;

  if (pf_download_products()) return 1;
  pf_collect_cputime();
  if (Fmax_PDF()) return 1; /* needs the device histogram: before finalize_fft */
  if (finalize_fft()) return 1;

  cputime.fmax = pf_wtime() - cputime.fmax;
  if (!ThisTask)
    printf("[%s] Finishing fmax, total fmax cpu time = %14.6f\n"
           "\t\t IO       : %14.6f (%14.6f total time without I/O)\n"
           "\t\t FFT      : %14.6f\n"
           "\t\t COLLAPSE : %14.6f\n",
           fdate(), cputime.fmax, cputime.io, cputime.fmax - cputime.io, cputime.fft, cputime.coll);
  return 0;
}

/* bin of the Fmax histogram: tenths of F, everything below 0 in the first and everything from NBINS/10 on in the last bin
   (the definition behind pinocchio.*.FmaxPDF.out, src/fmax.c:517-525) */
static int pf_pdf_bin(PRODFLOAT F) { /* the record's own type: a -DDOUBLE_PRECISION_PRODUCTS build bins the double */
  const int b = (int)(F * 10.);
  return b < 0 ? 0 : b >= NBINS ? NBINS - 1 : b;
}

/* Fmax_PDF (src/fmax.c:509-550): the histogram comes from the device (already summed over the ranks) or, when the products
   only exist on the host (after read_dumps), from this rank's records; rank 0 writes the reference's ASCII file */
int Fmax_PDF(void) {
  unsigned long long counter[NBINS];
  if (pf_context) {
    if (pf_fmax_pdf(pf_context, counter)) return 1;
  } else {
    const product_data *p = products, *end = products + MyGrids[0].total_local_size;
    memset(counter, 0, sizeof(counter));
    for (; p != end; p++) counter[pf_pdf_bin(p->Fmax)]++;
#ifdef PF_IN_PINOCCHIO_TREE
    {
      unsigned long long mine[NBINS];
      memcpy(mine, counter, sizeof(mine));
      MPI_Reduce(mine, counter, NBINS, MPI_UNSIGNED_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    }
#endif
  }
  if (!ThisTask) {
    char filename[LBLENGTH];
    unsigned long long collapsed = 0;
    FILE *file;
    int b;
    for (b = 10; b < NBINS; b++) collapsed += counter[b]; /* F >= 1: collapsed by z = 0 */
    printf("[%s] Number of collapsed particles to z=0: %llu\n", fdate(), collapsed);
    sprintf(filename, "pinocchio.%s.FmaxPDF.out", params.RunFlag);
    file = fopen(filename, "w");
    if (!file) { printf("ERROR on task %d: could not open file %s\n", ThisTask, filename); return 1; }
    fprintf(file, "# Fmax PDF over %llu particles\n# 1-2: F interval\n# 3: number of particles in that interval\n#\n", MyGrids[0].Ntotal);
    for (b = 0; b < NBINS; b++)
      fprintf(file, " %6.1f   %6.1f  %llu\n", (double)b / 10., b + 1 < NBINS ? (double)(b + 1) / 10. : 999.0, counter[b]);
    fclose(file);
  }
  return 0;
}

/* ---- dump / restore of the products (src/fmax.c:372-506): DumpDir/summary (four "%d   # label" lines), DumpDir/TrueVariance
   (Nsmooth doubles), DumpDir/Task.<rank> (this rank's product_data records); formats as in the reference ---- */
struct pf_summary_line { const char *label; int value; };
static int pf_summary_fill(struct pf_summary_line line[4]) {
  line[0].label = "NTasks";                  line[0].value = NTasks;
  line[1].label = "random seed";             line[1].value = params.RandomSeed;
  line[2].label = "grid size";               line[2].value = params.GridSize[0];
  line[3].label = "length of product_data";  line[3].value = (int)sizeof(product_data);
  return 4;
}
/* what a mismatch of line i is called in the reference's message */
static const char *const pf_summary_what[4] = {"number of tasks", "random seed", "grid size", "length of product_data"};

/* one binary file of the dump directory, written or read whole; 0 on success */
static int pf_dump_blob(const char *name, void *data, size_t size, size_t count, int writing, int task_in_message) {
  char fname[LBLENGTH];
  FILE *file;
  size_t done;
  snprintf(fname, sizeof(fname), "%s%s", params.DumpDir, name);
  file = fopen(fname, writing ? "wb" : "rb");
  if (!file) {
    if (task_in_message) printf("ERROR on Task %d: could not open file %s\n", ThisTask, fname);
    else printf("ERROR on Task 0: could not open file %s\n", fname);
    return 1;
  }
  done = writing ? fwrite(data, size, count, file) : fread(data, size, count, file);
  fclose(file);
  return done != count;
}

int dump_products(void) {
  char name[64];
  if (!ThisTask) {
    struct pf_summary_line line[4];
    struct stat dr;
    char fname[LBLENGTH];
    FILE *file;
    int i;
    const int nl = pf_summary_fill(line);
    if (stat(params.DumpDir, &dr)) {
      printf("Creating directory %s\n", params.DumpDir);
      if (mkdir(params.DumpDir, 0755)) { printf("ERROR IN CREATING DIRECTORY %s (task 0)\n", params.DumpDir); return 1; }
    }
    snprintf(fname, sizeof(fname), "%ssummary", params.DumpDir);
    file = fopen(fname, "w");
    if (!file) { printf("ERROR on Task 0: could not open file %s\n", fname); return 1; }
    for (i = 0; i < nl; i++) fprintf(file, "%d   # %s\n", line[i].value, line[i].label);
    fclose(file);
    if (pf_dump_blob("TrueVariance", Smoothing.TrueVariance, sizeof(double), Smoothing.Nsmooth, 1, 0)) return 1;
  }
#ifdef PF_IN_PINOCCHIO_TREE
  MPI_Barrier(MPI_COMM_WORLD); /* the directory exists before the other ranks write into it */
#endif
  snprintf(name, sizeof(name), "Task.%d", ThisTask);
  return pf_dump_blob(name, products, sizeof(product_data), MyGrids[0].total_local_size, 1, 1);
}

int read_dumps(void) {
  char name[64];
  if (!ThisTask) {
    struct pf_summary_line line[4];
    char fname[LBLENGTH], buf[SBLENGTH];
    FILE *file;
    int i, mismatches = 0;
    const int nl = pf_summary_fill(line);
    snprintf(fname, sizeof(fname), "%ssummary", params.DumpDir);
    file = fopen(fname, "r");
    if (!file) { printf("ERROR on Task 0: could not open file %s\n", fname); return 1; }
    for (i = 0; i < nl; i++) {
      int found = 0;
      if (!fgets(buf, SBLENGTH, file) || sscanf(buf, "%d", &found) != 1 || found != line[i].value) {
        /* the run that wrote the dump is named first, this one second */
        printf("ERROR: the %s in %s does not match - %d vs %d\n", pf_summary_what[i], fname, found, line[i].value);
        mismatches++;
      }
    }
    fclose(file);
    if (mismatches) return 1;
    if (pf_dump_blob("TrueVariance", Smoothing.TrueVariance, sizeof(double), Smoothing.Nsmooth, 0, 0)) return 1;
  }
#ifdef PF_IN_PINOCCHIO_TREE
  MPI_Bcast(Smoothing.TrueVariance, Smoothing.Nsmooth, MPI_DOUBLE, 0, MPI_COMM_WORLD);
#endif
  snprintf(name, sizeof(name), "Task.%d", ThisTask);
  return pf_dump_blob(name, products, sizeof(product_data), MyGrids[0].total_local_size, 0, 1);
}

/* ------------------------------------------------------------------------------------------------------------
 * The FFT-module seam on the reference's host arrays (src/fmax-pfft.c:191-228, 255-441, 459-631; src/fmax.c:193-258)
 * ------------------------------------------------------------------------------------------------------------ */
void write_in_cvector(int ThisGrid, double *vector) {
  memcpy(cvector_fft[ThisGrid], vector, sizeof(double) * MyGrids[ThisGrid].total_local_size_fft);
}
void write_from_cvector(int ThisGrid, double *vector) {
  memcpy(vector, cvector_fft[ThisGrid], sizeof(double) * MyGrids[ThisGrid].total_local_size_fft);
}
void write_in_rvector(int ThisGrid, double *vector) {
  memcpy(rvector_fft[ThisGrid], vector, sizeof(double) * MyGrids[ThisGrid].total_local_size);
}
void write_from_rvector(int ThisGrid, double *vector) {
  memcpy(vector, rvector_fft[ThisGrid], sizeof(double) * MyGrids[ThisGrid].total_local_size);
}

/* elapsed seconds, like the reference; a failure is reported on stdout and returns a negative time */
double forward_transform(int ThisGrid) {
  double t = pf_wtime();
  if ((!pf_context && compute_fft_plans()) || pf_forward_transform(pf_context, rvector_fft[ThisGrid], (double *)cvector_fft[ThisGrid])) {
    printf("ERROR on task %d: forward_transform failed: %s\n", ThisTask, pf_last_error());
    return -1.0;
  }
  return pf_wtime() - t;
}
double reverse_transform(int ThisGrid) {
  double t = pf_wtime();
  if ((!pf_context && compute_fft_plans()) || pf_reverse_transform(pf_context, (double *)cvector_fft[ThisGrid], rvector_fft[ThisGrid])) {
    printf("ERROR on task %d: reverse_transform failed: %s\n", ThisTask, pf_last_error());
    return -1.0;
  }
  return pf_wtime() - t;
}

/* cvector_fft[ThisGrid] -> rvector_fft[ThisGrid]; reads the reference's Rsmooth, ScaleDep.order, ScaleDep.redshift */
int compute_derivative(int ThisGrid, int first_derivative, int second_derivative) {
  double t = pf_wtime();
  if (!pf_context && compute_fft_plans()) return 1;
  if (ScaleDep.order >= 1 && pf_upload_growth(ScaleDep.redshift)) return 1;
  if (pf_derivative(pf_context, (double *)cvector_fft[ThisGrid], first_derivative, second_derivative, Rsmooth, ScaleDep.order,
                    rvector_fft[ThisGrid]))
    return 1;
  cputime.fft += pf_wtime() - t;
  return 0;
}

void write_from_rvector_to_products(int ThisGrid, int ia, int order) {
  unsigned int index;
  const unsigned int nloc = MyGrids[ThisGrid].total_local_size;
  const double *r = rvector_fft[ThisGrid];
  switch (order) {
    case 1: for (index = 0; index < nloc; index++) products[index].Vel[ia] = r[index]; break;
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(TWO_LPT) || defined(THREE_LPT)
    case 2: for (index = 0; index < nloc; index++) products[index].Vel_2LPT[ia] = r[index]; break;
#endif
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(THREE_LPT)
    case 3: for (index = 0; index < nloc; index++) products[index].Vel_3LPT_1[ia] = r[index]; break;
    case 4: for (index = 0; index < nloc; index++) products[index].Vel_3LPT_2[ia] = r[index]; break;
#endif
    default: break;
  }
}

/* src/fmax.c:193-222 */
int compute_first_derivatives(double R, int ThisGrid, int order, double *vector) {
  int ia;
  double timetmp = 0;
  Rsmooth = R / MyGrids[ThisGrid].CellSize;
  for (ia = 1; ia <= 3; ia++) {
    double tmp;
    if (!ThisTask) printf("[%s] Computing 1st derivative: %d\n", fdate(), ia);
    tmp = pf_wtime();
    write_in_cvector(ThisGrid, vector);
    timetmp += pf_wtime() - tmp;
    if (compute_derivative(ThisGrid, ia, 0)) return 1;
    tmp = pf_wtime();
    write_from_rvector_to_products(ThisGrid, ia - 1, order);
    timetmp += pf_wtime() - tmp;
  }
  cputime.mem_transf += timetmp;
  return 0;
}

/* src/fmax.c:225-258: the six fields land in second_derivatives[ThisGrid][0..5] on the host; the shared-pass
   transforms of the sweep are used (one upload of kdensity, six downloads) */
int compute_second_derivatives(double R, int ThisGrid) {
  int i;
  Rsmooth = R / MyGrids[ThisGrid].CellSize;
  if (pf_upload_inputs()) return 1;
  if (pf_second_derivatives(pf_context, Rsmooth)) return 1;
  for (i = 0; i < 6; i++)
    if (pf_get_second_derivative(pf_context, i, second_derivatives[ThisGrid][i])) return 1;
  return 0;
}

#ifdef PF_HAVE_CT
/* ------------------------------------------------------------------------------------------------------------
 * TABULATED_CT (src/collapse_times.c:780-1231): the table of each radius is computed on the device (or read from
 * params.CTtableFile) and the node splines are built there; the table file keeps the reference's binary format:
 *   header  int type (1 = ELL_CLASSIC), double Omega0, OmegaLambda, Hubble100, int Ncomputations, CT_NBINS_D, CT_NBINS_XY
 *   per radius  int ismooth, double CT_table[Ncomputations]
 * ------------------------------------------------------------------------------------------------------------ */
#define PF_CT_NBINS_XY 50
#define PF_CT_NBINS_D 100
#define PF_CT_NCOMP (PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY)
static FILE *CTtableFilePointer = NULL;
static double *pf_ct_host = NULL;

static int check_CTtable_header(void) { /* :1235-1296, ELL_CLASSIC */
  int fail = 0, dummy = 0;
  double fdummy = 0;
  if (fread(&dummy, sizeof(int), 1, CTtableFilePointer) != 1) return 1;
  if (!PF_ELL_SNG && dummy != 1) { printf("ERROR: CT table not constructed for ELL_CLASSIC, %d\n", dummy); fail = 1; }
  if (PF_ELL_SNG && PF_FR0 == 0.0 && dummy != 3) { printf("ERROR: CT table not constructed for ELL_SNG and standard gravity, %d\n", dummy); fail = 1; }
  if (PF_ELL_SNG && PF_FR0 != 0.0 && dummy != 4) { printf("ERROR: CT table not constructed for ELL_SNG and MOD_GRAV_FR, %d\n", dummy); fail = 1; }
  if (fread(&fdummy, sizeof(double), 1, CTtableFilePointer) != 1) return 1;
  if (fabs(fdummy - params.Omega0) > 1.e-10) { printf("ERROR: CT table constructed for the wrong Omega0, %f in place of %f\n", fdummy, params.Omega0); fail = 1; }
  if (fread(&fdummy, sizeof(double), 1, CTtableFilePointer) != 1) return 1;
  if (fabs(fdummy - params.OmegaLambda) > 1.e-10) { printf("ERROR: CT table constructed for the wrong OmegaLambda, %f in place of %f\n", fdummy, params.OmegaLambda); fail = 1; }
  if (fread(&fdummy, sizeof(double), 1, CTtableFilePointer) != 1) return 1;
  if (fabs(fdummy - params.Hubble100) > 1.e-10) { printf("ERROR: CT table constructed for the wrong Hubble100, %f in place of %f\n", fdummy, params.Hubble100); fail = 1; }
  if (fread(&dummy, sizeof(int), 1, CTtableFilePointer) != 1) return 1;
  if (dummy != PF_CT_NCOMP) { printf("ERROR: CT table has the wrong size, %d in place of %d\n", dummy, PF_CT_NCOMP); fail = 1; }
  if (fread(&dummy, sizeof(int), 1, CTtableFilePointer) != 1) return 1;
  if (dummy != PF_CT_NBINS_D) { printf("ERROR: CT table has the wrong density sampling, %d in place of %d\n", dummy, PF_CT_NBINS_D); fail = 1; }
  if (fread(&dummy, sizeof(int), 1, CTtableFilePointer) != 1) return 1;
  if (dummy != PF_CT_NBINS_XY) { printf("ERROR: CT table has the wrong x and y sampling, %d in place of %d\n", dummy, PF_CT_NBINS_XY); fail = 1; }
  return fail;
}
static void write_CTtable_header(void) { /* :1300-1340 */
  int dummy = PF_ELL_SNG ? (PF_FR0 != 0.0 ? 4 : 3) : 1;
  fwrite(&dummy, sizeof(int), 1, CTtableFilePointer);
  fwrite(&params.Omega0, sizeof(double), 1, CTtableFilePointer);
  fwrite(&params.OmegaLambda, sizeof(double), 1, CTtableFilePointer);
  fwrite(&params.Hubble100, sizeof(double), 1, CTtableFilePointer);
  dummy = PF_CT_NCOMP; fwrite(&dummy, sizeof(int), 1, CTtableFilePointer);
  dummy = PF_CT_NBINS_D; fwrite(&dummy, sizeof(int), 1, CTtableFilePointer);
  dummy = PF_CT_NBINS_XY; fwrite(&dummy, sizeof(int), 1, CTtableFilePointer);
}

/* src/collapse_times.c:820-1043 */
int initialize_collapse_times(int ismooth, int onlycompute) {
  int fail = 0, dummy;
  char fname[LBLENGTH];
  /* also the first call of "pinocchio.x parameterfile 1" (src/pinocchio.c:100-131): plans made, nothing uploaded yet */
  if ((!pf_context || !pf_inputs_on_device) && pf_upload_inputs()) return 1;
  if (!pf_ct_host) pf_ct_host = (double *)malloc(sizeof(double) * PF_CT_NCOMP);
  if (!ismooth && !ThisTask)
    printf("[%s] Grid for interpolating collapse times: CT_NBINS_D=%d, CT_NBINS_XY=%d\n", fdate(), PF_CT_NBINS_D, PF_CT_NBINS_XY);
  if (strcmp(params.CTtableFile, "none") && !onlycompute) { /* read the table of this radius from the file */
    if (!ismooth) {
      if (!ThisTask) {
        CTtableFilePointer = fopen(params.CTtableFile, "r");
        fail = CTtableFilePointer ? check_CTtable_header() : 1;
        if (!CTtableFilePointer) printf("ERROR on task %d: could not open file %s\n", ThisTask, params.CTtableFile);
      }
#ifdef PF_IN_PINOCCHIO_TREE
      MPI_Bcast(&fail, sizeof(int), MPI_BYTE, 0, MPI_COMM_WORLD);
#endif
      if (fail) return 1;
    }
    if (!ThisTask) {
      if (fread(&dummy, sizeof(int), 1, CTtableFilePointer) != 1 ||
          fread(pf_ct_host, sizeof(double), PF_CT_NCOMP, CTtableFilePointer) != (size_t)PF_CT_NCOMP) fail = 1;
      if (ismooth == Smoothing.Nsmooth - 1) { fclose(CTtableFilePointer); CTtableFilePointer = NULL; }
    }
#ifdef PF_IN_PINOCCHIO_TREE
    MPI_Bcast(&fail, sizeof(int), MPI_BYTE, 0, MPI_COMM_WORLD);
    MPI_Bcast(pf_ct_host, PF_CT_NCOMP, MPI_DOUBLE, 0, MPI_COMM_WORLD);
#endif
    if (fail) { printf("ERROR on task %d: short read of the collapse-time table\n", ThisTask); return 1; }
    return pf_ct_load(pf_context, ismooth, Smoothing.Variance[ismooth], pf_ct_host);
  }
  /* compute the table of this radius: every task computes all of it on its GPU (0.25 M evaluations) instead of the
     reference's split over tasks + MPI_Allgatherv; task 0 writes the file */
  if (!ismooth && pf_upload_collapse_model()) return 1;
  if (pf_ct_build(pf_context, ismooth, Smoothing.Variance[ismooth], pf_ct_host)) return 1;
  if (!ThisTask) {
    if (onlycompute) strcpy(fname, params.CTtableFile);
    else sprintf(fname, "pinocchio.%s.CTtable.out", params.RunFlag);
    CTtableFilePointer = fopen(fname, ismooth ? "a" : "w");
    if (!CTtableFilePointer) { printf("ERROR on task %d: could not open file %s\n", ThisTask, fname); return 1; }
    if (!ismooth) write_CTtable_header();
    fwrite(&ismooth, sizeof(int), 1, CTtableFilePointer);
    fwrite(pf_ct_host, sizeof(double), PF_CT_NCOMP, CTtableFilePointer);
    fclose(CTtableFilePointer);
    CTtableFilePointer = NULL;
  }
  return 0;
}
int reset_collapse_times(int ismooth) { (void)ismooth; return 0; } /* :1046-1108: debug statistics only */
#endif
