/*
 * pf_compat_types.h -- the subset of PINOCCHIO's global state (src/pinocchio.h,
 * src/variables.c) that the hot path reads and writes, declared with the
 * reference's own names and field names so that pf_compat.c compiles either
 * here (stand-alone, against these mirrors) or inside the reference tree
 * (-DPF_IN_PINOCCHIO_TREE, against the real "pinocchio.h").
 *
 * Only fields the path touches are mirrored; flags fixed to the HMF_Validation
 * build: -DTWO_LPT -DTHREE_LPT -DELL_CLASSIC, float products.
 */
#ifndef PF_COMPAT_TYPES_H
#define PF_COMPAT_TYPES_H

#include <stddef.h>

#define NBINS 210      /* src/pinocchio.h:65 */
#define LBLENGTH 400   /* :57 */
#define SBLENGTH 100   /* :58 */
#define MAXOUTPUTS 100 /* :62 */
#define ALIGN 32
#define _x_ 0
#define _y_ 1
#define _z_ 2

typedef float PRODFLOAT; /* :219-225, no DOUBLE_PRECISION_PRODUCTS */

typedef struct /* src/pinocchio.h:233-259 */
{
  int Rmax;
  PRODFLOAT Fmax, Vel[3];
  PRODFLOAT Vel_2LPT[3];
  PRODFLOAT Vel_3LPT_1[3], Vel_3LPT_2[3];
#ifdef RECOMPUTE_DISPLACEMENTS /* the reference's default Makefile flags: 104-byte record */
  PRODFLOAT Vel_prev[3];
  PRODFLOAT Vel_2LPT_prev[3];
  PRODFLOAT Vel_3LPT_1_prev[3], Vel_3LPT_2_prev[3];
#endif
} product_data __attribute__((aligned(ALIGN)));

typedef struct /* :284-292 */
{
  int Nsmooth;
  double *Radius, *Variance, *TrueVariance;
} smoothing_data;

typedef struct /* :295-308, without the pfft plans */
{
  unsigned int total_local_size, total_local_size_fft;
  unsigned int off, ParticlesPerTask;
  ptrdiff_t GSglobal[3];
  ptrdiff_t GSlocal[3];
  ptrdiff_t GSstart[3];
  ptrdiff_t GSlocal_k[3];
  ptrdiff_t GSstart_k[3];
  double lower_k_cutoff, upper_k_cutoff, norm, BoxSize, CellSize;
  unsigned long long Ntotal;
} grid_data;

typedef struct /* :536-542 */
{
  int nseg, myseg, no_interp, order;
  double z[MAXOUTPUTS], D[MAXOUTPUTS], D2[MAXOUTPUTS], D31[MAXOUTPUTS], D32[MAXOUTPUTS];
  double redshift;
} ScaleDep_data;

typedef struct /* :368-378, the accumulators the path fills */
{
  double fft, coll, lpt, fmax, io, deriv, mem_transf;
} cputime_data;

typedef struct /* the tags of param_data (:311-352) the path uses */
{
  char RunFlag[SBLENGTH], DumpDir[SBLENGTH];
  int GridSize[3], RandomSeed;
  double Omega0, OmegaBaryon, Hubble100, Sigma8, PrimordialIndex, BoxSize_htrue; /* read by pf_compat_genic only */
  double OmegaLambda;        /* with Omega0 and Hubble100 in the header of the collapse-time table file */
  char CTtableFile[LBLENGTH]; /* "none": compute the tables (TABULATED_CT build) */
} param_data;

/* gsl_spline as far as my_spline_eval dereferences it (src/cosmo.c:2016-2027) */
typedef struct { size_t size; double *x, *y; } pf_spline_knots;

typedef double pfft_complex[2];

extern int ThisTask, NTasks;
extern product_data *products;
extern double **kdensity;
/* host work arrays of the FFT-module seam (src/pinocchio.h:270-272, 310-311; allocated by the reference in
   src/allocations.c:327-383): only the fine-seam functions below touch them */
extern double **density;
extern double ***first_derivatives, ***second_derivatives;
extern pfft_complex **cvector_fft;
extern double **rvector_fft;
extern smoothing_data Smoothing;
extern grid_data *MyGrids;
extern ScaleDep_data ScaleDep;
extern cputime_data cputime;
extern param_data params;
extern double Rsmooth;
extern char date_string[25];
/* SPLINE[SP_INVGROW] (src/cosmo.c:401) and the growth functions of src/cosmo.c:1789-1819,
   supplied by the caller in the stand-alone build */
extern pf_spline_knots pf_invgrow_knots;
extern double (*pf_GrowingMode)(double z, double k);
extern double (*pf_GrowingMode_2LPT)(double z, double k);
extern double (*pf_GrowingMode_3LPT_1)(double z, double k);
extern double (*pf_GrowingMode_3LPT_2)(double z, double k);
/* non-zero: behave like a -DSCALE_DEPENDENT build -- the growth functions above are sampled at the NkBINS k-bins
   (src/def_splines.h:40-42) and applied per mode; pf_invgrow_knots_radius[ismooth] = SPLINE_INVGROW[ismooth] */
/* non-zero: behave like a -DTABULATED_CT build (src/collapse_times.c:780-1231) */
extern int pf_compat_tabulated_ct;
/* non-zero: behave like a -DELL_SNG build (src/collapse_times.c:222-400); Hubble(z) in km/s/Mpc as src/cosmo.c:1691 */
extern int pf_compat_ell_sng;
extern double (*pf_Hubble)(double z);
extern double pf_compat_fr0; /* non-zero: -DMOD_GRAV_FR with this FR0 */
extern int pf_compat_scale_dependent;
extern pf_spline_knots pf_invgrow_knots_radius[64];

#endif
