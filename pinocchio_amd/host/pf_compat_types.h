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

/* sizes and axis names the adapter uses (src/pinocchio.h:56-65, 84-85) */
enum { NBINS = 210, LBLENGTH = 400, SBLENGTH = 100, MAXOUTPUTS = 100, ALIGN = 32 };
enum { _x_ = 0, _y_ = 1, _z_ = 2 };

#if defined(PF_IN_PINOCCHIO_TREE) && defined(DOUBLE_PRECISION_PRODUCTS) /* :219-225 (only the type-check of the in-tree branches gets here) */
typedef double PRODFLOAT;
#else
typedef float PRODFLOAT;
#endif

/* One mirror per reference struct: same member names, types and order (the adapter source compiles against either),
   one member per line with the field's role on the path. */
typedef struct product_data_mirror /* src/pinocchio.h:233-259, -DTWO_LPT -DTHREE_LPT */
{
  int       Rmax;           /* index of the smoothing radius that gave Fmax */
  PRODFLOAT Fmax;           /* 1 + z of collapse */
  PRODFLOAT Vel[3];         /* Zel'dovich displacement */
  /* (the type-check of the in-tree branches, tests/intree_decls/, drops the fields a build without TWO_LPT / THREE_LPT lacks) */
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(TWO_LPT) || defined(THREE_LPT)
  PRODFLOAT Vel_2LPT[3];    /* 2LPT */
#endif
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(THREE_LPT)
  PRODFLOAT Vel_3LPT_1[3];  /* 3LPT, first term */
  PRODFLOAT Vel_3LPT_2[3];  /* 3LPT, second term */
#endif
#ifdef RECOMPUTE_DISPLACEMENTS /* the reference's default Makefile flags: 104-byte record */
  PRODFLOAT Vel_prev[3];        /* the four above at the previous redshift segment, */
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(TWO_LPT) || defined(THREE_LPT)
  PRODFLOAT Vel_2LPT_prev[3];   /* filled by fragment.c's shift_all_displacements */
#endif
#if !defined(PF_IN_PINOCCHIO_TREE) || defined(THREE_LPT)
  PRODFLOAT Vel_3LPT_1_prev[3];
  PRODFLOAT Vel_3LPT_2_prev[3];
#endif
#endif
} product_data __attribute__((aligned(32))); /* on the typedef, as in the reference: aligns objects, does not pad the record */

typedef struct smoothing_data_mirror /* :284-292 */
{
  int     Nsmooth;        /* number of smoothing radii */
  double *Radius;         /* Mpc */
  double *Variance;       /* expected, from P(k) */
  double *TrueVariance;   /* measured on the grid: written by the path */
} smoothing_data;

typedef struct grid_data_mirror /* :295-308, without the pfft plans */
{
  unsigned int       total_local_size;      /* real cells of this task */
  unsigned int       total_local_size_fft;  /* doubles of the half-spectrum of this task */
  unsigned int       off;
  unsigned int       ParticlesPerTask;
  ptrdiff_t          GSglobal[3];
  ptrdiff_t          GSlocal[3];
  ptrdiff_t          GSstart[3];
  ptrdiff_t          GSlocal_k[3];
  ptrdiff_t          GSstart_k[3];
  double             lower_k_cutoff;
  double             upper_k_cutoff;
  double             norm;                  /* 1 / Ntotal */
  double             BoxSize;
  double             CellSize;
  unsigned long long Ntotal;
} grid_data;

typedef struct ScaleDep_data_mirror /* :536-542 */
{
  int    nseg;
  int    myseg;
  int    no_interp;
  int    order;            /* which growth multiplier compute_derivative applies: 0 none, 1..4 */
  double z[100];           /* MAXOUTPUTS redshift segments */
  double D[100];
  double D2[100];
  double D31[100];
  double D32[100];
  double redshift;
} ScaleDep_data;

typedef struct cputime_data_mirror /* :368-378, the accumulators the path fills */
{
  double fft;
  double coll;
  double lpt;
  double fmax;
  double io;
  double deriv;
  double mem_transf;
} cputime_data;

typedef struct param_data_mirror /* the tags of param_data (:311-352) the path uses; stand-alone order */
{
  char   RunFlag[100];
  char   DumpDir[100];
  int    GridSize[3];
  int    RandomSeed;
  double Omega0;              /* the next six: read by pf_compat_genic only */
  double OmegaBaryon;
  double Hubble100;
  double Sigma8;
  double PrimordialIndex;
  double BoxSize_htrue;
  double OmegaLambda;         /* with Omega0 and Hubble100 in the header of the collapse-time table file */
  char   CTtableFile[400];    /* "none": compute the tables (TABULATED_CT build) */
  int    use_transposed_fft;  /* UseTransposedFFT: the path keeps k-space in x, y, z order and refuses anything else */
  int    FixedIC;             /* GenIC options (src/GenIC.c:371-376), read by pf_compat_genic */
  int    PairedIC;
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
extern int pf_compat_lpt_order;    /* 3 (default), 2, 1: the -DTWO_LPT / -DTHREE_LPT choice of a stand-alone run */
extern int pf_compat_tabulated_ct;
extern int pf_compat_ct_interpolation;
/* non-zero: behave like a -DELL_SNG build (src/collapse_times.c:222-400); Hubble(z) in km/s/Mpc as src/cosmo.c:1691 */
extern int pf_compat_ell_sng;
extern double (*pf_Hubble)(double z);
extern double pf_compat_fr0; /* non-zero: -DMOD_GRAV_FR with this FR0 */
extern int pf_compat_scale_dependent;
extern pf_spline_knots pf_invgrow_knots_radius[64];

#endif
