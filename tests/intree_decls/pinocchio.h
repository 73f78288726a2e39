/*
 * tests/intree_decls/pinocchio.h -- DECLARATIONS ONLY, test infrastructure.
 *
 * Lets `cc -fsyntax-only -DPF_IN_PINOCCHIO_TREE pinocchio_amd/host/pf_compat.c` see the names the in-tree build of the
 * adapter binds with the types src/pinocchio.h, src/def_splines.h, <mpi.h> and <gsl/gsl_spline.h>
 * give them, so that those #ifdef branches are type-checked here, where MPI, GSL, FFTW and PFFT are not installed.
 * Nothing is defined here: this is NOT a build of the reference and NOT an oracle -- it pins no result.  (One test does
 * link and run the adapter compiled against it -- tests/test_mpi_boundary.py, with the globals defined by its own driver and
 * every pf_* entry point replaced by a recording mock: the MPI call order of the adapter is what it checks.)  Only what pf_compat.c touches is declared; the shared mirrors (product_data, grid_data, ...) come from
 * the adapter's own pf_compat_types.h, whose members follow src/pinocchio.h:233-378.
 */
#ifndef PF_TEST_INTREE_PINOCCHIO_H
#define PF_TEST_INTREE_PINOCCHIO_H

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* --- <mpi.h>: the calls and handles the adapter uses (-DPF_TEST_REAL_MPI: the header of an installed MPI instead, for
   tests/test_mpi_boundary.py, which links the adapter against a recording mock of the C ABI and runs it under mpiexec) --- */
#ifdef PF_TEST_REAL_MPI
#include <mpi.h>
#else
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
extern MPI_Comm MPI_COMM_WORLD;
extern MPI_Datatype MPI_BYTE, MPI_DOUBLE, MPI_UNSIGNED_LONG_LONG;
extern MPI_Op MPI_SUM;
int MPI_Bcast(void *buffer, int count, MPI_Datatype datatype, int root, MPI_Comm comm);
int MPI_Reduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype datatype, MPI_Op op, int root, MPI_Comm comm);
int MPI_Barrier(MPI_Comm comm);
#endif

/* --- <gsl/gsl_spline.h>: the members my_spline_eval and the adapter dereference --- */
typedef struct { size_t size; double *x; double *y; } gsl_spline;
typedef struct gsl_interp_accel_tag gsl_interp_accel;

/* --- src/def_splines.h:38-58 --- */
#ifdef SCALE_DEPENDENT
#define NkBINS 10
#define LOGKMIN ((double)-3.0)
#define DELTALOGK ((double)0.5)
#else
#define NkBINS 1
#endif
#define SP_INVGROW 7
#define SP_GROW1 9
#define SP_GROW2 (9 + NkBINS)
#define SP_GROW31 (9 + 2 * NkBINS)
#define SP_GROW32 (9 + 3 * NkBINS)
#define SP_PK (9 + 8 * NkBINS)

/* --- src/pinocchio.h: the path's state.  The records the adapter mirrors are taken from its own header, then the
   in-tree-only members are added through a differently named struct for params (k_for_GM, use_transposed_fft). --- */
#define param_data param_data_standalone_unused
#define params params_standalone_unused
#include "../../pinocchio_amd/host/pf_compat_types.h"
#undef param_data
#undef params

typedef struct /* src/pinocchio.h:311-352, the tags the adapter reads */
{
  double Omega0, OmegaLambda, OmegaBaryon, Hubble100, Sigma8, PrimordialIndex, BoxSize_htrue, k_for_GM, WDM_PartMass_in_kev;
  char RunFlag[SBLENGTH], DumpDir[SBLENGTH], CTtableFile[LBLENGTH], FileWithInputSpectrum[LBLENGTH];
  int GridSize[3], RandomSeed, use_transposed_fft, FixedIC, PairedIC;
} param_data;
extern param_data params;

typedef struct { int tasks_subdivision_dim; } internal_data; /* :200-219 */
extern internal_data internal;

extern gsl_spline **SPLINE;            /* :473 */
extern gsl_interp_accel **ACCEL;       /* :474 */
extern gsl_spline **SPLINE_INVGROW;    /* :476 */
extern double H_over_c;                /* :481 */
double Hubble(double);                                             /* :603 */
double OmegaMatter(double);
double OmegaLambda(double);
double GrowingMode(double, double);                                /* :612-615 */
double GrowingMode_2LPT(double, double);
double GrowingMode_3LPT_1(double, double);
double GrowingMode_3LPT_2(double, double);
double my_spline_eval(gsl_spline *, double, gsl_interp_accel *);   /* :630 */

/* names that exist only in the stand-alone build of the adapter: an in-tree branch must not touch them */
#pragma GCC poison pf_invgrow_knots pf_GrowingMode pf_GrowingMode_2LPT pf_GrowingMode_3LPT_1 pf_GrowingMode_3LPT_2
#pragma GCC poison pf_compat_lpt_order pf_compat_ct_interpolation
#pragma GCC poison pf_compat_tabulated_ct pf_compat_ell_sng pf_Hubble pf_compat_fr0 pf_compat_scale_dependent pf_invgrow_knots_radius
#pragma GCC poison params_standalone_unused

#endif
