/*
 * tests/mpi_boundary/driver.c -- TEST INFRASTRUCTURE: drives the in-tree build of the adapter (pf_compat.c,
 * -DPF_IN_PINOCCHIO_TREE) the way the reference's main() and its restart path do (src/pinocchio.c:146-150, 229;
 * src/initialization.c:139, 512; src/fmax.c:372-506, 527), under a real MPI, against the recording mock of the C ABI.
 * It defines the globals src/variables.c would define; it computes nothing.  Output: one tagged line per fact on stdout.
 */
#include "pinocchio.h"

int ThisTask, NTasks;
product_data *products;
static double *kdensity_slots[1];
double **kdensity = kdensity_slots;
double **density;
double ***first_derivatives, ***second_derivatives;
static pfft_complex *cvector_slots[1];
static double *rvector_slots[1];
pfft_complex **cvector_fft = cvector_slots;
double **rvector_fft = rvector_slots;
smoothing_data Smoothing;
static grid_data grid0;
grid_data *MyGrids = &grid0;
ScaleDep_data ScaleDep;
cputime_data cputime;
param_data params;
internal_data internal;
double Rsmooth;
char date_string[25];
static gsl_spline *spline_slots[128];
gsl_spline **SPLINE = spline_slots;
gsl_interp_accel **ACCEL;
gsl_spline **SPLINE_INVGROW;
double H_over_c = 100. / 299792.458;
double Hubble(double z) { return 100. * (1. + z); }
double OmegaMatter(double z) { (void)z; return 0.3; }
double OmegaLambda(double z) { (void)z; return 0.7; }
double GrowingMode(double z, double k) { (void)k; return 1. / (1. + z); }
double GrowingMode_2LPT(double z, double k) { (void)z; (void)k; return 3. / 7.; }
double GrowingMode_3LPT_1(double z, double k) { (void)z; (void)k; return -1. / 9.; }
double GrowingMode_3LPT_2(double z, double k) { (void)z; (void)k; return 5. / 42.; }
double my_spline_eval(gsl_spline *s, double x, gsl_interp_accel *a) { (void)s; (void)a; return x; }

int set_one_grid(int);
int compute_fft_plans(void);
int compute_fmax(void);
int dump_products(void);
int read_dumps(void);
int Fmax_PDF(void);

#define NS 3
int main(int argc, char **argv) {
  static double radius[NS] = {4.0, 2.0, 0.0}, variance[NS] = {1.0, 2.0, 3.0}, truevar[NS];
  static double knots_x[4] = {-4, -2, -1, 0}, knots_y[4] = {-4, -2, -1, 0};
  static gsl_spline invgrow = {4, knots_x, knots_y};
  int n = argc > 1 ? atoi(argv[1]) : 16, provided, rc, i;
  unsigned long long sum = 0;
  unsigned int k;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_FUNNELED, &provided);
  MPI_Comm_rank(MPI_COMM_WORLD, &ThisTask);
  MPI_Comm_size(MPI_COMM_WORLD, &NTasks);
  for (i = 0; i < 3; i++) { grid0.GSglobal[i] = n; params.GridSize[i] = n; }
  grid0.Ntotal = (unsigned long long)n * n * n;
  grid0.BoxSize = 2.0 * n;
  strcpy(params.RunFlag, "mpitest");
  strcpy(params.DumpDir, "Dumps/");
  strcpy(params.CTtableFile, "none");
  strcpy(params.FileWithInputSpectrum, "no");
  params.RandomSeed = 486604;
  internal.tasks_subdivision_dim = 1;
  SPLINE[SP_INVGROW] = &invgrow;
  Smoothing.Nsmooth = NS; Smoothing.Radius = radius; Smoothing.Variance = variance; Smoothing.TrueVariance = truevar;
  ScaleDep.nseg = 1; ScaleDep.z[0] = 0.0;

  rc = set_one_grid(0);
  printf("GRID task=%d rc=%d xl=%ld x0=%ld cells=%u\n", ThisTask, rc, (long)grid0.GSlocal[_x_], (long)grid0.GSstart[_x_], grid0.total_local_size);
  if (rc) { MPI_Finalize(); return 2; }
  rc = compute_fft_plans();
  printf("PLANS task=%d rc=%d\n", ThisTask, rc);
  fflush(stdout);
  if (rc) { MPI_Finalize(); return 3; }  /* (every task sees the same communicator count: all leave together) */

  products = (product_data *)calloc(grid0.total_local_size, sizeof(product_data));
  kdensity[0] = (double *)calloc(grid0.total_local_size_fft, sizeof(double));
  rc = compute_fmax();
  printf("FMAX task=%d rc=%d tv=%g,%g,%g\n", ThisTask, rc, truevar[0], truevar[1], truevar[2]);
  if (rc) { MPI_Finalize(); return 4; }
  rc = dump_products();
  printf("DUMP task=%d rc=%d\n", ThisTask, rc);
  MPI_Barrier(MPI_COMM_WORLD);

  /* a run restarted from the dumps (ReadProductsFromDumps, src/pinocchio.c:229): nothing in memory, no device context */
  memset(products, 0, grid0.total_local_size * sizeof(product_data));
  memset(truevar, 0, sizeof(truevar));
  rc = read_dumps();
  for (k = 0; k < grid0.total_local_size; k++) sum += (unsigned long long)(products[k].Fmax * 10.f + 0.5f) + 1000ull * (unsigned)products[k].Rmax;
  printf("READ task=%d rc=%d tv=%g,%g,%g sum=%llu\n", ThisTask, rc, truevar[0], truevar[1], truevar[2], sum);
  strcpy(params.RunFlag, "mpitest_host");
  rc = Fmax_PDF();
  printf("PDF task=%d rc=%d\n", ThisTask, rc);
  MPI_Finalize();
  return 0;
}
