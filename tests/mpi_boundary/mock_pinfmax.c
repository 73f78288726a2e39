/*
 * tests/mpi_boundary/mock_pinfmax.c -- TEST INFRASTRUCTURE: a recording mock of the C ABI (include/pinfmax.h) for the one
 * test that runs the MPI side of the adapter (pinocchio_amd/host/pf_compat.c built -DPF_IN_PINOCCHIO_TREE) under a real
 * MPI (MPICH: /opt/conda/bin/mpicc, mpiexec -n 2 / 4).  No device, no arithmetic: every entry point appends one line to
 * calls.<task>.log in the working directory, and the few that are collective in the real library are collective here too:
 *   pf_rccl_unique_id   task 0 only: a fixed 128-byte pattern
 *   pf_init_rccl        every task: checks that the id it was handed is that pattern (i.e. the broadcast came first), then an
 *                       MPI_Allreduce counts who joined -- as ncclCommInitRank it returns only when all tasks have called it.
 *                       PF_MOCK_OUTSIDER=<task>: that task joins with weight 0 (a task that is not part of the communicator)
 *   pf_rccl_comm_count  the count of that all-reduce
 *   pf_sweep            fills true_variance[i] = 1000 + i on every task (the real library all-reduces it)
 *   pf_get_products     record k of task t gets Fmax = 0.05 + 0.1 ((k + t) % 250), Rmax = t
 */
#include <mpi.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pinfmax.h"

struct pf_ctx { int rank, nranks, comm_count, n; };
static struct pf_ctx the_ctx;
static int live = 0;

static void rec(const char *fmt, ...) {
  char name[64];
  int task = 0;
  FILE *f;
  va_list ap;
  MPI_Comm_rank(MPI_COMM_WORLD, &task);
  snprintf(name, sizeof(name), "calls.%d.log", task);
  f = fopen(name, "a");
  if (!f) abort();
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fputc('\n', f);
  fclose(f);
}

static void id_pattern(unsigned char *id) { int i; for (i = 0; i < 128; i++) id[i] = (unsigned char)(37 * i + 11); }

int pf_create(pf_ctx **out, const pf_config *cfg) {
  rec("pf_create n=%d rank=%d nranks=%d device=%d field_bytes=%d", cfg->n, cfg->rank, cfg->nranks, cfg->device, cfg->field_bytes);
  the_ctx.rank = cfg->rank; the_ctx.nranks = cfg->nranks; the_ctx.comm_count = 0; the_ctx.n = cfg->n;
  live = 1;
  *out = &the_ctx;
  return 0;
}
int pf_destroy(pf_ctx *ctx) { rec("pf_destroy"); live = 0; (void)ctx; return 0; }
const char *pf_last_error(void) { return "mock"; }
int pf_rccl_unique_id(void *id128) {
  int task;
  MPI_Comm_rank(MPI_COMM_WORLD, &task);
  rec("pf_rccl_unique_id");
  if (task != 0) { rec("FAULT pf_rccl_unique_id called on task %d", task); return 1; }
  id_pattern((unsigned char *)id128);
  return 0;
}
int pf_init_rccl(pf_ctx *ctx, const void *id128) {
  unsigned char want[128];
  const char *outsider = getenv("PF_MOCK_OUTSIDER");
  int mine[2] = {1, 0}, all[2] = {0, 0};
  id_pattern(want);
  mine[1] = memcmp(want, id128, 128) ? 1 : 0;  /* a task that was handed something else spoils the set-up for everyone, as with RCCL */
  rec("pf_init_rccl id_ok=%d", !mine[1]);
  if (outsider && atoi(outsider) == ctx->rank) mine[0] = 0;
  MPI_Allreduce(mine, all, 2, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  ctx->comm_count = all[0];
  return all[1] ? 1 : 0;
}
int pf_rccl_comm_count(pf_ctx *ctx) { rec("pf_rccl_comm_count -> %d", ctx->comm_count); return ctx->comm_count; }

#define PLAIN(name, args) int name args { rec(#name); return 0; }
PLAIN(pf_set_lpt_order, (pf_ctx *ctx, int order))
PLAIN(pf_set_ct_interpolation, (pf_ctx *ctx, int flavour))
PLAIN(pf_set_transposed_spectra, (pf_ctx *ctx, int on))
PLAIN(pf_set_invgrow, (pf_ctx *ctx, int ismooth, const double *x, const double *y, int n))
PLAIN(pf_set_sources_in_sweep, (pf_ctx *ctx, int on))
PLAIN(pf_set_modified_gravity, (pf_ctx *ctx, double fr0, double h_over_c, int nsmooth, const double *size))
PLAIN(pf_set_growth_table, (pf_ctx *ctx, int order, const double *t, int nk, double logkmin, double dlogk, double sign))
PLAIN(pf_set_collapse_model, (pf_ctx *ctx, int model, const double cosmo[4], int nsmooth, const double *D_in))
PLAIN(pf_second_derivatives, (pf_ctx *ctx, double radius_cells))
PLAIN(pf_update_products, (pf_ctx *ctx, void *products_host, const pf_product_layout *layout))
PLAIN(pf_set_growth, (pf_ctx *ctx, const double g[4]))
PLAIN(pf_set_density, (pf_ctx *ctx, const double *kdensity_slab))
PLAIN(pf_reverse_transform, (pf_ctx *ctx, const double *spec_host, double *real_host))
PLAIN(pf_forward_transform, (pf_ctx *ctx, const double *real_host, double *spec_host))
PLAIN(pf_reset_cputime, (pf_ctx *ctx))
PLAIN(pf_pk_norm, (const pf_genic_params *p, double sigma8, double *pknorm))
PLAIN(pf_get_second_derivative, (pf_ctx *ctx, int i, double *host))
PLAIN(pf_genic_density, (pf_ctx *ctx, const pf_genic_params *p))
PLAIN(pf_displacements, (pf_ctx *ctx, int compute_sources, int recompute_sd))
PLAIN(pf_derivative, (pf_ctx *ctx, const double *spec_host, int fd, int sd, double rs_cells, int order, double *real_host))
PLAIN(pf_ct_load, (pf_ctx *ctx, int ismooth, double variance, const double *table_host))
PLAIN(pf_ct_build, (pf_ctx *ctx, int ismooth, double variance, double *table_host))
PLAIN(pf_collapse_times, (pf_ctx *ctx, int ismooth, double *true_variance))

int pf_get_cputime(pf_ctx *ctx, pf_cputime *t) { rec("pf_get_cputime"); memset(t, 0, sizeof(*t)); return 0; }
int pf_sweep(pf_ctx *ctx, int ns, const double *radius_cells, double *true_variance) {
  int i;
  rec("pf_sweep ns=%d", ns);
  for (i = 0; i < ns; i++) true_variance[i] = 1000.0 + i;
  return 0;
}
/* the device histogram is "already summed over the ranks": what every task would hold after the library's all-reduce */
int pf_fmax_pdf(pf_ctx *ctx, unsigned long long hist[PF_NBINS]) {
  int b;
  rec("pf_fmax_pdf");
  for (b = 0; b < PF_NBINS; b++) hist[b] = (unsigned long long)(b * ctx->nranks);
  return 0;
}
int pf_get_products(pf_ctx *ctx, void *products_host, const pf_product_layout *lay) {
  const long long cells = (long long)ctx->n * ctx->n * (ctx->n / ctx->nranks);
  long long k;
  rec("pf_get_products stride=%d", (int)lay->stride);
  for (k = 0; k < cells; k++) {
    char *r = (char *)products_host + k * (long long)lay->stride;
    const float F = 0.05f + 0.1f * (float)((k + ctx->rank) % 250);
    const int R = ctx->rank;
    memcpy(r + lay->off_Fmax, &F, sizeof(F));
    memcpy(r + lay->off_Rmax, &R, sizeof(R));
  }
  return 0;
}
