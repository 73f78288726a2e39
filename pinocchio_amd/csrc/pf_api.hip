// pf_api.hip -- context, slab 3-D FFT pipelines and the C ABI of libpinfmax_hip.so.
//
// Data layout in HBM (per rank; P ranks, nxl = nyl = n/P, nzh = n/2+1,
// nzp = n/2+8 (fp64; n/2+16 for fp32 fields) so that every row of a half-spectrum starts on a 128-byte line):
//   KY  k-space, y-distributed : [n (kx)][nyl (ky slab)][nzp]  complex F   (dk, A*, S* spectra)
//   XS  x-slab, k in y and z   : [nxl][n (ky)][nzp]            complex F   (B*)
//   R   real x-slab            : [nxl][n][2*nzp]               F, first n of a row used
//       (a real row lives in the storage of the complex row it came from: c2r and
//        r2c z-passes run in place)
//   products                   : SoA  fmax f32, rmax i32, vel[12] f32, cell = z + n*(y + n*xl)
//
// One c2r 3-D transform = x-pass (KY) -> [all-to-all, P > 1] -> y-pass -> z-pass (XS -> R).
// The six second derivatives share passes (the filter k_a k_b / k^2 * W(k) is
// separable apart from W/k^2):  x-pass 1 -> 3 fields {1, kx, kx^2} * dk*W/k^2,
// y-pass 3 -> 6, z-pass 6 -> 6: 4 + 9 + 12 = 25 field transfers instead of the
// reference's 6 x (1 + 6) = 42 (src/fmax.c:225-258), and 3 all-to-alls instead of 6.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <functional>
#include <vector>

#include "../../include/pinfmax.h"
#include "pf_collapse_core.h"
#include "pf_hostpool.h"
#include "pf_internal.h"

extern "C" void pf_rccl_release(void *link);  // pf_rccl.cpp
extern "C" int pf_rccl_link_count(void *link);

// ------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";
static int pf_fail(int task, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  printf("ERROR on task %d: %s\n", task, g_err);  // the reference's convention (src/fmax.c, src/collapse_times.c:575)
  fflush(stdout);
  return 1;
}
extern "C" const char *pf_last_error(void) { return g_err; }

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return pf_fail((ctx) ? (ctx)->rank : 0, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)
#define PFCHK(ctx, call)                                                                   \
  do {                                                                                     \
    int r__ = (call);                                                                      \
    if (r__) return pf_fail((ctx)->rank, "%s failed (%d) (%s:%d)", #call, r__, __FILE__, __LINE__); \
  } while (0)

#define PFCHK0(call)                                                                       \
  do {                                                                                     \
    int r__ = (call);                                                                      \
    if (r__) return pf_fail(0, "%s failed (%d) (%s:%d)", #call, r__, __FILE__, __LINE__);  \
  } while (0)

// ----------------------------------------------------------------- context --
enum {
  KS_XPASS_HESS = 0, KS_YPASS_HESS, KS_ZPASS_HESS, KS_COLLAPSE, KS_LPT_SRC, KS_LPT_ACC, KS_R2C_Z, KS_YPASS_FWD,
  KS_XPASS_FWD, KS_XPASS_DISP, KS_YPASS_DISP, KS_ZPASS_DISP, KS_XPASS_PLAIN, KS_YPASS_PLAIN, KS_ZPASS_PLAIN,
  KS_EXCHANGE, KS_MISC, KS_ZCOLLAPSE, KS_ZPASS_INV, KS_COLLAPSE_INV, KS_ZPASS_LPT3B, KS_COLLAPSE_SRC, KS_COUNT
};
static const char *ks_names[KS_COUNT] = {
    "xpass_hess_1to3", "ypass_hess_3to6", "zpass_c2r_hess_6", "collapse", "lpt_sources", "lpt_accum", "zpass_r2c",
    "ypass_fwd", "xpass_fwd", "xpass_disp_1to2", "ypass_disp_2to3", "zpass_c2r_disp_3", "xpass_plain", "ypass_plain",
    "zpass_c2r_plain", "exchange", "misc", "zpass_collapse_fused", "zpass_c2r_hess_6to3inv", "collapse_inv", "zpass_c2r_hess_6_lpt3b",
    "collapse_lpt_sources"};

struct EvPair { int kind; hipEvent_t a, b; double bytes; };

struct SplineHost { std::vector<double> x, y, c; };

struct PfLoopback { int copies_left; int nranks; };  // pf_set_loopback_exchange
// Host side of the boundary.  Everything that crosses it in bulk -- products out (src/fmax-pfft.c:563-631 writes them into host
// memory), kdensity in -- goes through two pinned buffers on two streams: the DMA of piece i + 1 runs while the host threads move
// piece i between its buffer and the caller's pageable array (a copy, or the scatter of the named fields for pf_update_products).
// PF_HANDOFF_CHUNK_MB (256), PF_HANDOFF_THREADS (the cores of the cgroup, at most 16), PF_HOST_REGISTER=1: the caller's product
// array is registered with the driver at first sight (hipHostRegister, page-aligned arrays only; silently not when that fails) and
// the DMA writes into it directly.
struct PfHandoff {
  hipStream_t st[2];
  char *pin[2];
  size_t chunk;
  PfHostPool *pool;
  hipEvent_t ev;
  bool want_register;
  void *reg_ptr; size_t reg_bytes;
};

struct pf_ctx {
  pf_config cfg;
  int n, nzh, nzp, P, rank, nxl, nyl, fb;
  bool timing;
  hipStream_t stream;   // every kernel of the path; the exchanges of a pipelined multi-rank run go to cstream
  int collapse_blocks;
  bool fast_libm;
  int ncu, dev;
  bool solve_ran_beside;  // the last sweep used the solve stream (pf_solve_ran_beside_zpass)
  int inv_reruns;       // sweeps repeated with six components because the invariant z-pass met a q == 0 cell (pf_sweep)
  PfTuning tune;        // run-time switches, read from the environment once, in pf_create
  double prune_eps;
  bool own_stream;
  size_t field_bytes;  // one spectrum-sized field
  size_t dev_bytes;
  char *blockA;        // A[0..2] contiguous (also host<->device staging through A[1..2])
  void *dk, *A[3], *B[6], *B2[6], *S[3];
  // S[0..2] and B2[0..5] are pieces of ONE allocation each (blockS at pf_create, blockB2 at first use: the LPT part, or a sweep that
  // needs a second field set).  fp32 fields: the fp64 invariant rows of the sweep live in them -- a row set of ncell doubles is two
  // fields long to within the row padding -- since the sweep and the LPT part never hold both (inv_rows; 234 -> 200 GB per rank for
  // BASELINE config 5)
  char *blockS, *blockB2;
  void *recvA;         // P > 1: 3 fields, all-to-all destination
  // P > 1 (PF_REPLICATE_DK, default on): every rank also holds the WHOLE delta(k), [kx][ky][kz] with row pitch nzp.  The
  // passes that start from delta(k) -- the second derivatives of every radius and the Zel'dovich displacements -- then
  // transform every line on every rank and keep only their own x-slab: no all-to-all for 38 of the 50 field exchanges of
  // a 12-radius step, at the price of an x-pass that does not shrink with P.  Gathered once per density (dk_full_valid).
  void *dk_full;
  bool replicate, dk_full_valid;
  // the recipe of the resident density when it came from pf_genic_density (with its own copies of a tabulated spectrum's knots): a rank
  // that runs on its own behind the loopback exchange generates the WHOLE replicated spectrum from it (ensure_dk_full)
  bool have_genic;
  pf_genic_params genic;
  std::vector<double> genic_logk, genic_logk3p;
  double *INV[2][3];   // fp32 fields: the fp64 invariant rows of the sweep's z-pass (compact, pitch n), placed at first use (inv_rows); two
                       // sets for the solve stream (the z-pass writes set inv_w while the solve of the radius before reads the other).
                       // Set 1 is blockB2, INV[0][0] is blockS (fields 0, 1); INV[0][1..2] are allocations of their own (inv_own)
  double *inv_own[2];
  int inv_w;
  // P > 1: second set of send/receive fields and a communication stream, so that the all-to-all of transform i+1
  // runs beside the y/z passes (and the collapse solve) of transform i (pipelined(), PF_PIPELINE=0 to disable)
  bool pipeline;
  hipStream_t cstream;
  // the solve stream of a sweep (PF_SOLVE_BESIDE_Z, sweep_body): the solve of radius i runs beside the z-pass of radius i + 1,
  // whose passes write the other of the two field sets B / B2.  ev_y[set]: the y-pass that follows the z-pass which wrote the
  // invariants of `set` is enqueued; ev_s[set]: the solve that read them is done
  hipStream_t sstream;
  hipEvent_t ev_y[2], ev_s[2];
  char *blockA2;
  void *recvA2;
  hipEvent_t ev_x[2], ev_r[2];
  void *tw;
  double *etab;        // [n] window of the transformed axis for the x-pass in flight (pf_launch_exp_table)
  // products, SoA: Fmax and the twelve displacement columns as PRODFLOAT -- float, or double with PF_FLAG_DOUBLE_PRODUCTS (pb = 8)
  void *fmax, *vel12;
  int pb;
  int *rmax;
  double *partials;    // 2 * PF_NBLK
  double *scal;        // device scalars, see SC_*
  unsigned long long *hist;
  double *spl;         // device spline tables: [slot][5][PF_KNOT_CAP] = x, y, c, b, d
  int spl_n[PF_MAX_SMOOTH + 1];
  bool spl_set[PF_MAX_SMOOTH + 1];
  // the same splines as polynomial tables of 10^(-S(log10 D)) in D (pf_gtab.h): [slot][PF_GT_DOUBLES], start tables [slot][PF_GT_MAX_BINS]
  double *gt;
  unsigned short *gt_lut;
  bool gt_ok[PF_MAX_SMOOTH + 1];
  double gt_err[PF_MAX_SMOOTH + 1];
  double growth[4];
  // k-binned growth (SCALE_DEPENDENT build): per order log10-growth table on the device, 0 entries = scalar
  int gt_n[4];
  double gt_logkmin[4], gt_dlogk[4], gt_sign[4];
  double *gtab;        // [4][PF_KBIN_CAP]
  // TABULATED_CT build: variance per radius (0 radii = direct solve) and the table of the radius in place
  int tab_ns;
  double tab_var[PF_MAX_SMOOTH];
  bool tab_ready;
  int tab_ismooth;
  int model;           // 0 ELL_CLASSIC, 1 ELL_SNG (fills the collapse-time table of a TABULATED_CT build; per cell without one)
  double sng_cosmo[7], sng_Din[PF_MAX_SMOOTH], sng_size[PF_MAX_SMOOTH];
  int sng_ns;
  PfCtDev ct;
  double *ct_block;    // delta | alpha | gamma | y | b | c | d
  bool have_density, have_hessian, have_sources, products_init;
  int sources_order;  // LPT order the resident source spectra were made for
  // pf_set_sources_in_sweep: the solve of a sweep's last radius also writes the LPT sources (S[0..2], real space) and the
  // sum of S2; sources_fresh says they are what pf_displacements(1, 0) would compute from the Hessian in B
  bool sweep_sources, sources_fresh;
  bool transposed; // spectra cross the boundary as [ky_local][kx][kz] (pf_set_transposed_spectra)
  int ct_flavour; // table interpolation of the build: 0 BILINEAR_SPLINE, 1 TRILINEAR, 2 ALL_SPLINE (pf_set_ct_interpolation)
  int lpt_order;  // 3: -DTWO_LPT -DTHREE_LPT (default), 2: -DTWO_LPT only, 1: Zel'dovich only (pf_set_lpt_order)
  double *partials_src;  // PF_NBLK
  // general path: grid sizes that are not a power of two (one rank, fp64) go through library transforms
  bool general;
  void *W;                  // scratch spectrum of the filter
  void *fft_c2r, *fft_r2c;  // the chirp-z plan of the general path (pf_gfft.hip: one plan under both names)
  bool vel_zero_pending;  // the Vel* columns are to read as zero (src/collapse_times.c:472-489) but have not been cleared yet
  int last_ns;
  struct PfHandoff *handoff;  // host side of the boundary (handoff_get): two streams, two pinned buffers, the host threads
  PfLoopback *loopback;
  pf_alltoall_fn a2a; void *a2a_user;
  pf_alltoallv_fn a2av; void *a2av_user;  // optional: exchange of a row range of every block (pruned radii)
  pf_allreduce_fn ared; void *ared_user;
  void *rccl;          // ncclComm_t when pf_init_rccl was used
  // measurement
  std::vector<EvPair> evs;
  std::vector<hipEvent_t> evpool;
  double ks_ms[KS_COUNT], ks_bytes[KS_COUNT];
  unsigned long long ks_n[KS_COUNT];
  pf_cputime cpu;
  std::vector<EvPair> phase_evs;  // kind: 0 deriv 1 coll 2 lpt 3 mem
};

static void handoff_release(pf_ctx *c);
#define PF_GT_DOUBLES (PF_GT_HEADER + (PF_GT_MAX_INT + 1) * PF_GT_REC)
#define PF_NBLK 2048
#define PF_KBIN_CAP 32
enum { SC_SUM = 0, SC_SUM2 = 1, SC_DC_DK = 2, SC_DC_S2 = 3, SC_POWER = 4, SC_DSCALE = 5, SC_DC_TMP = 6, SC_INV_FLAG = 7 /* raised by the invariant z-pass, see pf_sweep; adjacent to SC_VAR0: one all-reduce */, SC_VAR0 = 8 /* 2 per radius */, SC_COUNT = 8 + 2 * PF_MAX_SMOOTH };

static hipEvent_t ev_get(pf_ctx *c) {
  if (!c->evpool.empty()) { hipEvent_t e = c->evpool.back(); c->evpool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
struct KTimer {
  pf_ctx *c; int kind; double bytes; hipEvent_t a; hipStream_t st;
  KTimer(pf_ctx *c_, int kind_, double bytes_, hipStream_t st_ = nullptr) : c(c_), kind(kind_), bytes(bytes_), a(nullptr), st(st_ ? st_ : c_->stream) {
    if (c->timing) { a = ev_get(c); hipEventRecord(a, st); }
  }
  ~KTimer() {
    if (c->timing) { hipEvent_t b = ev_get(c); hipEventRecord(b, st); c->evs.push_back({kind, a, b, bytes}); }
  }
};
struct PhaseTimer {
  pf_ctx *c; int kind; hipEvent_t a; hipStream_t st;
  PhaseTimer(pf_ctx *c_, int kind_, hipStream_t st_ = nullptr) : c(c_), kind(kind_), st(st_ ? st_ : c_->stream) { a = ev_get(c); hipEventRecord(a, st); }
  ~PhaseTimer() { hipEvent_t b = ev_get(c); hipEventRecord(b, st); c->phase_evs.push_back({kind, a, b, 0}); }
};
static void resolve_events(pf_ctx *c) {
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->cstream);
  hipStreamSynchronize(c->sstream);
  for (auto &e : c->evs) {
    float ms = 0; hipEventElapsedTime(&ms, e.a, e.b);
    c->ks_ms[e.kind] += ms; c->ks_bytes[e.kind] += e.bytes; c->ks_n[e.kind]++;
    c->evpool.push_back(e.a); c->evpool.push_back(e.b);
  }
  c->evs.clear();
  for (auto &e : c->phase_evs) {
    float ms = 0; hipEventElapsedTime(&ms, e.a, e.b);
    double s = 1e-3 * ms;
    if (e.kind == 0) c->cpu.deriv += s; else if (e.kind == 1) c->cpu.coll += s; else if (e.kind == 2) c->cpu.lpt += s;
    else if (e.kind == 3) c->cpu.mem_transf += s; else if (e.kind == 4) c->cpu.fmax += s;
    c->evpool.push_back(e.a); c->evpool.push_back(e.b);
  }
  c->phase_evs.clear();
}

static size_t ncell_of(const pf_ctx *c) { return (size_t)c->nxl * c->n * c->n; }
static int dev_alloc(pf_ctx *c, void **p, size_t bytes) {
  HIPCHK(c, hipMalloc(p, bytes));
  c->dev_bytes += bytes;
  return 0;
}

// the second field set B2[0..5] (one allocation): the Hessian of the 2LPT potential and the scratch of the displacement passes, the
// second field set of a sweep with the solve stream and fp64 fields, the second set of invariant rows with fp32 fields.
// quiet: a failure is the caller's to handle (a sweep falls back to one field set), nothing is printed
static int ensure_b2(pf_ctx *c, bool quiet = false) {
  if (c->blockB2) return 0;
  if (hipMalloc((void **)&c->blockB2, 6 * c->field_bytes) != hipSuccess) {
    (void)hipGetLastError(); c->blockB2 = nullptr;
    return quiet ? 1 : pf_fail(c->rank, "hipMalloc of the second field set failed (%.1f GB; %.1f GB held by this context)", 6e-9 * c->field_bytes, 1e-9 * c->dev_bytes);
  }
  c->dev_bytes += 6 * c->field_bytes;
  for (int i = 0; i < 6; i++) c->B2[i] = c->blockB2 + i * c->field_bytes;
  return 0;
}
// fp32 fields: where the three fp64 invariant rows (ncell doubles each, pitch n) of set 0 / 1 live.  Set 1: the six fields of
// blockB2, two per row set; set 0: blockS (fields 0 and 1) and two allocations of its own.  2 fields = n nyl (n + 32) 8 bytes
// >= ncell 8 bytes, whatever P.  The LPT source spectra that may be resident in S are gone after a sweep that uses them so.
static int inv_rows(pf_ctx *c, int set, bool quiet = false) {
  if (c->INV[set][0]) return 0;
  const size_t row_set = ncell_of(c) * sizeof(double);
  if (2 * c->field_bytes < row_set) return pf_fail(c->rank, "internal: an invariant row set does not fit two fields");
  if (set == 1) {
    if (ensure_b2(c, quiet)) return 1;
    for (int k = 0; k < 3; k++) c->INV[1][k] = (double *)(c->blockB2 + 2 * k * c->field_bytes);
    return 0;
  }
  for (int k = 0; k < 2; k++)
    if (!c->inv_own[k]) PFCHK(c, dev_alloc(c, (void **)&c->inv_own[k], row_set));
  c->INV[0][0] = (double *)c->blockS; c->INV[0][1] = c->inv_own[0]; c->INV[0][2] = c->inv_own[1];
  return 0;
}

// twiddles exp(+2 pi i j / n), computed in long double on the host, exact on the axes
static void host_twiddles(int n, int fb, std::vector<char> &h) {
  h.resize((size_t)n * 2 * fb);
  for (int j = 0; j < n; j++) {
    long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)n;
    double re = (double)cosl(a), im = (double)sinl(a);
    if (j == 0) { re = 1; im = 0; } else if (4 * j == n) { re = 0; im = 1; } else if (2 * j == n) { re = -1; im = 0; } else if (4 * j == 3 * n) { re = 0; im = -1; }
    if (fb == 8) { ((double *)h.data())[2 * j] = re; ((double *)h.data())[2 * j + 1] = im; }
    else { ((float *)h.data())[2 * j] = (float)re; ((float *)h.data())[2 * j + 1] = (float)im; }
  }
}

extern "C" void pf_layout_3lpt(pf_product_layout *l) {
  l->stride = 56; l->off_Rmax = 0; l->off_Fmax = 4; l->off_Vel = 8; l->off_Vel_2LPT = 20; l->off_Vel_3LPT_1 = 32; l->off_Vel_3LPT_2 = 44;
}

static size_t ncell(const pf_ctx *c) { return (size_t)c->nxl * c->n * c->n; }
static long long rpitch(const pf_ctx *c) { return c->general ? c->n : 2 * c->nzp; }  // reals per row of a real field
static double spec_bytes_alg(const pf_ctx *c) { return (double)c->n * c->nyl * c->nzh * 2.0 * c->fb; }  // one half-spectrum field
static double real_bytes_alg(const pf_ctx *c) { return (double)ncell(c) * c->fb; }

static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
// The run-time switches of DESIGN.md section 6: read here, once per context, never on a launch path.
static void read_tuning(PfTuning *t) {
  t->zpass_persist = env_int("PF_ZPASS_PERSIST", 24);
  t->zpass_inv_wg_per_cu = env_int("PF_ZPASS_INV_WG_PER_CU", 32);
  if (t->zpass_inv_wg_per_cu <= 0) t->zpass_inv_wg_per_cu = 32;
  t->spline_lut = env_int("PF_SPLINE_LUT", 1) != 0;
  t->exchange_rows = env_int("PF_EXCHANGE_ROWS", 1) != 0;
  t->invariants = env_int("PF_INVARIANTS", 1);
  t->lpt_fuse = env_int("PF_LPT_FUSE", 1);
  t->collapse_wg_per_cu = env_int("PF_COLLAPSE_WG_PER_CU", 8);
  if (t->collapse_wg_per_cu <= 0) t->collapse_wg_per_cu = 8;
  t->general = env_int("PF_GENERAL", 0) != 0;

  t->pipeline = env_int("PF_PIPELINE", 1) != 0;
  t->replicate = env_int("PF_REPLICATE_DK", -1);  // -1: by the number of ranks (pf_create), 0 / 1: off / on
  t->exact_libm = env_int("PF_EXACT_LIBM", 0) != 0;
  t->solve_beside_z = env_int("PF_SOLVE_BESIDE_Z", -1);
  t->gtab = env_int("PF_GTAB", 1) != 0;
  t->preflight = env_int("PF_PREFLIGHT", 1) != 0;
  t->handoff_chunk_mb = env_int("PF_HANDOFF_CHUNK_MB", 256);
  if (t->handoff_chunk_mb <= 0) t->handoff_chunk_mb = 256;
  t->handoff_threads = env_int("PF_HANDOFF_THREADS", 0);
  t->host_register = env_int("PF_HOST_REGISTER", 0) != 0;
  // fault injection for the tests of the exchange pipeline (tests/test_gpu_multirank.py): "recv" drops the wait of the
  // compute stream for the exchange it is about to consume, "send" the wait of the exchange for the x-pass that fills its blocks
  const char *fault = getenv("PF_DEBUG_PIPELINE_FAULT");
  t->debug_fault = !fault ? 0 : !strcmp(fault, "recv") ? 1 : !strcmp(fault, "send") ? 2 : 0;
  t->prune_eps = 8.673617379884035e-19;  // 2^-60; PF_PRUNE_EPS=0 transforms every mode
  if (const char *e = getenv("PF_PRUNE_EPS")) t->prune_eps = atof(e);
}

// Device memory of a context, in bytes: what pf_create allocates (*at_create) and the most it holds once a sweep and the LPT part
// have run (*peak: + the second field set, + two invariant row sets of their own with fp32 fields).  The formula create_body and
// the lazy allocations follow; pf_create holds it against hipMemGetInfo before it allocates anything.
static void plan_bytes(const pf_ctx *c, size_t *at_create, size_t *peak) {
  const size_t field = c->field_bytes, nc = ncell_of(c);
  size_t fields = 1 + 3 + 6 + 3;                       // dk, A, B, S
  if (c->P > 1) fields += 3;                            // recvA
  if (c->replicate) fields += (size_t)c->P;             // dk_full
  if (c->pipeline) fields += 6;                         // blockA2, recvA2
  if (c->general) fields += 1;                          // W
  size_t b = fields * field + nc * ((size_t)c->pb + sizeof(int) + 12 * (size_t)c->pb);
  b += (2 * PF_NBLK + PF_NBLK + SC_COUNT) * sizeof(double) + PF_NBINS * sizeof(unsigned long long) +
       (size_t)(PF_MAX_SMOOTH + 1) * (5 * PF_KNOT_CAP * sizeof(double) + PF_GT_DOUBLES * sizeof(double) + PF_GT_MAX_BINS * sizeof(unsigned short)) +
       4 * PF_KBIN_CAP * sizeof(double) + (size_t)c->n * sizeof(double) + (size_t)c->n * 2 * c->fb;
  *at_create = b;
  b += 6 * field;                                       // blockB2
  if (c->fb == 4 && !c->general && pf_c2r_invariants_supported(c->fb, c->n)) b += 2 * nc * sizeof(double);  // inv_own
  *peak = b;
}
extern "C" int pf_plan_bytes(const pf_config *cfg, size_t *at_create, size_t *peak) {
  if (!cfg || cfg->nranks < 1 || cfg->n < 4 || cfg->n % cfg->nranks || (cfg->field_bytes != 4 && cfg->field_bytes != 8)) return 1;
  PfTuning tune;
  read_tuning(&tune);
  pf_ctx v;   // (a view that is never created: geometry only)
  const int n = (int)cfg->n;
  const bool pow2 = !(n & (n - 1));
  v.n = n; v.P = cfg->nranks; v.fb = cfg->field_bytes; v.pb = (cfg->flags & PF_FLAG_DOUBLE_PRODUCTS) ? 8 : 4;
  v.general = tune.general || (!pow2 && !pf_mixed_supported(n));
  v.nzh = n / 2 + 1; v.nzp = v.general ? v.nzh : n / 2 + 64 / v.fb; v.nxl = v.nyl = n / v.P;
  v.field_bytes = (size_t)v.n * v.nyl * v.nzp * 2 * v.fb;
  v.pipeline = v.P > 1 && tune.pipeline;
  v.replicate = v.P > 1 && !v.general && (tune.replicate < 0 ? v.P <= 4 : tune.replicate != 0);
  size_t a = 0, b = 0;
  plan_bytes(&v, &a, &b);
  if (at_create) *at_create = a;
  if (peak) *peak = b;
  return 0;
}

static int create_body(pf_ctx *c, const pf_config *cfg) {
  const int rank = cfg->rank;
  HIPCHK(c, hipSetDevice(cfg->device));
  c->field_bytes = (size_t)c->n * c->nyl * c->nzp * 2 * c->fb;
  c->replicate = c->P > 1 && !c->general && (c->tune.replicate < 0 ? c->P <= 4 : c->tune.replicate != 0);
  if (c->tune.preflight) {  // the whole plan against the device's free memory, before anything is allocated (PF_PREFLIGHT=0: no check)
    size_t at_create = 0, peak = 0, free_b = 0, total_b = 0;
    plan_bytes(c, &at_create, &peak);
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    if (peak > free_b)
      return pf_fail(rank, "pf_create: %d^3 on %d ranks with %d-byte fields needs %.1f GB of device memory per rank (%.1f GB at creation, the rest with the "
                           "LPT part), device %d has %.1f GB free of %.1f GB", c->n, c->P, c->fb, 1e-9 * peak, 1e-9 * at_create, cfg->device, 1e-9 * free_b, 1e-9 * total_b);
  }
  HIPCHK(c, hipStreamCreate(&c->stream));
  HIPCHK(c, hipStreamCreate(&c->cstream));
  {  // the solve stream takes what the compute stream leaves: lowest priority
    int lo = 0, hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->sstream, hipStreamNonBlocking, lo));
  }
  for (int i = 0; i < 2; i++) {
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_y[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_s[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_x[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_r[i], hipEventDisableTiming));
  }
  {
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, cfg->device));
    // Default: the sincos / cbrt / exp10 forms of the solver's transcendental hot spots (pf_collapse_core.h), 25 %
    // fewer fp64 instructions.  PF_EXACT_LIBM=1: the reference's own calls (cos x3, pow, pow), bit-comparable with the CPU.
    c->fast_libm = !c->tune.exact_libm;
    c->ncu = prop.multiProcessorCount;
    c->prune_eps = c->tune.prune_eps;
    // four 256-thread workgroups of the solve fit per CU (128 VGPRs); 8 per CU in the grid evens out the tail
    c->collapse_blocks = prop.multiProcessorCount * c->tune.collapse_wg_per_cu;
    if (c->collapse_blocks > PF_NBLK) c->collapse_blocks = PF_NBLK;
  }
  PFCHK(c, dev_alloc(c, &c->dk, c->field_bytes));
  PFCHK(c, dev_alloc(c, (void **)&c->blockA, 3 * c->field_bytes));
  for (int i = 0; i < 3; i++) c->A[i] = c->blockA + i * c->field_bytes;
  for (int i = 0; i < 6; i++) PFCHK(c, dev_alloc(c, &c->B[i], c->field_bytes));
  PFCHK(c, dev_alloc(c, (void **)&c->blockS, 3 * c->field_bytes));
  for (int i = 0; i < 3; i++) c->S[i] = c->blockS + i * c->field_bytes;
  if (c->P > 1) PFCHK(c, dev_alloc(c, &c->recvA, 3 * c->field_bytes));
  // default: up to four ranks.  There the all-to-alls of the sweep cost more link time than its kernels take (one xGMI link
  // per peer pair: N^3 W / P^2 bytes per field and link), while the replicated x-pass costs each rank one more read of the
  // whole delta(k) per radius; from eight ranks on the transposes are small enough to hide behind the kernels
  if (c->replicate) PFCHK(c, dev_alloc(c, &c->dk_full, (size_t)c->P * c->field_bytes));
  if (c->pipeline) {
    PFCHK(c, dev_alloc(c, (void **)&c->blockA2, 3 * c->field_bytes));
    PFCHK(c, dev_alloc(c, &c->recvA2, 3 * c->field_bytes));
  }
  if (c->general) {
    PFCHK(c, dev_alloc(c, &c->W, c->field_bytes));
    const int rc = pf_gfft_create(c->n, c->stream, &c->fft_c2r, &c->fft_r2c);
    if (rc) return pf_fail(rank, "pf_create: the chirp-z plan of the general transform path for %d^3 failed (%d)", c->n, rc);
  }
  const size_t nc = ncell(c);
  PFCHK(c, dev_alloc(c, (void **)&c->fmax, nc * (size_t)c->pb));
  PFCHK(c, dev_alloc(c, (void **)&c->rmax, nc * sizeof(int)));
  PFCHK(c, dev_alloc(c, (void **)&c->vel12, 12 * nc * (size_t)c->pb));
  PFCHK(c, dev_alloc(c, (void **)&c->partials, 2 * PF_NBLK * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->partials_src, PF_NBLK * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->scal, SC_COUNT * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->hist, PF_NBINS * sizeof(unsigned long long)));
  PFCHK(c, dev_alloc(c, (void **)&c->spl, (size_t)(PF_MAX_SMOOTH + 1) * 5 * PF_KNOT_CAP * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->gtab, 4 * PF_KBIN_CAP * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->gt, (size_t)(PF_MAX_SMOOTH + 1) * PF_GT_DOUBLES * sizeof(double)));
  PFCHK(c, dev_alloc(c, (void **)&c->gt_lut, (size_t)(PF_MAX_SMOOTH + 1) * PF_GT_MAX_BINS * sizeof(unsigned short)));
  PFCHK(c, dev_alloc(c, (void **)&c->etab, (size_t)c->n * sizeof(double)));
  HIPCHK(c, hipMemsetAsync(c->scal, 0, SC_COUNT * sizeof(double), c->stream));
  {
    std::vector<char> h;
    host_twiddles(c->n, c->fb, h);
    PFCHK(c, dev_alloc(c, &c->tw, h.size()));
    HIPCHK(c, hipMemcpy(c->tw, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pf_create(pf_ctx **out, const pf_config *cfg) {
  if (!out || !cfg) return pf_fail(0, "pf_create: null argument");
  *out = nullptr;
  const int rank = cfg->rank;
  const long long n = cfg->n;
  const bool pow2 = !(n & (n - 1));
  PfTuning tune;
  read_tuning(&tune);
  // not a power of two: the run-time stage plans of pf_mixed_kernels.hip where they apply (n = 8 m, m = 2^a 3^b 5^c; slabs of
  // n / P planes over 1, 2, 4 or 8 ... ranks), library transforms (the "general path", one rank) for the rest
  const bool mixed = !pow2 && !tune.general && pf_mixed_supported((int)n);
  const bool want_general = (!pow2 && !mixed) || tune.general;
  if (want_general) {
    if (n < 4 || n > 2048 || (n & 1)) return pf_fail(rank, "pf_create: grid size %lld must be even, in [4, 2048]", n);
    if (cfg->nranks != 1 || cfg->field_bytes != 8)
      return pf_fail(rank, "pf_create: grid size %lld is not 8 * 2^a 3^b 5^c: the chirp-z transform path (any even size) takes one rank and fp64 fields; "
                           "slab decomposition and fp32 fields need n = 8 * 2^a 3^b 5^c in [16, 2048]", n);
  } else if (!mixed && (n < 16 || n > 2048)) return pf_fail(rank, "pf_create: grid size %lld must lie in [16, 2048] (n = 8 * 2^a 3^b 5^c on slabs, any even size on one rank)", n);
  if (cfg->nranks < 1 || n % cfg->nranks || cfg->rank < 0 || cfg->rank >= cfg->nranks)
    return pf_fail(rank, "pf_create: nranks %d must divide the grid size %lld (slab decomposition)", cfg->nranks, n);
  // a power-of-two grid splits by shifts and masks (PfAddr::el_shift): a power-of-two number of ranks.  The mixed-radix passes split a line over
  // the slabs by multiply-high (PfAddr::el_len), any slab thickness of two planes and more: any number of ranks that divides the grid
  // (round 6: 96^3 on 3, 120^3 on 6, 200^3 on 5 ... -- the reference takes any NTasks its slabs allow, src/fmax-pfft.c:95-111)
  if ((cfg->nranks & (cfg->nranks - 1)) && !mixed)
    return pf_fail(rank, "pf_create: nranks %d must be a power of two for a grid of %lld = 2^k points per side (any divisor of the grid size for n = 8 * 2^a 3^b 5^c that is not a power of two)", cfg->nranks, n);
  if (mixed && cfg->nranks > 1 && n / cfg->nranks < 2) return pf_fail(rank, "pf_create: slabs of one plane are not supported (grid %lld on %d ranks)", n, cfg->nranks);
  if (cfg->field_bytes != 8 && cfg->field_bytes != 4) return pf_fail(rank, "pf_create: field_bytes must be 8 or 4");
  if ((cfg->flags & PF_FLAG_DOUBLE_PRODUCTS) && cfg->field_bytes != 8)
    return pf_fail(rank, "pf_create: PF_FLAG_DOUBLE_PRODUCTS (fp64 Fmax and displacements) needs fp64 fields");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return pf_fail(rank, "pf_create: no HIP device available (libpinfmax_hip has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return pf_fail(rank, "pf_create: device %d out of range (%d devices)", cfg->device, ndev);
  pf_ctx *c = new pf_ctx();
  c->cfg = *cfg; c->rank = rank; c->P = cfg->nranks; c->n = (int)n; c->nzh = c->n / 2 + 1; c->nzp = c->n / 2 + 64 / cfg->field_bytes;  // rows of whole 128-byte lines: + 8 complex (fp64), + 16 (fp32)
  c->tune = tune; c->dev = cfg->device; c->inv_reruns = 0;
  c->general = want_general; c->fft_c2r = c->fft_r2c = nullptr; c->W = nullptr;
  if (c->general) c->nzp = c->nzh;  // natural layout [n][n][n/2+1], the boundary layout itself
  c->nxl = c->n / c->P; c->nyl = c->n / c->P; c->fb = cfg->field_bytes; c->timing = (cfg->flags & PF_FLAG_TIMING) != 0; c->pb = (cfg->flags & PF_FLAG_DOUBLE_PRODUCTS) ? 8 : 4;
  c->dev_bytes = 0; c->own_stream = true; c->stream = nullptr;
  c->a2a = nullptr; c->a2av = nullptr; c->a2av_user = nullptr; c->ared = nullptr; c->a2a_user = c->ared_user = nullptr; c->rccl = nullptr;
  c->have_density = c->have_hessian = c->have_sources = c->products_init = false; c->last_ns = 0;
  c->sweep_sources = c->sources_fresh = false; c->partials_src = nullptr; c->lpt_order = 3; c->ct_flavour = 0; c->transposed = false;
  c->vel_zero_pending = false; c->have_genic = false;
  memset(c->ks_ms, 0, sizeof(c->ks_ms)); memset(c->ks_bytes, 0, sizeof(c->ks_bytes)); memset(c->ks_n, 0, sizeof(c->ks_n));
  memset(&c->cpu, 0, sizeof(c->cpu)); memset(c->spl_set, 0, sizeof(c->spl_set)); memset(c->spl_n, 0, sizeof(c->spl_n));
  c->growth[0] = 1.0; c->growth[1] = 3. / 7.; c->growth[2] = -1. / 9.; c->growth[3] = 5. / 42.;
  for (int i = 0; i < 6; i++) { c->B[i] = nullptr; c->B2[i] = nullptr; }
  c->blockA = nullptr; c->dk = nullptr; c->recvA = nullptr; c->tw = nullptr;
  c->blockA2 = nullptr; c->recvA2 = nullptr; c->cstream = nullptr; c->dk_full = nullptr; c->replicate = false; c->dk_full_valid = false; for (int k = 0; k < 3; k++) c->INV[0][k] = c->INV[1][k] = nullptr; c->inv_w = 0; c->inv_own[0] = c->inv_own[1] = nullptr; c->blockS = c->blockB2 = nullptr;
  c->fmax = nullptr; c->rmax = nullptr; c->vel12 = nullptr; c->partials = nullptr; c->scal = nullptr; c->hist = nullptr; c->spl = nullptr;
  c->gtab = nullptr; c->etab = nullptr; c->ct_block = nullptr; c->gt = nullptr; c->gt_lut = nullptr;
  memset(c->gt_ok, 0, sizeof(c->gt_ok)); memset(c->gt_err, 0, sizeof(c->gt_err));
  for (int i = 0; i < 2; i++) c->ev_x[i] = c->ev_r[i] = c->ev_y[i] = c->ev_s[i] = nullptr;
  c->sstream = nullptr; c->solve_ran_beside = false; c->loopback = nullptr; c->handoff = nullptr;
  c->pipeline = c->P > 1 && tune.pipeline;
  for (int i = 0; i < 3; i++) { c->A[i] = nullptr; c->S[i] = nullptr; }
  memset(c->gt_n, 0, sizeof(c->gt_n));
  c->tab_ns = 0; c->tab_ready = false; c->tab_ismooth = -1; c->model = 0; c->sng_ns = 0; memset(c->sng_cosmo, 0, sizeof(c->sng_cosmo)); memset(c->sng_size, 0, sizeof(c->sng_size)); memset(&c->ct, 0, sizeof(c->ct));
  const int rc = create_body(c, cfg);
  if (rc) {  // nothing of a half-built context stays behind (13 fields of 8.7 GB each at 1024^3)
    std::string msg = g_err;
    pf_destroy(c);
    snprintf(g_err, sizeof(g_err), "%s", msg.c_str());
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" int pf_destroy(pf_ctx *c) {
  if (!c) return 0;
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->cstream) hipStreamSynchronize(c->cstream);
  if (c->sstream) hipStreamSynchronize(c->sstream);
  pf_rccl_release(c->rccl); c->rccl = nullptr;
  delete c->loopback; c->loopback = nullptr;
  handoff_release(c);
  hipFree(c->dk); hipFree(c->blockA); hipFree(c->recvA); hipFree(c->tw); hipFree(c->blockA2); hipFree(c->recvA2); hipFree(c->dk_full); hipFree(c->inv_own[0]); hipFree(c->inv_own[1]);
  for (int i = 0; i < 6; i++) hipFree(c->B[i]);
  hipFree(c->blockS); hipFree(c->blockB2);
  hipFree(c->fmax); hipFree(c->rmax); hipFree(c->vel12); hipFree(c->partials); hipFree(c->partials_src); hipFree(c->scal); hipFree(c->hist); hipFree(c->spl); hipFree(c->gt); hipFree(c->gt_lut); hipFree(c->gtab); hipFree(c->etab); hipFree(c->ct_block); hipFree(c->W);
  pf_gfft_destroy(c->fft_c2r);
  if (c->fft_r2c != c->fft_c2r) pf_gfft_destroy(c->fft_r2c);
  for (auto &e : c->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto &e : c->phase_evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto e : c->evpool) hipEventDestroy(e);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  if (c->cstream) hipStreamDestroy(c->cstream);
  if (c->sstream) hipStreamDestroy(c->sstream);
  for (int i = 0; i < 2; i++) { if (c->ev_x[i]) hipEventDestroy(c->ev_x[i]); if (c->ev_r[i]) hipEventDestroy(c->ev_r[i]); if (c->ev_y[i]) hipEventDestroy(c->ev_y[i]); if (c->ev_s[i]) hipEventDestroy(c->ev_s[i]); }
  delete c;
  return 0;
}

extern "C" int pf_set_stream(pf_ctx *c, void *stream) {
  if (!c) return 1;
  hipStreamSynchronize(c->stream);
  if (c->own_stream) hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)stream; c->own_stream = false;
  return 0;
}
extern "C" void *pf_get_stream(pf_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int pf_synchronize(pf_ctx *c) {
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->cstream)); HIPCHK(c, hipStreamSynchronize(c->sstream));
  return 0;
}
extern "C" size_t pf_device_bytes(pf_ctx *c) { return c ? c->dev_bytes : 0; }

extern "C" int pf_set_exchange(pf_ctx *c, pf_alltoall_fn fn, void *user) { c->a2a = fn; c->a2a_user = user; return 0; }
extern "C" int pf_set_exchange_rows(pf_ctx *c, pf_alltoallv_fn fn, void *user) { c->a2av = fn; c->a2av_user = user; return 0; }
extern "C" int pf_set_allreduce(pf_ctx *c, pf_allreduce_fn fn, void *user) { c->ared = fn; c->ared_user = user; return 0; }
// Measurement aid (pf_set_loopback_exchange): one rank of a P-rank decomposition on its own.  The all-to-all hands every block of
// the send buffer back to this rank -- copied for the first `copies` calls, so that the receive buffers hold finite, field-like
// numbers, then not at all -- and the all-reduce leaves the rank's own contribution.  The kernels of the rank run on the slab of
// the full-size box with the launch geometry of the real run (line lengths, tile counts, pitches); the RESULTS ARE NOT THOSE OF
// THE BOX.  What it gives is the compute time per rank of a configuration whose box does not fit one GPU (BASELINE config 5).
static int loopback_a2a(void *user, const void *send, void *recv, size_t bytes_per_peer, void *stream) {
  PfLoopback *lb = (PfLoopback *)user;
  if (lb->copies_left <= 0) return 0;
  lb->copies_left--;
  // (the whole field: P blocks of bytes_per_peer; the block from "peer" p is this rank's own block p)
  return hipMemcpyAsync(recv, send, bytes_per_peer * (size_t)lb->nranks, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess;
}
static int loopback_ared(void *, void *, size_t, int, void *) { return 0; }

extern "C" int pf_set_loopback_exchange(pf_ctx *c, int copies) {
  if (!c) return 1;
  // never on top of a real exchange: the results would silently be those of no box at all
  if (c->rccl || (c->a2a && c->a2a != loopback_a2a) || (c->ared && c->ared != loopback_ared))
    return pf_fail(c->rank, "pf_set_loopback_exchange: an exchange is already installed (pf_init_rccl / pf_set_exchange / fabric); release it first");
  if (!c->loopback) c->loopback = new PfLoopback;
  c->loopback->copies_left = copies; c->loopback->nranks = c->P;
  c->a2a = loopback_a2a; c->a2a_user = c->loopback; c->a2av = nullptr; c->a2av_user = nullptr;
  c->ared = loopback_ared; c->ared_user = nullptr;
  return 0;
}
extern "C" int pf_loopback_active(pf_ctx *c) { return c ? (c->a2a == loopback_a2a ? 1 : 0) : -1; }
extern "C" int pf_exchange_buffers(pf_ctx *c, void **sendbuf, void **recvbuf, size_t *bytes) {
  if (sendbuf) *sendbuf = c->blockA;
  if (recvbuf) *recvbuf = c->recvA;
  if (bytes) *bytes = 3 * c->field_bytes;
  return 0;
}

// all-to-all of one field (KY block q <-> XS block p); a no-op on one rank
static int exchange(pf_ctx *c, const void *send, void *recv, hipStream_t st = nullptr) {
  if (c->P == 1) return 0;
  if (!c->a2a) return pf_fail(c->rank, "no exchange installed for %d ranks (pf_set_exchange / pf_init_rccl)", c->P);
  if (!st) st = c->stream;
  KTimer t(c, KS_EXCHANGE, (double)c->field_bytes, st);
  if (c->a2a(c->a2a_user, send, recv, c->field_bytes / c->P, (void *)st)) return pf_fail(c->rank, "all-to-all failed");
  return 0;
}
// The same all-to-all for a band-limited (Gaussian-smoothed) spectrum: of every block only the slab rows whose ky lies
// in the band carry data (the x-pass writes nothing else, the y-pass reads nothing else).  The blocks of the inverse
// transforms are laid out [q][y_local][x_local][nzp], so the in-band rows of a rank are ONE contiguous piece at the
// same offset in each of its P send blocks: rank p owns ky = p*nyl .. (p+1)*nyl - 1, in band if ky <= band or
// ky >= n - band (one interval per rank once P >= 2, band < n/2).  Ranks without in-band rows send nothing.
// ... and of those rows only the in-band kz columns: the blocks of a pruned item use the compact row pitch below
// (a multiple of 128 bytes: 8 complex fp64, 16 fp32) instead of nzp, in the x-pass output, the exchange and the y-pass input alike
static int band_zpitch(const pf_ctx *c, int band) {
  if (band >= c->n / 2) return c->nzp;
  const int m = 64 / c->fb;  // complex numbers per 128-byte line
  const int zp = (band + 1 + m - 1) & ~(m - 1);
  return zp < c->nzp ? zp : c->nzp;
}
static void band_rows(const pf_ctx *c, int p, int band, int *lo, int *hi) {
  const int y0 = p * c->nyl, y1 = y0 + c->nyl;  // global ky range of rank p
  *lo = *hi = 0;
  if (y0 <= band) { *lo = 0; *hi = (band + 1 < y1 ? band + 1 : y1) - y0; }
  else if (y1 > c->n - band) { *lo = (c->n - band > y0 ? c->n - band : y0) - y0; *hi = c->nyl; }
  // an odd number of ranks (mixed-radix grids, round 6): the middle slab straddles n / 2 and can hold rows of BOTH ends of the band.  One
  // piece per rank is what the exchange moves: the rows between the two ends travel with them (never written by the x-pass, never read
  // by the y-pass: stale bytes that nobody looks at)
  if (y0 <= band && y1 > c->n - band) { *lo = 0; *hi = c->nyl; }
}
static int exchange_band(pf_ctx *c, const void *send, void *recv, int band, hipStream_t st = nullptr) {
  if (c->P == 1) return 0;
  if (band >= c->n / 2 || !c->a2av || !c->tune.exchange_rows) return exchange(c, send, recv, st);  // PF_EXCHANGE_ROWS=0: always whole blocks
  if (!st) st = c->stream;
  const size_t row_bytes = (size_t)c->nxl * band_zpitch(c, band) * 2 * c->fb, block_bytes = c->field_bytes / c->P;
  std::vector<size_t> roff(c->P), rbytes(c->P);
  size_t total = 0;
  for (int p = 0; p < c->P; p++) {
    int lo, hi;
    band_rows(c, p, band, &lo, &hi);
    roff[p] = (size_t)lo * row_bytes; rbytes[p] = (size_t)(hi - lo) * row_bytes;
    total += rbytes[p];
  }
  KTimer t(c, KS_EXCHANGE, (double)total, st);
  if (c->a2av(c->a2av_user, send, recv, block_bytes, roff[c->rank], rbytes[c->rank], roff.data(), rbytes.data(), (void *)st))
    return pf_fail(c->rank, "all-to-all (row range) failed");
  return 0;
}
static int allreduce_dev(pf_ctx *c, void *buf, size_t count, int is_u64) {
  if (c->P == 1) return 0;
  if (!c->ared) return pf_fail(c->rank, "no all-reduce installed for %d ranks (pf_set_allreduce / pf_init_rccl)", c->P);
  if (c->ared(c->ared_user, buf, count, is_u64, (void *)c->stream)) return pf_fail(c->rank, "all-reduce failed");
  return 0;
}


// ---------------------------------------------------------------- hand-off --
static void handoff_release(pf_ctx *c) {
  PfHandoff *h = c->handoff;
  if (!h) return;
  for (int i = 0; i < 2; i++) { if (h->st[i]) { hipStreamSynchronize(h->st[i]); hipStreamDestroy(h->st[i]); } if (h->pin[i]) hipHostFree(h->pin[i]); }
  if (h->ev) hipEventDestroy(h->ev);
  if (h->reg_ptr) (void)hipHostUnregister(h->reg_ptr);
  delete h->pool;
  delete h;
  c->handoff = nullptr;
}
static int handoff_get(pf_ctx *c, PfHandoff **out) {
  if (!c->handoff) {
    PfHandoff *h = new PfHandoff();
    h->st[0] = h->st[1] = nullptr; h->pin[0] = h->pin[1] = nullptr; h->ev = nullptr; h->pool = nullptr; h->reg_ptr = nullptr; h->reg_bytes = 0;
    c->handoff = h;
    size_t chunk = (size_t)c->tune.handoff_chunk_mb << 20;
    if (chunk > c->field_bytes) chunk = c->field_bytes;  // (a piece is staged in ONE of the two staging fields on the device)
    h->chunk = chunk & ~(size_t)4095;
    if (h->chunk < 4096) h->chunk = c->field_bytes;
    int nt = c->tune.handoff_threads;
    if (nt <= 0) { nt = pf_host_cores(); if (nt > 16) nt = 16; }
    h->want_register = c->tune.host_register;
    for (int i = 0; i < 2; i++) {
      HIPCHK(c, hipStreamCreateWithFlags(&h->st[i], hipStreamNonBlocking));
      HIPCHK(c, hipHostMalloc((void **)&h->pin[i], h->chunk, hipHostMallocDefault));
    }
    HIPCHK(c, hipEventCreateWithFlags(&h->ev, hipEventDisableTiming));
    h->pool = new PfHostPool(nt);
  }
  *out = c->handoff;
  return 0;
}
// the two hand-off streams start behind everything the compute stream holds
static int handoff_begin(pf_ctx *c, PfHandoff *h) {
  HIPCHK(c, hipEventRecord(h->ev, c->stream));
  for (int i = 0; i < 2; i++) HIPCHK(c, hipStreamWaitEvent(h->st[i], h->ev, 0));
  return 0;
}
// ... and the compute stream goes on behind them (both are idle when a hand-off returns: it has synchronised them)
static char *handoff_dev(pf_ctx *c, int b) { return (char *)c->A[1 + b]; }
// `bytes` from the device to the caller's pageable memory; fill(piece_offset, piece_bytes, device_staging, stream) puts a piece in place
// on the device first (null: src_dev is copied as it lies)
template <class Fill>
static int handoff_d2h(pf_ctx *c, char *host, const char *src_dev, size_t bytes, size_t granule, Fill fill) {
  PfHandoff *h;
  if (handoff_get(c, &h)) return 1;
  PFCHK(c, handoff_begin(c, h));
  const size_t piece = (h->chunk / granule) * granule;
  if (!piece) return pf_fail(c->rank, "hand-off: a record of %zu bytes does not fit the staging pieces", granule);
  const size_t np = (bytes + piece - 1) / piece;
  auto issue = [&](size_t k) -> int {
    const int b = (int)(k & 1);
    const size_t off = k * piece, len = bytes - off < piece ? bytes - off : piece;
    const char *src = src_dev ? src_dev + off : handoff_dev(c, b);
    if (fill(off, len, handoff_dev(c, b), h->st[b])) return 1;
    HIPCHK(c, hipMemcpyAsync(h->pin[b], src, len, hipMemcpyDeviceToHost, h->st[b]));
    return 0;
  };
  if (np && issue(0)) return 1;
  for (size_t k = 0; k < np; k++) {
    if (k + 1 < np && issue(k + 1)) return 1;
    const int b = (int)(k & 1);
    HIPCHK(c, hipStreamSynchronize(h->st[b]));
    const size_t off = k * piece, len = bytes - off < piece ? bytes - off : piece;
    char *dst = host + off; const char *src = h->pin[b];
    h->pool->run(len, [=](size_t a, size_t e) { memcpy(dst + a, src + a, e - a); });
  }
  return 0;
}
static int no_fill(size_t, size_t, char *, hipStream_t) { return 0; }
// `bytes` from the caller's pageable memory to dst_dev: the host threads fill one pinned buffer while the DMA empties the other
static int handoff_h2d(pf_ctx *c, char *dst_dev, const char *host, size_t bytes) {
  PfHandoff *h;
  if (handoff_get(c, &h)) return 1;
  PFCHK(c, handoff_begin(c, h));
  const size_t piece = h->chunk, np = (bytes + piece - 1) / piece;
  for (size_t k = 0; k < np; k++) {
    const int b = (int)(k & 1);
    const size_t off = k * piece, len = bytes - off < piece ? bytes - off : piece;
    HIPCHK(c, hipStreamSynchronize(h->st[b]));  // the DMA that last read this buffer (piece k - 2)
    char *dst = h->pin[b]; const char *src = host + off;
    h->pool->run(len, [=](size_t a, size_t e) { memcpy(dst + a, src + a, e - a); });
    HIPCHK(c, hipMemcpyAsync(dst_dev + off, h->pin[b], len, hipMemcpyHostToDevice, h->st[b]));
  }
  for (int i = 0; i < 2; i++) HIPCHK(c, hipStreamSynchronize(h->st[i]));
  return 0;
}
// the caller's product array, registered with the driver once (PF_HOST_REGISTER=1): true when the DMA may write into it directly
static bool handoff_registered(pf_ctx *c, PfHandoff *h, void *host, size_t bytes) {
  if (!h->want_register || ((uintptr_t)host & 4095)) return false;
  if (h->reg_ptr == host && h->reg_bytes >= bytes) return true;
  if (h->reg_ptr) { (void)hipHostUnregister(h->reg_ptr); h->reg_ptr = nullptr; h->reg_bytes = 0; }
  if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
  h->reg_ptr = host; h->reg_bytes = bytes;
  return true;
}

// ---------------------------------------------------------- pass helpers ----
static int ilog2i(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static PfAddr addr_ky_x(const pf_ctx *c) {  // KY layout, e = x, outer = y_local
  PfAddr a; a.os = c->nzp; a.el_shift = ilog2i(c->n); a.el_len = c->n; a.ehs = 0; a.els = (long long)c->nyl * c->nzp; return a;
}
static PfAddr addr_blocks_y(const pf_ctx *c) {  // P blocks [p][nxl][nyl][nzp], e = y = p*nyl + yl, outer = x_local
  PfAddr a; a.os = (long long)c->nyl * c->nzp; a.el_shift = ilog2i(c->nyl); a.el_len = c->nyl; a.ehs = (long long)c->nxl * c->nyl * c->nzp; a.els = c->nzp; return a;
}
// inverse transforms: the x-pass outputs are laid out with the slab row slowest, [q][y_local][x_local][nzp] -- the send /
// receive blocks of a multi-rank run, and for one rank simply [y][x][nzp]: the x-pass then writes its 1024 segments of a
// tile at a stride of one row (8.3 KB) instead of one x-plane (8.5 MB = 65 * 2^17 bytes, an address pattern the memory
// channels do not like: 7.7 -> 6.1 ms per Hessian x-pass at 1024^3), the y-pass reads at the plane stride instead
// (unchanged within the box-to-box spread).
// zp: row pitch inside the blocks (nzp, or the compact pitch of a band-limited item); the blocks keep their places
static PfAddr addr_yblocks_x(const pf_ctx *c, int zp) {  // x-pass output: e = x = q*nxl + xl, outer = y_local
  PfAddr a; a.os = (long long)c->nxl * zp; a.el_shift = ilog2i(c->nxl); a.el_len = c->nxl; a.ehs = (long long)c->nyl * c->nxl * c->nzp; a.els = zp; return a;
}
static PfAddr addr_yblocks_y(const pf_ctx *c, int zp) {  // y-pass input: e = y = p*nyl + yl, outer = x_local
  PfAddr a; a.os = zp; a.el_shift = ilog2i(c->nyl); a.el_len = c->nyl; a.ehs = (long long)c->nyl * c->nxl * c->nzp; a.els = (long long)c->nxl * zp; return a;
}
// replicated delta(k): the x-pass reads the whole spectrum [kx][ky][kz] (outer = global ky) ...
static PfAddr addr_full_x(const pf_ctx *c) {
  PfAddr a; a.os = c->nzp; a.el_shift = ilog2i(c->n); a.el_len = c->n; a.ehs = 0; a.els = (long long)c->n * c->nzp; return a;
}
// ... and writes this rank's x-slab of every line as [ky][x_local][zp]: what the y-pass then reads in place (e = ky)
static PfAddr addr_local_x(const pf_ctx *c, int zp) {
  PfAddr a; a.os = (long long)c->nxl * zp; a.el_shift = ilog2i(c->nxl); a.el_len = c->nxl; a.ehs = 0; a.els = zp; return a;
}
static PfAddr addr_local_y(const pf_ctx *c, int zp) {
  PfAddr a; a.os = zp; a.el_shift = ilog2i(c->n); a.el_len = c->n; a.ehs = 0; a.els = (long long)c->nxl * zp; return a;
}
static PfAddr addr_xs_y(const pf_ctx *c) {  // XS layout, e = y, outer = x_local
  PfAddr a; a.os = (long long)c->n * c->nzp; a.el_shift = ilog2i(c->n); a.el_len = c->n; a.ehs = 0; a.els = c->nzp; return a;
}

struct Job { const void *in; void *out; int mul; };

static int xpass(pf_ctx *c, int kind, int dir, int njobs, const Job *jobs, int pre, double rs, double growth, int nin, int band = 1 << 30,
                 bool out_yblocks = false, bool replicated = false) {
  PfStridedParams p; memset(&p, 0, sizeof(p));
  p.njobs = njobs;
  for (int j = 0; j < njobs; j++) { p.job[j].in = jobs[j].in; p.job[j].out = jobs[j].out; p.job[j].mul = jobs[j].mul; }
  p.ain = p.aout = addr_ky_x(c);
  if (out_yblocks) p.aout = addr_yblocks_x(c, band_zpitch(c, band));
  p.ncols = c->nzh; p.nouter = c->nyl; p.pre = pre; p.outer_offset = c->rank * c->nyl; p.rs = rs; p.growth = growth; p.tw = c->tw;
  p.etab = c->etab; p.dev = c->dev;
  if (replicated) {  // the input is the whole spectrum: every ky line, only this rank's x-slab of the result
    p.ain = addr_full_x(c); p.aout = addr_local_x(c, band_zpitch(c, band));
    p.nouter = c->n; p.outer_offset = 0; p.out_e0 = c->rank * c->nxl; p.out_ne = c->nxl;
  }
  if (pre && rs != 0.0) PFCHK(c, pf_launch_exp_table(c->etab, c->n, rs, c->stream));  // stream order: after the previous x-pass
  p.band_e = p.band_outer = c->n;
  double frac_cols = 1.0, frac_in = 1.0, frac_outer = 1.0;
  if (band < c->n / 2) {  // pruned: in-band columns (kz, ky) only, in-band x read, all x written
    p.band_e = p.band_outer = band;
    p.ncols = band + 1;
    frac_cols = (double)(band + 1) / c->nzh; frac_in = frac_outer = (double)(2 * band + 1) / c->n;
  }
  // (replicated: the whole spectrum is read, P slabs; one slab per output is written)
  KTimer t(c, kind, (nin * frac_in * (replicated ? c->P : 1) + njobs) * frac_cols * frac_outer * spec_bytes_alg(c));
  PFCHK(c, pf_launch_strided(c->fb, c->n, dir, p, c->stream));
  return 0;
}
// in_blocks: input is the P received blocks (after an all-to-all) else XS; out_blocks likewise (forward direction)
static int ypass(pf_ctx *c, int kind, int dir, int njobs, const Job *jobs, bool in_blocks, bool out_blocks, int nin, int band = 1 << 30,
                 bool in_yblocks = false, bool in_local = false) {
  PfStridedParams p; memset(&p, 0, sizeof(p));
  p.njobs = njobs;
  for (int j = 0; j < njobs; j++) { p.job[j].in = jobs[j].in; p.job[j].out = jobs[j].out; p.job[j].mul = jobs[j].mul; }
  p.ain = in_blocks ? (in_yblocks ? addr_yblocks_y(c, band_zpitch(c, band)) : addr_blocks_y(c)) : addr_xs_y(c);
  if (in_local) p.ain = addr_local_y(c, band_zpitch(c, band));  // written in place by a replicated x-pass
  p.aout = out_blocks ? addr_blocks_y(c) : addr_xs_y(c);
  p.ncols = c->nzh; p.nouter = c->nxl; p.pre = 0; p.outer_offset = 0; p.rs = 0; p.growth = 1; p.tw = c->tw; p.dev = c->dev;
  p.band_e = p.band_outer = c->n;
  double frac_cols = 1.0, frac_in = 1.0;
  if (band < c->n / 2) {  // pruned: kz columns in band, in-band ky read, all y written, every x
    p.band_e = band;
    p.ncols = band + 1;
    frac_cols = (double)(band + 1) / c->nzh; frac_in = (double)(2 * band + 1) / c->n;
  }
  KTimer t(c, kind, (nin * frac_in + njobs) * frac_cols * spec_bytes_alg(c));
  PFCHK(c, pf_launch_strided(c->fb, c->n, dir, p, c->stream));
  return 0;
}
struct ZJob { const void *in; void *out; int mul; int f32; };
static int zpass_c2r(pf_ctx *c, int kind, int njobs, const ZJob *jobs, const double *dc, int band = 1 << 30, bool invariants = false,
                     void *acc = nullptr) {
  PfC2RParams p; memset(&p, 0, sizeof(p));
  p.njobs = njobs;
  double outb = 0;
  for (int j = 0; j < njobs; j++) {
    p.job[j].in = jobs[j].in; p.job[j].out = jobs[j].out; p.job[j].mul = jobs[j].mul; p.job[j].out_f32 = jobs[j].f32;
    outb += jobs[j].f32 ? (double)ncell(c) * 4.0 : real_bytes_alg(c);
  }
  p.nlines = (long long)c->nxl * c->n; p.in_pitch = c->nzp; p.out_pitch = 2 * c->nzp;
  p.norm = 1.0 / ((double)c->n * c->n * c->n); p.dc = dc; p.tw = c->tw;
  p.band_k = c->n;
  double frac_in = 1.0;
  if (band < c->n / 2) { p.band_k = band; frac_in = (double)(band + 1) / c->nzh; }
  p.ncu = c->ncu; p.dev = c->dev; p.persist_per_cu = c->tune.zpass_persist; p.inv_per_cu = c->tune.zpass_inv_wg_per_cu;
  p.flag = c->scal + SC_INV_FLAG;
  if (acc) {  // six components in, none out: acc -= 2 phi2_ab h_ab with the first-order Hessian h in jobs[].out (src/LPT.c:134-137)
    p.acc = acc;
    KTimer t(c, KS_ZPASS_LPT3B, njobs * frac_in * spec_bytes_alg(c) + 8.0 * real_bytes_alg(c));
    PFCHK(c, pf_launch_c2r_invariants(c->fb, c->n, p, c->stream, 1));
    return 0;
  }
  if (invariants) {  // six components in, the three invariants of the tensor out (fields 0..2; fp32 fields: fp64 rows in INV)
    if (c->fb == 4) {
      if (inv_rows(c, c->inv_w)) return 1;
      for (int k = 0; k < 3; k++) p.inv_out[k] = c->INV[c->inv_w][k];
      p.inv_pitch = c->n;
      if (c->inv_w == 0) c->have_sources = false;  // (row 0 of set 0 lies in S[0..1]: LPT spectra resident there are overwritten)
    }
    KTimer t(c, KS_ZPASS_INV, njobs * frac_in * spec_bytes_alg(c) + 3.0 * (double)ncell(c) * 8.0);
    PFCHK(c, pf_launch_c2r_invariants(c->fb, c->n, p, c->stream));
    return 0;
  }
  KTimer t(c, kind, njobs * frac_in * spec_bytes_alg(c) + outb);
  PFCHK(c, pf_launch_c2r(c->fb, c->n, p, c->stream));
  return 0;
}

// ---- software pipeline over transforms that share the exchange buffers ----
// pre(i, A) fills the send fields A[0..nf-1] (x-pass side, compute stream); the all-to-all moves field f to
// dst(i, set, f); post(i, R) consumes R[0..nf-1] (compute stream).  P = 1: R = A, no exchange.  P > 1 with the
// pipeline on: two buffer sets; the exchange of item i+1 is enqueued on the communication stream before post(i), so
// the wire time hides behind the y/z passes and the collapse solve of item i ("overlap transposes of
// field i with passes of field i+-1").  Hazards: pre(i) reuses the send set of item i-2, whose exchange post(i-2)
// waited for; exchange(i) overwrites the receive set of item i-2 after ev_x(i), recorded behind post(i-2).
// band(i): the spectrum of item i is band-limited to |k| <= band (>= n/2: not): only the in-band slab rows travel
// (fault injection of the tests only: holds the compute stream back so that an exchange that does not wait starts too early)
__global__ void k_debug_idle(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
template <class Pre, class Dst, class Post, class Band>
static int pipelined_band(pf_ctx *c, int count, int nf, Pre pre, Dst dst, Post post, Band band, bool local = false) {
  void *A[2][3];
  const void *R[3];
  if (local) {  // replicated spectrum: pre writes this rank's slab straight into what post reads -- nothing travels
    for (int i = 0; i < count; i++) {
      void *D[3];
      for (int f = 0; f < nf; f++) { D[f] = dst(i, 0, f); R[f] = D[f]; }
      if (pre(i, D)) return 1;
      if (post(i, R)) return 1;
    }
    return 0;
  }
  for (int f = 0; f < 3; f++) { A[0][f] = c->A[f]; A[1][f] = c->pipeline ? (void *)(c->blockA2 + f * c->field_bytes) : c->A[f]; }
  if (c->P == 1 || !c->pipeline) {
    for (int i = 0; i < count; i++) {
      if (pre(i, A[0])) return 1;
      for (int f = 0; f < nf; f++) {
        R[f] = A[0][f];
        if (c->P > 1) { void *d = dst(i, 0, f); PFCHK(c, exchange_band(c, A[0][f], d, band(i))); R[f] = d; }
      }
      if (post(i, R)) return 1;
    }
    return 0;
  }
  auto issue = [&](int i) -> int {
    const int s = i & 1;
    if (c->tune.debug_fault == 2) hipLaunchKernelGGL(k_debug_idle, dim3(1), dim3(1), 0, c->stream, 200000LL);  // 2 ms at 100 MHz
    if (pre(i, A[s])) return 1;
    HIPCHK(c, hipEventRecord(c->ev_x[s], c->stream));
    if (c->tune.debug_fault != 2) HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_x[s], 0));
    for (int f = 0; f < nf; f++) PFCHK(c, exchange_band(c, A[s][f], dst(i, s, f), band(i), c->cstream));
    HIPCHK(c, hipEventRecord(c->ev_r[s], c->cstream));
    return 0;
  };
  if (issue(0)) return 1;
  for (int i = 0; i < count; i++) {
    if (i + 1 < count && issue(i + 1)) return 1;
    if (c->tune.debug_fault != 1) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_r[i & 1], 0));
    for (int f = 0; f < nf; f++) R[f] = dst(i, i & 1, f);
    if (post(i, R)) return 1;
  }
  return 0;
}
template <class Pre, class Dst, class Post>
static int pipelined(pf_ctx *c, int count, int nf, Pre pre, Dst dst, Post post) {
  return pipelined_band(c, count, nf, pre, dst, post, [](int) { return 1 << 30; });
}
static void *recv_field(pf_ctx *c, int set, int f) { return (char *)(set ? c->recvA2 : c->recvA) + (size_t)f * c->field_bytes; }

// ---------------------------------------------------------------------------------------------- general path ----
// Grid sizes that are not a power of two (one rank, fp64): the reference's own structure -- one k-space filter and one
// 3-D c2r per component (compute_derivative, src/fmax-pfft.c:255-441) -- with the chirp-z transforms of pf_gfft.hip on the natural layouts
// (spectrum [n][n][n/2+1] = the boundary layout, real fields [n][n][n]).  Per-cell kernels are the same as in the
// fused path.  Unfused, so it moves ~2x the bytes of the power-of-two path; it exists for completeness of the drop-in.
static int g_filter(pf_ctx *c, const void *in, void *out, int a, int b, double rs, int order, bool normalise) {
  double growth = order ? c->growth[order - 1] : 1.0;
  const int o = order - 1;
  const bool tab = order && c->gt_n[o];
  KTimer t(c, KS_MISC, 2.0 * spec_bytes_alg(c));
  PFCHK(c, pf_launch_gen_filter(in, out, c->n, a, b, rs, growth, tab ? c->gtab + o * PF_KBIN_CAP : nullptr, tab ? c->gt_n[o] : 0,
                                tab ? c->gt_logkmin[o] : 0.0, tab ? c->gt_dlogk[o] : 1.0, tab ? c->gt_sign[o] : 1.0,
                                normalise ? 1.0 / ((double)c->n * c->n * c->n) : 1.0, c->stream));
  return 0;
}
static int g_c2r(pf_ctx *c, void *spec, void *real) {
  KTimer t(c, KS_ZPASS_PLAIN, spec_bytes_alg(c) + real_bytes_alg(c));
  if (pf_gfft_c2r(c->fft_c2r, spec, real)) return pf_fail(c->rank, "the c2r transform of the general path failed");
  return 0;
}
static int g_r2c(pf_ctx *c, void *real, void *spec) {
  KTimer t(c, KS_R2C_Z, spec_bytes_alg(c) + real_bytes_alg(c));
  if (pf_gfft_r2c(c->fft_r2c, real, spec)) return pf_fail(c->rank, "the r2c transform of the general path failed");
  return 0;
}
static int g_hessian_of(pf_ctx *c, const void *spec, double rs, void *const out[6]) {
  static const int pa[6] = {1, 2, 3, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};  // storage order 11,22,33,12,13,23 (src/LPT.c:36-44)
  for (int i = 0; i < 6; i++) {
    PFCHK(c, g_filter(c, spec, c->W, pa[i], pb[i], rs, 0, true));
    PFCHK(c, g_c2r(c, c->W, out[i]));
  }
  return 0;
}
// column k (0..11 = 3 * order + axis) of the SoA displacements
static void *velcol(const pf_ctx *c, int k) { return (char *)c->vel12 + (size_t)k * ncell(c) * (size_t)c->pb; }

static int g_displacements_of(pf_ctx *c, int count, const void *const *specs, const int *orders, void *tmp) {
  const size_t nc = ncell(c);
  for (int j = 0; j < count; j++)
    for (int ia = 1; ia <= 3; ia++) {
      PFCHK(c, g_filter(c, specs[j], c->W, ia, 0, 0.0, orders[j] + 1, true));
      PFCHK(c, g_c2r(c, c->W, tmp));
      PFCHK(c, pf_launch_real_to_col((const double *)tmp, velcol(c, 3 * orders[j] + ia - 1), nc, c->pb, c->stream));
    }
  return 0;
}
// real field in f (pitch n) -> its unnormalised spectrum in f
static int g_forward_of(pf_ctx *c, void *f) {
  PFCHK(c, g_r2c(c, f, c->W));
  HIPCHK(c, hipMemcpyAsync(f, c->W, c->field_bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
// spectrum in f -> c2r / N^3 in f
static int g_reverse_of(pf_ctx *c, void *f) {
  PFCHK(c, g_filter(c, f, c->W, -1, -1, 0.0, 0, true));
  return g_c2r(c, c->W, f);
}

// six second derivatives of `spec` (KY layout) at smoothing rs -> six real fields out[0..5] (R layout)
// order 11,22,33,12,13,23 (src/LPT.c:36-44); compute_second_derivatives, src/fmax.c:225-258
// Gaussian window exp(-k^2 rs^2/2) < prune_eps (2^-60) beyond |k| = sqrt(-2 ln eps)/rs: those modes are dropped
// (pruned FFT).  Their total contribution is < 2^-56 of the unsmoothed rms: below the rounding of the transform.
static int hess_band(const pf_ctx *c, double rs) {
  int band = 1 << 30;
  if (rs > 0.0 && c->prune_eps > 0.0) {
    const double kc = sqrt(-2.0 * log(c->prune_eps)) / rs;
    const double kb = kc * c->n / (2.0 * 3.14159265358979323846);
    if (kb < c->n / 2 - 1) band = (int)kb + 1;
  }
  return band;
}
static int hess_x(pf_ctx *c, const void *spec, double rs, void *const A[3], int band, bool replicated = false) {
  const Job xj[3] = {{spec, A[0], PF_MUL_ONE}, {spec, A[1], PF_MUL_K}, {spec, A[2], PF_MUL_K2}};
  return xpass(c, KS_XPASS_HESS, +1, 3, xj, 1, rs, 1.0, 1, band, true, replicated);
}
// gathers delta(k) of all ranks into dk_full, once per density: this rank's ky-slab goes to its place in a zeroed array
// and an integer all-reduce adds the other ranks' zeros to it (exact for either field type; ~2 x 8.6 GB per rank at 1024^3)
static int ensure_dk_full(pf_ctx *c) {
  if (!c->replicate || c->dk_full_valid) return 0;
  if (c->a2a == loopback_a2a && c->have_genic) {
    // One rank on its own (pf_set_loopback_exchange): no peer contributes its slab.  A density that came from pf_genic_density is a
    // function of (seed, cosmology) alone, every column on its own: the rank generates all P ky-slabs itself -- the KY layout of ONE
    // rank of the whole box is the layout of dk_full -- and the sweep that follows runs on the box's own delta(k), exchange-free.
    // (Any other density behind the loopback: the rank's own slab among zeros, below -- a timing aid whose numbers are no box's.)
    unsigned int *dseed = nullptr;
    const int rc = pf_genic_launch(c->fb, c->dk_full, c->n, c->nzp, c->n, 0, &c->genic, c->stream, &dseed);
    if (!rc) HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(dseed);
    if (rc) return pf_fail(c->rank, "pf_genic_density (whole spectrum): launch failed");
    c->dk_full_valid = true;
    return 0;
  }
  const size_t rows = (size_t)c->nyl * c->nzp * 2 * c->fb;  // the nyl rows this rank holds of one kx plane
  HIPCHK(c, hipMemsetAsync(c->dk_full, 0, (size_t)c->P * c->field_bytes, c->stream));
  HIPCHK(c, hipMemcpy2DAsync((char *)c->dk_full + (size_t)c->rank * rows, (size_t)c->P * rows, c->dk, rows, rows, (size_t)c->n,
                             hipMemcpyDeviceToDevice, c->stream));
  PFCHK(c, allreduce_dev(c, c->dk_full, (size_t)c->P * c->field_bytes / 8, 1));
  c->dk_full_valid = true;
  return 0;
}
static int hess_yz(pf_ctx *c, const void *const R[3], const double *dc, void *const out[6], int band, bool invariants = false,
                   void *acc = nullptr, void *const *hfirst = nullptr, bool in_local = false, const std::function<int()> *between = nullptr) {
  const Job yj[6] = {{R[2], out[0], PF_MUL_ONE}, {R[1], out[3], PF_MUL_K}, {R[1], out[4], PF_MUL_ONE},
                     {R[0], out[1], PF_MUL_K2}, {R[0], out[5], PF_MUL_K},  {R[0], out[2], PF_MUL_ONE}};
  PFCHK(c, ypass(c, KS_YPASS_HESS, +1, 6, yj, true, false, 3, band, true, in_local));
  if (between && (*between)()) return 1;  // (the solve stream of a sweep: what is to run beside the z-pass is enqueued here)
  const ZJob zj[6] = {{out[0], out[0], PF_MUL_ONE, 0}, {out[1], out[1], PF_MUL_ONE, 0}, {out[2], out[2], PF_MUL_K2, 0},
                      {out[3], out[3], PF_MUL_ONE, 0}, {out[4], out[4], PF_MUL_K, 0},   {out[5], out[5], PF_MUL_K, 0}};
  if (acc) {  // the z-pass contracts the six rows with the first-order Hessian `hfirst` into `acc` and stores nothing else
    ZJob cj[6];
    for (int i = 0; i < 6; i++) { cj[i] = zj[i]; cj[i].out = hfirst[i]; }
    PFCHK(c, zpass_c2r(c, KS_ZPASS_HESS, 6, cj, dc, band, false, acc));
    return 0;
  }
  PFCHK(c, zpass_c2r(c, KS_ZPASS_HESS, 6, zj, dc, band, invariants));
  return 0;
}
static int hessian_of(pf_ctx *c, const void *spec, double rs, const double *dc, void *const out[6],
                      void *acc = nullptr, void *const *hfirst = nullptr) {
  if (c->general) return g_hessian_of(c, spec, rs, out);
  const int band = hess_band(c, rs);
  const bool rep = c->replicate && spec == c->dk;  // the LPT source spectra stay distributed: their transposes travel
  if (rep) PFCHK(c, ensure_dk_full(c));
  return pipelined_band(c, 1, 3,
                        [&](int, void *const *A) { return hess_x(c, rep ? c->dk_full : spec, rs, A, band, rep); },
                        [&](int, int set, int f) { return recv_field(c, set, f); },
                        [&](int, const void *const *R) { return hess_yz(c, R, dc, out, band, false, acc, hfirst, rep); },
                        [&](int) { return band; }, rep);
}

// three first derivatives (displacement components) of specs[j] times growths[j] -> vel12[3*orders[j] .. +2]
// compute_first_derivatives + write_from_rvector_to_products, src/fmax.c:193-222, src/fmax-pfft.c:563-631
// The growth multiplier of ScaleDep.order = orders[j] + 1 is a scalar (c->growth) or, with a k-binned table
// installed (pf_set_growth_table), applied per mode by k_apply_growth into the spare send field first.
static int displacements_of(pf_ctx *c, int count, const void *const *specs, const int *orders, void *const tmp[3]) {
  if (c->general) return g_displacements_of(c, count, specs, orders, tmp[0]);
  // items that start from delta(k) itself (the Zel'dovich term) use the replicated spectrum when there is one and no
  // per-mode growth table has to be applied to it first: nothing travels for them.  They run after the others.
  auto run = [&](int cnt, const void *const *sp, const int *ord, bool rep) -> int {
    return pipelined_band(c, cnt, 2,
                   [&](int j, void *const *A) {
                     const int o = ord[j];
                     const void *spec = rep ? c->dk_full : sp[j];
                     double growth = c->growth[o];
                     if (c->gt_n[o]) {
                       KTimer t(c, KS_MISC, 2.0 * spec_bytes_alg(c));
                       PFCHK(c, pf_launch_apply_growth(c->fb, spec, A[2], c->n, c->nyl, c->nzh, c->nzp, c->rank * c->nyl, c->gtab + o * PF_KBIN_CAP,
                                                       c->gt_n[o], c->gt_logkmin[o], c->gt_dlogk[o], c->gt_sign[o], c->stream));
                       spec = A[2];
                       growth = 1.0;
                     }
                     const Job xj[2] = {{spec, A[0], PF_MUL_ONE}, {spec, A[1], PF_MUL_IK}};
                     return xpass(c, KS_XPASS_DISP, +1, 2, xj, 1, 0.0, growth, 1, 1 << 30, true, rep);
                   },
                   [&](int, int set, int f) { return recv_field(c, set, f); },
                   [&](int j, const void *const *R) {
                     const int o = ord[j];
                     const Job yj[3] = {{R[1], tmp[0], PF_MUL_ONE}, {R[0], tmp[1], PF_MUL_IK}, {R[0], tmp[2], PF_MUL_ONE}};
                     PFCHK(c, ypass(c, KS_YPASS_DISP, +1, 3, yj, true, false, 2, 1 << 30, true, rep));
                     const int compact = c->pb == 8 ? 2 : 1;  // rows of pitch n: float, or the fields' own double
                     const ZJob zj[3] = {{tmp[0], velcol(c, 3 * o + 0), PF_MUL_ONE, compact},
                                         {tmp[1], velcol(c, 3 * o + 1), PF_MUL_ONE, compact},
                                         {tmp[2], velcol(c, 3 * o + 2), PF_MUL_IK, compact}};
                     PFCHK(c, zpass_c2r(c, KS_ZPASS_DISP, 3, zj, nullptr));
                     return 0;
                   },
                   [](int) { return 1 << 30; }, rep);
  };
  const void *far[4], *near_[4];
  int far_o[4], near_o[4], nfar = 0, nnear = 0;
  for (int j = 0; j < count; j++) {
    if (c->replicate && specs[j] == c->dk && !c->gt_n[orders[j]]) { near_[nnear] = specs[j]; near_o[nnear++] = orders[j]; }
    else { far[nfar] = specs[j]; far_o[nfar++] = orders[j]; }
  }
  if (nfar && run(nfar, far, far_o, false)) return 1;
  if (nnear) {
    PFCHK(c, ensure_dk_full(c));
    if (run(nnear, near_, near_o, true)) return 1;
  }
  return 0;
}

// unnormalised r2c of the real fields fs[i] (R layout) -> spectra in place (KY layout)
// forward_transform, src/fmax-pfft.c:191-200
static int forward_r2c(pf_ctx *c, void *f) {
  PfR2CParams p; p.in = f; p.out = f; p.nlines = (long long)c->nxl * c->n; p.in_pitch = 2 * c->nzp; p.out_pitch = c->nzp; p.tw = c->tw;
  KTimer t(c, KS_R2C_Z, real_bytes_alg(c) + spec_bytes_alg(c));
  PFCHK(c, pf_launch_r2c(c->fb, c->n, p, c->stream));
  return 0;
}
static int forward_many(pf_ctx *c, int count, void *const *fs) {
  if (c->general) {
    for (int i = 0; i < count; i++) PFCHK(c, g_forward_of(c, fs[i]));
    return 0;
  }
  if (c->P == 1) {  // y-pass in place, nothing to exchange
    for (int i = 0; i < count; i++) {
      void *f = fs[i];
      PFCHK(c, forward_r2c(c, f));
      const Job yj[1] = {{f, f, PF_MUL_ONE}};
      PFCHK(c, ypass(c, KS_YPASS_FWD, -1, 1, yj, false, false, 1));
      const Job xj[1] = {{f, f, PF_MUL_ONE}};
      PFCHK(c, xpass(c, KS_XPASS_FWD, -1, 1, xj, 0, 0.0, 1.0, 1));
    }
    return 0;
  }
  return pipelined(c, count, 1,
                   [&](int i, void *const *A) {
                     PFCHK(c, forward_r2c(c, fs[i]));
                     const Job yj[1] = {{fs[i], A[0], PF_MUL_ONE}};
                     return ypass(c, KS_YPASS_FWD, -1, 1, yj, false, true, 1);
                   },
                   [&](int i, int, int) { return fs[i]; },
                   [&](int i, const void *const *) {
                     const Job xj[1] = {{fs[i], fs[i], PF_MUL_ONE}};
                     return xpass(c, KS_XPASS_FWD, -1, 1, xj, 0, 0.0, 1.0, 1);
                   });
}
static int forward_of(pf_ctx *c, void *f) { return forward_many(c, 1, &f); }

// plain c2r of the spectrum in `f` (KY) -> real in `f` (R) times 1/N^3, in place (reverse_transform)
static int reverse_of(pf_ctx *c, void *f) {
  if (c->general) return g_reverse_of(c, f);
  const Job xj[1] = {{f, f, PF_MUL_ONE}};
  PFCHK(c, xpass(c, KS_XPASS_PLAIN, +1, 1, xj, 0, 0.0, 1.0, 1));
  const void *R = f;
  if (c->P > 1) { PFCHK(c, exchange(c, f, c->recvA)); R = c->recvA; }
  const Job yj[1] = {{R, f, PF_MUL_ONE}};
  PFCHK(c, ypass(c, KS_YPASS_PLAIN, +1, 1, yj, true, false, 1));
  const ZJob zj[1] = {{f, f, PF_MUL_ONE, 0}};
  PFCHK(c, zpass_c2r(c, KS_ZPASS_PLAIN, 1, zj, nullptr));
  return 0;
}

static char *staging(pf_ctx *c) { return (char *)c->A[1]; }  // 2 fields of scratch for fp64 host<->device traffic

// boundary layout (this rank's x-slab [nxl][n][nzh] fp64 complex, host) <-> KY layout on the device.  With P > 1 the
// regrouping is one all-to-all; scratch: the staging fields A[1..2] and receive fields 1, 2 (never A[0], receive
// field 0 or the Hessian buffers, so the transform taps can run between sweep and displacements).
static int import_spec(pf_ctx *c, const double *host, void *dst) {
  const long long nrows = (long long)c->nxl * c->n;
  if (handoff_h2d(c, staging(c), (const char *)host, (size_t)nrows * c->nzh * 2 * sizeof(double))) return 1;
  if (c->transposed) {  // the caller's slab is a ky-slab already: a local swap of the two leading indices, nothing to exchange
    HIPCHK(c, hipMemsetAsync(dst, 0, c->field_bytes, c->stream));
    PFCHK(c, pf_launch_spec_import_t(c->fb, (const double *)staging(c), dst, c->n, c->nyl, c->nzh, c->nzp, c->stream));
    return 0;
  }
  if (c->P == 1) {
    HIPCHK(c, hipMemsetAsync(dst, 0, c->field_bytes, c->stream));
    PFCHK(c, pf_launch_spec_import(c->fb, (const double *)staging(c), dst, nrows, c->nzh, c->nzp, c->stream));
    return 0;
  }
  void *rows = recv_field(c, 0, 1), *blocks = recv_field(c, 0, 2);
  HIPCHK(c, hipMemsetAsync(rows, 0, c->field_bytes, c->stream));
  PFCHK(c, pf_launch_spec_import(c->fb, (const double *)staging(c), rows, nrows, c->nzh, c->nzp, c->stream));
  PFCHK(c, pf_launch_to_blocks(c->fb, rows, blocks, c->nxl, c->n, c->nyl, c->nzp, 0, c->stream));
  PFCHK(c, exchange(c, blocks, dst));
  return 0;
}
static int export_spec(pf_ctx *c, const void *spec, double *host) {
  const long long nrows = (long long)c->nxl * c->n;
  const void *rows = spec;
  if (c->transposed) {
    PFCHK(c, pf_launch_spec_export_t(c->fb, spec, (double *)staging(c), c->n, c->nyl, c->nzh, c->nzp, c->stream));
    return handoff_d2h(c, (char *)host, staging(c), (size_t)nrows * c->nzh * 2 * sizeof(double), 1, no_fill);
  }
  if (c->P > 1) {  // KY has x slowest: the block for the owner of an x-slab is contiguous, no pack
    void *blocks = recv_field(c, 0, 1), *r = recv_field(c, 0, 2);
    PFCHK(c, exchange(c, spec, blocks));
    PFCHK(c, pf_launch_to_blocks(c->fb, blocks, r, c->nxl, c->n, c->nyl, c->nzp, 1, c->stream));
    rows = r;
  }
  PFCHK(c, pf_launch_spec_export(c->fb, rows, (double *)staging(c), nrows, c->nzh, c->nzp, c->stream));
  return handoff_d2h(c, (char *)host, staging(c), (size_t)nrows * c->nzh * 2 * sizeof(double), 1, no_fill);
}
// Re of the k = 0 mode / N^3 on every rank: the k-filter leaves that mode untouched (k^2 = 0, src/fmax-pfft.c:368), so a
// second derivative carries it as a constant; it lives on the rank that owns x = 0
static int dc_of_host_spec(pf_ctx *c, const double *host, int slot) {
  const double dcv = (c->rank == 0) ? host[0] / ((double)c->n * c->n * c->n) : 0.0;
  HIPCHK(c, hipMemcpyAsync(c->scal + slot, &dcv, sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  PFCHK(c, allreduce_dev(c, c->scal + slot, 1, 0));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------ inputs --
extern "C" int pf_set_density(pf_ctx *c, const double *kd) {
  if (!c || !kd) return pf_fail(0, "pf_set_density: null argument");
  PhaseTimer pt(c, 3);
  // boundary layout: this rank's x-slab [nxl][n][nzh] (non-transposed PFFT output, src/fmax-pfft.c:366)
  PFCHK(c, import_spec(c, kd, c->dk));
  PFCHK(c, dc_of_host_spec(c, kd, SC_DC_DK));
  c->have_genic = false;
  c->have_density = true; c->have_hessian = false; c->have_sources = false; c->sources_fresh = false; c->dk_full_valid = false;
  return 0;
}

extern "C" int pf_synth_density(pf_ctx *c, uint64_t seed, double sigma0, double slope) {
  if (!c) return 1;
  // white noise (R layout) -> r2c -> shape -> normalise
  if (c->general) {  // real noise in a real field, library r2c into the spectrum
    PFCHK(c, pf_launch_white(c->fb, c->B[0], (long long)c->n * c->n, 0, c->n, c->n, seed, c->stream));
    PFCHK(c, g_r2c(c, c->B[0], c->dk));
  } else {
    PFCHK(c, pf_launch_white(c->fb, c->dk, (long long)c->nxl * c->n, (long long)c->rank * c->nxl * c->n, c->n, 2 * c->nzp, seed, c->stream));
    PFCHK(c, forward_of(c, c->dk));
  }
  PfShapeParams p; memset(&p, 0, sizeof(p));
  p.spec = c->dk; p.n = c->n; p.nzp = c->nzp; p.nyl = c->nyl; p.y0 = c->rank * c->nyl; p.slope = slope; p.scale = 1.0;
  p.partials = c->partials; p.nblocks = PF_NBLK; p.mode = 0; p.dscale = nullptr;
  PFCHK(c, pf_launch_shape(c->fb, p, c->stream));
  PFCHK(c, pf_launch_sum1(c->partials, PF_NBLK, 1.0, c->scal + SC_POWER, c->stream));
  PFCHK(c, allreduce_dev(c, c->scal + SC_POWER, 1, 0));
  PFCHK(c, pf_launch_sigma_scale(c->scal + SC_POWER, sigma0, (double)c->n * c->n * c->n, c->scal + SC_DSCALE, c->stream));
  p.mode = 1; p.dscale = c->scal + SC_DSCALE;
  PFCHK(c, pf_launch_shape(c->fb, p, c->stream));
  HIPCHK(c, hipMemsetAsync(c->scal + SC_DC_DK, 0, sizeof(double), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_genic = false;
  c->have_density = true; c->have_hessian = false; c->have_sources = false; c->sources_fresh = false; c->dk_full_valid = false;
  return 0;
}

extern "C" int pf_genic_density(pf_ctx *c, const pf_genic_params *p) {
  if (!c || !p) return pf_fail(0, "pf_genic_density: null argument");
  if (!(p->BoxSize_true_Mpc > 0) || !(p->PkNorm > 0) || !(p->Omega0 > 0) || !(p->Hubble100 > 0))
    return pf_fail(c->rank, "pf_genic_density: BoxSize, PkNorm, Omega0 and Hubble100 must be positive");
  unsigned int *dseed = nullptr;
  if (pf_genic_launch(c->fb, c->dk, c->n, c->nzp, c->nyl, c->rank * c->nyl, p, c->stream, &dseed)) {
    hipFree(dseed);
    return pf_fail(c->rank, "pf_genic_density: launch failed");
  }
  HIPCHK(c, hipMemsetAsync(c->scal + SC_DC_DK, 0, sizeof(double), c->stream));  // the DC mode is left at zero
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(dseed);
  c->genic = *p;
  if (p->pk_n > 0 && p->pk_logk && p->pk_logk3p) {
    c->genic_logk.assign(p->pk_logk, p->pk_logk + p->pk_n); c->genic_logk3p.assign(p->pk_logk3p, p->pk_logk3p + p->pk_n);
    c->genic.pk_logk = c->genic_logk.data(); c->genic.pk_logk3p = c->genic_logk3p.data();
  }
  c->have_genic = true;
  c->have_density = true; c->have_hessian = false; c->have_sources = false; c->sources_fresh = false; c->dk_full_valid = false;
  return 0;
}

extern "C" int pf_set_invgrow(pf_ctx *c, int ismooth, const double *x, const double *y, int n) {
  if (!c || !x || !y) return pf_fail(0, "pf_set_invgrow: null argument");
  if (n < 3 || n > PF_KNOT_CAP) return pf_fail(c->rank, "pf_set_invgrow: %d knots not in [3, %d]", n, PF_KNOT_CAP);
  if (ismooth < -1 || ismooth >= PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_set_invgrow: ismooth %d out of range", ismooth);
  for (int i = 1; i < n; i++)
    if (!(x[i] > x[i - 1])) return pf_fail(c->rank, "pf_set_invgrow: knots must be strictly increasing (i=%d)", i);
  std::vector<double> h(5 * PF_KNOT_CAP, 0.0);
  memcpy(&h[0], x, n * sizeof(double));
  memcpy(&h[PF_KNOT_CAP], y, n * sizeof(double));
  if (pf_spline_coeffs(x, y, n, &h[2 * PF_KNOT_CAP])) return pf_fail(c->rank, "pf_set_invgrow: spline set-up failed");
  pf_spline_bd(x, y, &h[2 * PF_KNOT_CAP], n, &h[3 * PF_KNOT_CAP], &h[4 * PF_KNOT_CAP]);
  const int slot = ismooth + 1;  // slot 0 = shared spline
  HIPCHK(c, hipMemcpy(c->spl + (size_t)slot * 5 * PF_KNOT_CAP, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  c->spl_n[slot] = n; c->spl_set[slot] = true;
  {  // the polynomial table of the fast flavour (refused for knots it cannot serve: the solve then takes the series forms)
    std::vector<double> g(PF_GT_DOUBLES);
    std::vector<unsigned short> lut(PF_GT_MAX_BINS);
    g[5] = NAN;  // (a build that is refused before its accuracy check -- too many or too dense knots -- leaves no error figure: NaN, not 0)
    const bool ok = pf_gtab_build(&h[0], &h[PF_KNOT_CAP], &h[2 * PF_KNOT_CAP], &h[3 * PF_KNOT_CAP], &h[4 * PF_KNOT_CAP], n, g.data(), lut.data()) == 0;
    c->gt_ok[slot] = ok; c->gt_err[slot] = ok || g[5] > 0.0 ? g[5] : NAN;
    if (ok) {
      HIPCHK(c, hipMemcpy(c->gt + (size_t)slot * PF_GT_DOUBLES, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->gt_lut + (size_t)slot * PF_GT_MAX_BINS, lut.data(), lut.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
  }
  return 0;
}
// 1: the spline of `ismooth` (-1: the shared one) is served by its polynomial table in the fast flavour; *max_rel_err: the
// table's largest relative error against the long double composite, found when it was built
extern "C" int pf_invgrow_table_status(pf_ctx *c, int ismooth, double *max_rel_err) {
  if (!c || ismooth < -1 || ismooth >= PF_MAX_SMOOTH) return -1;
  const int slot = (ismooth >= 0 && c->spl_set[ismooth + 1]) ? ismooth + 1 : 0;
  if (max_rel_err) *max_rel_err = c->gt_err[slot];
  return c->spl_set[slot] && c->gt_ok[slot] && c->tune.gtab && c->fast_libm ? 1 : 0;
}
static int spline_for(pf_ctx *c, int ismooth, PfSplineDev *s) {
  int slot = (ismooth >= 0 && ismooth < PF_MAX_SMOOTH && c->spl_set[ismooth + 1]) ? ismooth + 1 : 0;
  if (!c->spl_set[slot]) return pf_fail(c->rank, "inverse-growth spline not set (pf_set_invgrow)");
  const double *t = c->spl + (size_t)slot * 5 * PF_KNOT_CAP;
  s->x = t; s->y = t + PF_KNOT_CAP; s->c = t + 2 * PF_KNOT_CAP; s->b = t + 3 * PF_KNOT_CAP; s->d = t + 4 * PF_KNOT_CAP;
  s->n = c->spl_n[slot];
  const bool gt = c->gt_ok[slot] && c->tune.gtab;
  s->gt = gt ? c->gt + (size_t)slot * PF_GT_DOUBLES : nullptr;
  s->gt_lut = gt ? c->gt_lut + (size_t)slot * PF_GT_MAX_BINS : nullptr;
  return 0;
}

extern "C" int pf_set_growth(pf_ctx *c, const double g[4]) {
  if (!c || !g) return 1;
  memcpy(c->growth, g, 4 * sizeof(double));
  return 0;
}

extern "C" int pf_set_growth_table(pf_ctx *c, int order, const double *log10_growth, int nk, double logkmin, double dlogk, double sign) {
  if (!c || order < 1 || order > 4) return pf_fail(0, "pf_set_growth_table: order must be 1..4");
  if (nk < 0 || nk == 1 || nk > PF_KBIN_CAP || (nk && !log10_growth)) return pf_fail(c->rank, "pf_set_growth_table: %d k-bins not in {0, 2..%d}", nk, PF_KBIN_CAP);
  const int o = order - 1;
  if (nk) HIPCHK(c, hipMemcpyAsync(c->gtab + o * PF_KBIN_CAP, log10_growth, nk * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gt_n[o] = nk; c->gt_logkmin[o] = logkmin; c->gt_dlogk[o] = dlogk; c->gt_sign[o] = sign;
  return 0;
}

// ---- TABULATED_CT (src/collapse_times.c:780-1231) ----
static int ct_alloc(pf_ctx *c) {
  if (c->ct_block) return 0;
  const size_t nd = PF_CT_NBINS_D, nt = (size_t)PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY;
  PFCHK(c, dev_alloc(c, (void **)&c->ct_block, (3 * nd + 4 * nt) * sizeof(double)));
  double h[3 * PF_CT_NBINS_D];
  memset(h, 0, sizeof(h));
  pf_ct_delta_vector(h);
  pf_ct_tridiag(h, PF_CT_NBINS_D, h + nd, h + 2 * nd);
  HIPCHK(c, hipMemcpy(c->ct_block, h, sizeof(h), hipMemcpyHostToDevice));
  c->ct.delta = c->ct_block; c->ct.alpha = c->ct_block + nd; c->ct.gamma = c->ct_block + 2 * nd;
  c->ct.y = c->ct_block + 3 * nd; c->ct.b = c->ct.y + nt; c->ct.c = c->ct.b + nt; c->ct.d = c->ct.c + nt;
  return 0;
}
// initialize_collapse_times(ismooth, .): table of ell() for this radius (or the caller's table) + the node splines
static int ct_build(pf_ctx *c, int ismooth, double variance, const double *table_host, hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  if (!(variance > 0.0)) return pf_fail(c->rank, "collapse-time table: Smoothing.Variance[%d] = %g must be positive", ismooth, variance);
  PFCHK(c, ct_alloc(c));
  c->ct.ampl = sqrt(variance);
  c->ct.model = c->model; c->ct.flavour = c->ct_flavour;
  if (c->model == 1 && !table_host) {
    if (ismooth >= c->sng_ns) return pf_fail(c->rank, "collapse-time table: no ELL_SNG growth factor for radius %d (pf_set_collapse_model)", ismooth);
    memcpy(c->ct.sng_cosmo, c->sng_cosmo, sizeof(c->sng_cosmo));
    c->ct.sng_cosmo[6] = c->sng_size[ismooth];
    c->ct.sng_Din = c->sng_Din[ismooth];
  }
  PfSplineDev sp; memset(&sp, 0, sizeof(sp));
  if (table_host) HIPCHK(c, hipMemcpyAsync(c->ct.y, table_host, (size_t)PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY * sizeof(double), hipMemcpyHostToDevice, st));
  else if (c->model == 0 && spline_for(c, ismooth, &sp)) return 1;
  {
    KTimer t(c, KS_MISC, 0.0, st);
    PFCHK(c, pf_launch_ct_build(sp, c->ct, c->fast_libm ? 1 : 0, table_host ? 0 : 1, st));
  }
  c->tab_ready = true; c->tab_ismooth = ismooth;
  return 0;
}
extern "C" int pf_set_collapse_model(pf_ctx *c, int model, const double cosmo[4], int nsmooth, const double *D_in) {
  if (!c) return 1;
  if (model != 0 && model != 1) return pf_fail(c->rank, "pf_set_collapse_model: model %d (0 ELL_CLASSIC, 1 ELL_SNG)", model);
  if (model == 1) {
    if (!cosmo || !D_in || nsmooth < 1 || nsmooth > PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_set_collapse_model: ELL_SNG needs the cosmology and D_in per radius");
    memcpy(c->sng_cosmo, cosmo, 4 * sizeof(double));
    memcpy(c->sng_Din, D_in, nsmooth * sizeof(double));
    c->sng_ns = nsmooth;
  }
  c->model = model; c->tab_ready = false;
  return 0;
}
extern "C" int pf_set_modified_gravity(pf_ctx *c, double fr0, double h_over_c, int nsmooth, const double *size) {
  if (!c) return 1;
  if (nsmooth < 0 || nsmooth > PF_MAX_SMOOTH || (fr0 != 0.0 && (!size || nsmooth < 1 || !(h_over_c > 0.0))))
    return pf_fail(c->rank, "pf_set_modified_gravity: f(R) needs H_over_c and the smoothing radius of every table");
  c->sng_cosmo[4] = fr0; c->sng_cosmo[5] = h_over_c;
  if (nsmooth && size) memcpy(c->sng_size, size, nsmooth * sizeof(double));
  c->tab_ready = false;
  return 0;
}
extern "C" int pf_set_tabulated_ct(pf_ctx *c, int nsmooth, const double *variance) {
  if (!c) return 1;
  if (nsmooth < 0 || nsmooth > PF_MAX_SMOOTH || (nsmooth && !variance)) return pf_fail(c->rank, "pf_set_tabulated_ct: bad argument");
  c->tab_ns = nsmooth; c->tab_ready = false;
  if (nsmooth) memcpy(c->tab_var, variance, nsmooth * sizeof(double));
  return 0;
}
extern "C" int pf_ct_build(pf_ctx *c, int ismooth, double variance, double *table_host) {
  if (!c) return 1;
  if (ismooth < 0 || ismooth >= PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_ct_build: ismooth %d out of range", ismooth);
  PFCHK(c, ct_build(c, ismooth, variance, nullptr));
  if (table_host) HIPCHK(c, hipMemcpyAsync(table_host, c->ct.y, (size_t)PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (ismooth < PF_MAX_SMOOTH && c->tab_ns <= ismooth) c->tab_ns = ismooth + 1;
  c->tab_var[ismooth] = variance;
  return 0;
}
extern "C" int pf_ct_load(pf_ctx *c, int ismooth, double variance, const double *table_host) {
  if (!c || !table_host) return 1;
  if (ismooth < 0 || ismooth >= PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_ct_load: ismooth %d out of range", ismooth);
  PFCHK(c, ct_build(c, ismooth, variance, table_host));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->tab_ns <= ismooth) c->tab_ns = ismooth + 1;
  c->tab_var[ismooth] = variance;
  return 0;
}

// ---------------------------------------------------------------- the path --
extern "C" int pf_second_derivatives(pf_ctx *c, double rs) {
  if (!c) return 1;
  if (!c->have_density) return pf_fail(c->rank, "pf_second_derivatives: density not set");
  PhaseTimer pt(c, 0);
  PFCHK(c, hessian_of(c, c->dk, rs, c->scal + SC_DC_DK, c->B));
  c->have_hessian = true; c->sources_fresh = false;
  return 0;
}

static int collapse_enqueue(pf_ctx *c, int ismooth, void *const H[6], hipStream_t st, bool build_table = true, bool invariants = false,
                            bool sources = false, int inv_set = 0) {
  PfCollapseParams p; memset(&p, 0, sizeof(p));
  for (int i = 0; i < 6; i++) p.h[i] = H[i];
  p.pitch = rpitch(c); p.nrows = (long long)c->nxl * c->n; p.n = c->n; p.fmax = c->fmax; p.prod_f64 = c->pb == 8; p.rmax = c->rmax; p.ismooth = ismooth;
  if (spline_for(c, ismooth, &p.spline)) return 1;
  p.partials = c->partials; p.fast = c->fast_libm ? 1 : 0;
  p.no_lut = c->tune.spline_lut ? 0 : 1;
  if (c->model == 1 && c->tab_ns == 0) {  // ELL_SNG without TABULATED_CT: every cell integrates its own ellipsoid
    if (invariants) return pf_fail(c->rank, "ELL_SNG per cell reads the six components");
    if (ismooth >= c->sng_ns) return pf_fail(c->rank, "ELL_SNG: no growth factor for radius %d (pf_set_collapse_model)", ismooth);
    memcpy(p.ct.sng_cosmo, c->sng_cosmo, sizeof(c->sng_cosmo));
    p.ct.sng_cosmo[6] = c->sng_size[ismooth];
    p.ct.sng_Din = c->sng_Din[ismooth];
    p.sng = 1;
  }
  if (c->tab_ns > 0) {  // TABULATED_CT build: the table of this radius is made right before its pass (src/fmax.c:103-106)
    if (build_table) {
      if (ismooth >= c->tab_ns) return pf_fail(c->rank, "collapse-time table: no variance for radius %d (pf_set_tabulated_ct)", ismooth);
      if (ct_build(c, ismooth, c->tab_var[ismooth], nullptr, st)) return 1;
    } else if (!c->tab_ready)
      return pf_fail(c->rank, "collapse-time table not built (pf_ct_build / pf_ct_load)");
    p.tabulated = 1; p.ct = c->ct; p.ct.flavour = c->ct_flavour;
  }
  sources = sources && c->lpt_order >= 2 && !invariants && !c->tab_ns && !p.sng;
  // (with the sources formed in passing the grid is k_lpt_sources' own -- clamped to PF_NBLK, whatever PF_COLLAPSE_WG_PER_CU or
  //  the CU count say -- so that the partial sums of S2 are added in the same order as by pf_displacements(1, 0) on its own)
  size_t nb = (ncell(c) + 255) / 256;
  const size_t cap = sources ? (size_t)PF_NBLK : (size_t)c->collapse_blocks;
  if (nb > cap) nb = cap;
  p.nblocks = (int)nb;
  if (invariants) p.invariants = 1;
  int solve_fb = c->fb;
  if (invariants && c->fb == 4) {  // the invariants of fp32 fields are fp64 rows of their own (k_c2r_invariants<float>)
    for (int k = 0; k < 3; k++) p.h[k] = c->INV[inv_set][k];
    p.pitch = c->n; solve_fb = 8;
  }
  if (sources) {  // K7 in the same pass (the grid is k_lpt_sources' own: identical partial sums of S2)
    p.sources = 1; p.src[0] = c->S[0]; p.src[1] = c->S[1]; p.src[2] = c->S[2]; p.src_partials = c->partials_src;
    c->have_sources = false;  // S now holds real-space sources, not the resident LPT spectra
  }
  {
    KTimer t(c, sources ? KS_COLLAPSE_SRC : invariants ? KS_COLLAPSE_INV : KS_COLLAPSE,
             (double)ncell(c) * ((invariants ? 3.0 * 8.0 / c->fb : sources ? 9.0 : 6.0) * c->fb + 16.0), st);
    PFCHK(c, pf_launch_collapse(solve_fb, p, st));
  }
  PFCHK(c, pf_launch_final_sum(c->partials, p.nblocks, c->scal + SC_VAR0 + 2 * ismooth, st));
  if (sources) {
    PFCHK(c, pf_launch_sum1(c->partials_src, p.nblocks, 1.0 / ((double)c->n * c->n * c->n), c->scal + SC_DC_S2, st));
    c->sources_fresh = true;
  }
  return 0;
}

// compute_collapse_times at ismooth = 0 resets the products (src/collapse_times.c:461-492: Fmax = -10, Rmax = -1, every
// Vel = 0).  The collapse kernel of the first radius writes Fmax/Rmax of every cell itself; the 48 bytes per cell of
// velocities are cleared only if somebody reads them before pf_displacements has rewritten all twelve columns.
static int products_reset(pf_ctx *c) {
  c->vel_zero_pending = true;
  c->products_init = true;
  return 0;
}
static int velocities_ready(pf_ctx *c) {
  if (c->vel_zero_pending) {
    HIPCHK(c, hipMemsetAsync(c->vel12, 0, 12 * ncell(c) * (size_t)c->pb, c->stream));
    c->vel_zero_pending = false;
  }
  return 0;
}

extern "C" int pf_collapse_times(pf_ctx *c, int ismooth, double *tv) {
  if (!c) return 1;
  if (!c->have_hessian) return pf_fail(c->rank, "pf_collapse_times: second derivatives not computed");
  if (ismooth < 0 || ismooth >= PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_collapse_times: ismooth %d out of range", ismooth);
  if (ismooth == 0) {  // src/collapse_times.c:461-492
    PFCHK(c, products_reset(c));
  } else if (!c->products_init)
    return pf_fail(c->rank, "pf_collapse_times: products not initialised (ismooth 0 must come first)");
  {
    PhaseTimer pt(c, 1);
    // a table already in place for this radius (pf_ct_build / pf_ct_load, the reference's initialize_collapse_times) is used as it is
    if (collapse_enqueue(c, ismooth, c->B, c->stream, !(c->tab_ready && c->tab_ismooth == ismooth))) return 1;
  }
  PFCHK(c, allreduce_dev(c, c->scal + SC_VAR0 + 2 * ismooth, 2, 0));
  double s[2];
  HIPCHK(c, hipMemcpyAsync(s, c->scal + SC_VAR0 + 2 * ismooth, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tv) *tv = s[1] / ((double)c->n * c->n * c->n);  // src/collapse_times.c:662,670
  return 0;
}

static int sweep_body_run(pf_ctx *c, int ns, const double *radius_cells, double *true_variance, bool six_components);
// Whatever way the body returns, the context is left in order: the fp32 invariant set index back at 0, and after a failure
// nothing still running on the solve stream that later entry points (which share c->partials, fmax, B / B2 with it) could race with
static int sweep_body(pf_ctx *c, int ns, const double *radius_cells, double *true_variance, bool six_components) {
  const int rc = sweep_body_run(c, ns, radius_cells, true_variance, six_components);
  c->inv_w = 0;
  if (rc && rc != 2) {
    std::string msg = g_err;  // (the synchronisation below must not disturb the message of the failure)
    hipStreamSynchronize(c->sstream);
    (void)hipGetLastError();
    snprintf(g_err, sizeof(g_err), "%s", msg.c_str());
  }
  return rc;
}
static int sweep_body_run(pf_ctx *c, int ns, const double *radius_cells, double *true_variance, bool six_components) {
  PhaseTimer ft(c, 4);
  PFCHK(c, products_reset(c));
  c->sources_fresh = false;
  // (n <= 1024: the six-line workgroup of a 2048-point row would need 110 KB of dynamic LDS, a size this build never launches)
  // (n <= 1024 with fp64 fields, n <= 2048 with fp32 ones: the six lines of a row must fit the LDS of a workgroup)
  const bool invariants_ok = (c->tune.invariants >= 2 ? pf_c2r_invariants_supported(c->fb, (int)c->n) : pf_c2r_invariants_preferred(c->fb, (int)c->n)) && !c->general && c->tab_ns == 0 && c->model == 0 && c->tune.invariants && !six_components;
  HIPCHK(c, hipMemsetAsync(c->scal + SC_INV_FLAG, 0, sizeof(double), c->stream));
  const bool rep = c->replicate;
  if (rep) PFCHK(c, ensure_dk_full(c));
  // The solve stream (PF_SOLVE_BESIDE_Z=0 turns it off; fp64 fields, invariant radii).  The solve is bound by fp64 issue, the
  // z-pass leaves half of the issue slots free and one 256-thread workgroup of the solve fits a CU beside its two workgroups:
  // the solve of radius i is enqueued on the solve stream (lowest priority) behind the y-pass of radius i + 1 and so starts with
  // that z-pass; what is left of it when the z-pass is done has the chip to itself (the strided passes that follow need whole
  // CUs).  The passes of consecutive radii alternate between the field sets B and B2 (B2: the six fields the LPT part needs
  // anyway), so that y(i + 1) does not overwrite the invariants solve(i) reads.  Same kernels, same grids, same order of the
  // running maximum: results are bit for bit those of the in-line order.  792 against 806 ms per step at 1024^3 (three boxes);
  // a partition of the chip by CUs instead (the solve in 1024-thread workgroups holding 96 .. 144 CUs, the passes of the next
  // radius on the others) gained nothing: 806 .. 874 ms, the passes lose what the solve wins (profiles/r03_experiments.md).
  // Round 5: with fp64 fields the default is the in-line order again.  The z-pass of 1024-point fp64 rows now holds three workgroups
  // per CU (156 KB of LDS) and the strided passes one of 128 KB: a workgroup of the solve (41 KB) fits beside neither, runs in what
  // gaps there are and trails into the strided passes of the next radius, whose times then grow by what the overlap saved --
  // 720.2 / 723.6 ms per step beside, 720.2 / 719.0 in line on one box (profiles/r05_notes.md).  fp32 fields (28 KB of lines per
  // z-pass workgroup) keep it: 507 / 509 against 511 / 512.5 at 1024^3, 566 against 573 for BASELINE config 5's slab.
  bool beside_z = invariants_ok && (c->tune.solve_beside_z < 0 ? c->fb == 4 : c->tune.solve_beside_z != 0);
  // the second field set B2 (six fields: 52 GB at 1024^3 on one rank) is what the LPT part allocates anyway; an Fmax-only run that
  // cannot have it keeps every kernel in line on one field set instead of failing
  if (beside_z && c->fb == 8 && ensure_b2(c, true)) beside_z = false;
  if (beside_z && c->fb == 4 && inv_rows(c, 1, true)) beside_z = false;  // (fp32 fields: the second set of invariant rows is that block)
  const bool two_field_sets = beside_z && c->fb == 8;  // (fp32 fields keep their invariants apart, in INV: two sets of those)
  struct { bool valid; int ismooth, set; } pending = {false, 0, 0};  // the solve that waits for the next z-pass
  bool s_live[2] = {false, false};  // a solve on the solve stream reads set s; ev_s[s] tells when it is done
  auto join_solves = [&]() -> int {  // the compute stream goes on behind everything the solve stream holds
    for (int k = 0; k < 2; k++)
      if (s_live[k]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_s[k], 0)); s_live[k] = false; }
    return 0;
  };
  auto pre = [&](int ismooth, void *const *A) {
    PhaseTimer pt(c, 0);
    return hess_x(c, rep ? c->dk_full : c->dk, radius_cells[ismooth], A, hess_band(c, radius_cells[ismooth]), rep);
  };
  auto post = [&](int ismooth, const void *const *R) {
    // every radius but the last (its Hessian stays in B for the LPT sources): the z-pass stores the three invariants of the
    // tensor instead of its six components and the solve starts from them (PF_INVARIANTS=0: six components throughout)
    const bool inv = invariants_ok && ismooth < ns - 1;
    const bool side = inv && beside_z;
    const int set = side ? (ismooth & 1) : 0;
    void *const *H = set && two_field_sets ? c->B2 : c->B;
    auto fields_of = [&](int s_) -> void *const * { return s_ && two_field_sets ? c->B2 : c->B; };
    if (pending.valid && !side) {  // no invariant z-pass to run beside: the solve still waiting goes in line, first
      if (join_solves()) return 1;
      PhaseTimer pt(c, 1);
      if (collapse_enqueue(c, pending.ismooth, fields_of(pending.set), c->stream, true, true, false, pending.set)) return 1;
      pending.valid = false;
    }
    if (s_live[set]) {  // the y-pass rewrites this set: the solve that reads its invariants must be done
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_s[set], 0));
      s_live[set] = false;
    }
    const std::function<int()> between = [&]() -> int {  // called by hess_yz between its y-pass and its z-pass
      if (!pending.valid) return 0;
      const int ps = pending.set;
      HIPCHK(c, hipEventRecord(c->ev_y[ps], c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->sstream, c->ev_y[ps], 0));
      {
        PhaseTimer pt(c, 1, c->sstream);
        if (collapse_enqueue(c, pending.ismooth, fields_of(ps), c->sstream, true, true, false, ps)) return 1;
      }
      HIPCHK(c, hipEventRecord(c->ev_s[ps], c->sstream));
      s_live[ps] = true;
      pending.valid = false;
      return 0;
    };
    c->inv_w = set;
    {
      PhaseTimer pt(c, 0);
      PFCHK(c, hess_yz(c, R, c->scal + SC_DC_DK, H, hess_band(c, radius_cells[ismooth]), inv, nullptr, nullptr, rep, side ? &between : nullptr));
    }
    c->inv_w = 0;
    if (side) {
      pending.valid = true; pending.ismooth = ismooth; pending.set = set;
      return 0;
    }
    // in line: the running maximum and the partial sums follow the solves of the earlier radii
    if (join_solves()) return 1;
    PhaseTimer pt(c, 1);
    return collapse_enqueue(c, ismooth, c->B, c->stream, true, inv, c->sweep_sources && ismooth == ns - 1);
  };
  if (c->general) {  // one filter + one library c2r per component, then the same collapse pass
    for (int ismooth = 0; ismooth < ns; ismooth++) {
      {
        PhaseTimer pt(c, 0);
        PFCHK(c, g_hessian_of(c, c->dk, radius_cells[ismooth], c->B));
      }
      PhaseTimer pt(c, 1);
      if (collapse_enqueue(c, ismooth, c->B, c->stream, true, false, c->sweep_sources && ismooth == ns - 1)) return 1;
    }
  } else if (pipelined_band(c, ns, 3, pre, [&](int, int set, int f) { return recv_field(c, set, f); }, post,
                            [&](int ismooth) { return hess_band(c, radius_cells[ismooth]); }, rep)) return 1;
  c->solve_ran_beside = beside_z && ns > 2;
  if (pending.valid) return pf_fail(c->rank, "pf_sweep: a solve was left waiting (internal error)");
  if (join_solves()) return 1;
  // the R=0 Hessian (last radius) stays in B for the LPT sources
  c->have_hessian = true; c->last_ns = ns;
  PFCHK(c, allreduce_dev(c, c->scal + SC_INV_FLAG, 1 + 2 * (size_t)ns, 0));
  std::vector<double> s(1 + 2 * ns);
  HIPCHK(c, hipMemcpyAsync(s.data(), c->scal + SC_INV_FLAG, s.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (true_variance)
    for (int i = 0; i < ns; i++) true_variance[i] = s[1 + 2 * i + 1] / ((double)c->n * c->n * c->n);
  return s[0] != 0.0 ? 2 : 0;
}

extern "C" int pf_sweep(pf_ctx *c, int ns, const double *radius_cells, double *true_variance) {
  if (!c || !radius_cells) return pf_fail(0, "pf_sweep: null argument");
  if (ns < 1 || ns > PF_MAX_SMOOTH) return pf_fail(c->rank, "pf_sweep: Nsmooth %d not in [1, %d]", ns, PF_MAX_SMOOTH);
  if (!c->have_density) return pf_fail(c->rank, "pf_sweep: density not set");
  int rc = sweep_body(c, ns, radius_cells, true_variance, false);
  // rc 2: on some rank a cell's tensor had q = (mu1^2 - 3 mu2) / 9 == 0 in floating point without being exactly isotropic.
  // The reference then takes the tensor's own diagonal as the eigenvalues (src/collapse_times.c:722-727), which the three
  // invariants do not determine: the sweep is repeated with six components per cell.  (A Gaussian field never gets there --
  // five independent combinations would have to vanish to 1e-8 at once -- and the exactly isotropic tensors of an empty
  // field do not raise the flag; the flag is summed over the ranks with the variances, so all of them repeat together.)
  if (rc == 2) { c->inv_reruns++; rc = sweep_body(c, ns, radius_cells, true_variance, true); }
  return rc;
}
extern "C" int pf_debug_invariant_reruns(pf_ctx *c) { return c ? c->inv_reruns : -1; }
extern "C" int pf_solve_ran_beside_zpass(pf_ctx *c) { return c ? (c->solve_ran_beside ? 1 : 0) : -1; }
extern "C" int pf_replicated_spectrum(pf_ctx *c) { return c ? (c->replicate ? 1 : 0) : -1; }
// which transforms serve this grid size: 0 the power-of-two passes (pf_fft_kernels.hip), 1 the run-time stage plans for
// n = 8 m, m = 2^a 3^b 5^c (pf_mixed_kernels.hip), 2 library transforms, one per component (pf_gfft.cpp)
extern "C" int pf_transform_path(pf_ctx *c) { return !c ? -1 : c->general ? 2 : (c->n & (c->n - 1)) ? 1 : 0; }
extern "C" int pf_set_transposed_spectra(pf_ctx *c, int on) {
  if (!c) return 1;
  c->transposed = on != 0;
  return 0;
}
extern "C" int pf_set_ct_interpolation(pf_ctx *c, int flavour) {
  if (!c) return 1;
  if (flavour < 0 || flavour > 2) return pf_fail(c->rank, "pf_set_ct_interpolation: flavour %d (0 BILINEAR_SPLINE, 1 TRILINEAR, 2 ALL_SPLINE)", flavour);
  c->ct_flavour = flavour;
  return 0;
}
extern "C" int pf_set_lpt_order(pf_ctx *c, int order) {
  if (!c) return 1;
  if (order < 1 || order > 3) return pf_fail(c->rank, "pf_set_lpt_order: order %d (1 Zel'dovich, 2 -DTWO_LPT, 3 -DTHREE_LPT)", order);
  c->lpt_order = order;
  return 0;
}
extern "C" int pf_set_sources_in_sweep(pf_ctx *c, int on) {
  if (!c) return 1;
  c->sweep_sources = on != 0;
  return 0;
}

extern "C" int pf_displacements(pf_ctx *c, int compute_sources, int recompute_sd) {
  if (!c) return 1;
  if (!c->have_density) return pf_fail(c->rank, "pf_displacements: density not set");
  if (ensure_b2(c)) return 1;
  if (!c->products_init) {
    PFCHK(c, pf_launch_fill_products(c->fmax, c->rmax, c->vel12, ncell(c), c->pb, c->stream));
    c->products_init = true;
  }
  if (recompute_sd && c->lpt_order >= 2) {  // src/fmax.c:301-318 (inside #ifdef TWO_LPT)
    PhaseTimer pt(c, 0);
    PFCHK(c, hessian_of(c, c->dk, 0.0, c->scal + SC_DC_DK, c->B));
    c->have_hessian = true; c->sources_fresh = false;
  }
  {
    PhaseTimer pt(c, 2);
    if (compute_sources && c->lpt_order >= 2) {  // src/LPT.c:46-175 (#ifdef TWO_LPT)
      if (!c->have_hessian) return pf_fail(c->rank, "pf_displacements: second derivatives at R=0 not in place");
      if (!c->sources_fresh) {  // (else: the sweep's last solve has left S2, S3a, the S3b start and the sum of S2)
      PfLptSrcParams sp; memset(&sp, 0, sizeof(sp));
      for (int i = 0; i < 6; i++) sp.h[i] = c->B[i];
      sp.s2 = c->S[0]; sp.s3a = c->S[1]; sp.s3b = c->S[2]; sp.pitch = rpitch(c); sp.nrows = (long long)c->nxl * c->n; sp.n = c->n;
      sp.partials = c->partials;
      size_t nb = (ncell(c) + 255) / 256; if (nb > PF_NBLK) nb = PF_NBLK;
      sp.nblocks = (int)nb;
      {
        KTimer t(c, KS_LPT_SRC, 9.0 * real_bytes_alg(c));
        PFCHK(c, pf_launch_lpt_sources(c->fb, sp, c->stream));
      }
      // DC of the 2LPT source spectrum = sum of S2; it passes the k-filter untouched (k^2 = 0)
      PFCHK(c, pf_launch_sum1(c->partials, sp.nblocks, 1.0 / ((double)c->n * c->n * c->n), c->scal + SC_DC_S2, c->stream));
      }
      c->sources_fresh = false;  // S[0] is transformed in place below
      PFCHK(c, allreduce_dev(c, c->scal + SC_DC_S2, 1, 0));
      PFCHK(c, forward_of(c, c->S[0]));
      // Hessian of the 2LPT potential contracted with the first-order one into the 3LPT(b) source (src/LPT.c:112-137).  fp64
      // fields: the z-pass does the contraction while it holds a row's six components (nothing of that Hessian is stored;
      // PF_LPT_FUSE=0: six fields out, then k_lpt_accum); same operations per cell either way
      const bool fuse3b = (c->tune.lpt_fuse >= 2 ? pf_c2r_invariants_supported(c->fb, (int)c->n) : pf_c2r_invariants_preferred(c->fb, (int)c->n)) && !c->general && c->tune.lpt_fuse;
      if (c->lpt_order < 3) {  // no THREE_LPT (src/LPT.c:78-92, 113-175): the 2LPT source alone
      } else if (fuse3b) {
        PFCHK(c, hessian_of(c, c->S[0], 0.0, c->scal + SC_DC_S2, c->B2, c->S[2], c->B));
      } else {
        PFCHK(c, hessian_of(c, c->S[0], 0.0, c->scal + SC_DC_S2, c->B2));
        PfLptAccParams ap; memset(&ap, 0, sizeof(ap));
        for (int i = 0; i < 6; i++) { ap.h[i] = c->B[i]; ap.phi2[i] = c->B2[i]; }
        ap.s3b = c->S[2]; ap.pitch = rpitch(c); ap.nrows = (long long)c->nxl * c->n; ap.n = c->n;
        KTimer t(c, KS_LPT_ACC, 14.0 * real_bytes_alg(c));
        PFCHK(c, pf_launch_lpt_accum(c->fb, ap, c->stream));
      }
      void *const s12[2] = {c->S[1], c->S[2]};
      if (c->lpt_order >= 3) PFCHK(c, forward_many(c, 2, s12));
      c->have_sources = true; c->sources_order = c->lpt_order;
    } else if (c->lpt_order >= 2 && (!c->have_sources || c->sources_order < c->lpt_order))
      return pf_fail(c->rank, "pf_displacements: LPT sources not resident (call with compute_sources = 1 first)");
    // ScaleDep.order = 2, 3, 4 (src/LPT.c:181-184, 219-221, 226-228), then Zel'dovich (src/fmax.c:342-345); one
    // pipeline so that with P > 1 each exchange runs beside the y/z passes of the previous field
    void *tmp[3] = {c->B2[0], c->B2[1], c->B2[2]};
    // (a build without THREE_LPT / TWO_LPT has no such fields in product_data: their columns here are zero)
    const void *specs[4];
    int orders[4], count = 0;
    if (c->lpt_order >= 2) { specs[count] = c->S[0]; orders[count++] = 1; }
    if (c->lpt_order >= 3) { specs[count] = c->S[1]; orders[count++] = 2; specs[count] = c->S[2]; orders[count++] = 3; }
    specs[count] = c->dk; orders[count++] = 0;
    PFCHK(c, displacements_of(c, count, specs, orders, tmp));
    if (c->lpt_order < 3) {
      const size_t nc = ncell(c);
      const int k0 = c->lpt_order == 2 ? 6 : 3;  // columns 3 o .. 3 o + 2 of order o: 0 Zel'dovich, 1 2LPT, 2 3LPT(a), 3 3LPT(b)
      HIPCHK(c, hipMemsetAsync(velcol(c, k0), 0, (size_t)(12 - k0) * nc * (size_t)c->pb, c->stream));
    }
    c->vel_zero_pending = false;  // all twelve columns rewritten
  }
  return 0;
}

extern "C" int pf_fmax_pdf(pf_ctx *c, unsigned long long hist[PF_NBINS]) {
  if (!c || !hist) return 1;
  HIPCHK(c, hipMemsetAsync(c->hist, 0, PF_NBINS * sizeof(unsigned long long), c->stream));
  PFCHK(c, pf_launch_fmax_pdf(c->fmax, ncell(c), c->hist, c->pb, c->stream));
  PFCHK(c, allreduce_dev(c, c->hist, PF_NBINS, 1));
  HIPCHK(c, hipMemcpyAsync(hist, c->hist, PF_NBINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ----------------------------------------------------------------- outputs --
extern "C" int pf_get_products(pf_ctx *c, void *host, const pf_product_layout *l) {
  if (!c || !host || !l) return pf_fail(0, "pf_get_products: null argument");
  if (l->stride < 8 || l->stride % 4) return pf_fail(c->rank, "pf_get_products: bad stride %zu", l->stride);
  if (l->stride > c->field_bytes) return pf_fail(c->rank, "pf_get_products: a record of %zu bytes exceeds the staging area", l->stride);
  PFCHK(c, velocities_ready(c));
  PhaseTimer pt(c, 3);
  const size_t nc = ncell(c), stride = l->stride;
  const int ov[4] = {l->off_Vel, l->off_Vel_2LPT, l->off_Vel_3LPT_1, l->off_Vel_3LPT_2};
  const int off_rmax = l->off_Rmax, off_fmax = l->off_Fmax;
  // a piece of whole records is packed into one of the two staging fields and copied out while the other is being packed / moved on the host
  auto fill = [&](size_t off, size_t len, char *stage, hipStream_t st) -> int {
    HIPCHK(c, hipMemsetAsync(stage, 0, len, st));
    PFCHK(c, pf_launch_pack_products(c->pb, c->fmax, c->rmax, c->vel12, nc, off / stride, len / stride, stage, stride, off_rmax, off_fmax, ov, st));
    return 0;
  };
  PfHandoff *h;
  if (handoff_get(c, &h)) return 1;
  if (handoff_registered(c, h, host, nc * stride)) {  // the DMA writes into the caller's array itself
    PFCHK(c, handoff_begin(c, h));
    const size_t piece = (c->field_bytes / stride) * stride, bytes = nc * stride;
    size_t k = 0;
    for (size_t off = 0; off < bytes; off += piece, k++) {
      const int b = (int)(k & 1);
      const size_t len = bytes - off < piece ? bytes - off : piece;
      if (fill(off, len, handoff_dev(c, b), h->st[b])) return 1;
      HIPCHK(c, hipMemcpyAsync((char *)host + off, handoff_dev(c, b), len, hipMemcpyDeviceToHost, h->st[b]));
    }
    for (int i = 0; i < 2; i++) HIPCHK(c, hipStreamSynchronize(h->st[i]));
    return 0;
  }
  return handoff_d2h(c, (char *)host, nullptr, nc * stride, stride, fill);
}

// Same columns into records the caller already holds; every other byte of a record (Fmax / Rmax when their offsets are negative, the
// *_prev copies of a RECOMPUTE_DISPLACEMENTS build, padding) keeps its host value.  The re-entrant compute_displacements of
// src/fragment.c:398-410 rewrites only the twelve Vel* columns: only the named fields cross the link -- packed on the device into
// compact records (48 bytes per cell for the float build), copied into pinned memory, scattered into the caller's records by the
// host threads while the next piece travels.  Rounds 1-5 uploaded and downloaded the whole records: 2 x 112 GB at 1024^3 for the
// 104-byte records of the default build, to change 48 bytes per cell.
extern "C" int pf_update_products(pf_ctx *c, void *host, const pf_product_layout *l) {
  if (!c || !host || !l) return pf_fail(0, "pf_update_products: null argument");
  if (l->stride < 8 || l->stride % 4) return pf_fail(c->rank, "pf_update_products: bad stride %zu", l->stride);
  PFCHK(c, velocities_ready(c));
  PhaseTimer pt(c, 3);
  const size_t nc = ncell(c), stride = l->stride;
  // the named fields in the order of their offsets -> compact record (each field aligned as in a C struct), runs of bytes that are
  // adjacent on both sides merged into one segment
  struct Fld { int uoff, len, align, which; };
  Fld fl[6]; int nf = 0;
  if (l->off_Rmax >= 0) fl[nf++] = Fld{l->off_Rmax, 4, 4, 0};
  if (l->off_Fmax >= 0) fl[nf++] = Fld{l->off_Fmax, c->pb, c->pb, 1};
  const int ovu[4] = {l->off_Vel, l->off_Vel_2LPT, l->off_Vel_3LPT_1, l->off_Vel_3LPT_2};
  for (int o = 0; o < 4; o++) if (ovu[o] >= 0) fl[nf++] = Fld{ovu[o], 3 * c->pb, c->pb, 2 + o};
  if (!nf) return 0;
  for (int i = 1; i < nf; i++) for (int j = i; j > 0 && fl[j].uoff < fl[j - 1].uoff; j--) { const Fld t = fl[j]; fl[j] = fl[j - 1]; fl[j - 1] = t; }
  struct Seg { int uoff, coff, len; };
  Seg seg[6]; int nseg = 0;
  int coff_of[6] = {-1, -1, -1, -1, -1, -1}, cpos = 0;
  for (int i = 0; i < nf; i++) {
    if ((size_t)fl[i].uoff + fl[i].len > stride || (i && fl[i].uoff < fl[i - 1].uoff + fl[i - 1].len))
      return pf_fail(c->rank, "pf_update_products: fields of the layout overlap or leave the record");
    cpos = (cpos + fl[i].align - 1) / fl[i].align * fl[i].align;
    coff_of[fl[i].which] = cpos;
    if (nseg && seg[nseg - 1].uoff + seg[nseg - 1].len == fl[i].uoff && seg[nseg - 1].coff + seg[nseg - 1].len == cpos) seg[nseg - 1].len += fl[i].len;
    else seg[nseg++] = Seg{fl[i].uoff, cpos, fl[i].len};
    cpos += fl[i].len;
  }
  const size_t cstride = (size_t)((cpos + 7) / 8 * 8);
  const int cov[4] = {coff_of[2], coff_of[3], coff_of[4], coff_of[5]};
  PfHandoff *h;
  if (handoff_get(c, &h)) return 1;
  PFCHK(c, handoff_begin(c, h));
  const size_t per = h->chunk / cstride;
  if (!per) return pf_fail(c->rank, "pf_update_products: a record does not fit the staging pieces");
  const size_t np = (nc + per - 1) / per;
  auto issue = [&](size_t k) -> int {
    const int b = (int)(k & 1);
    const size_t first = k * per, cnt = nc - first < per ? nc - first : per;
    PFCHK(c, pf_launch_pack_products(c->pb, c->fmax, c->rmax, c->vel12, nc, first, cnt, handoff_dev(c, b), cstride, coff_of[0], coff_of[1], cov, h->st[b]));
    HIPCHK(c, hipMemcpyAsync(h->pin[b], handoff_dev(c, b), cnt * cstride, hipMemcpyDeviceToHost, h->st[b]));
    return 0;
  };
  if (issue(0)) return 1;
  for (size_t k = 0; k < np; k++) {
    if (k + 1 < np && issue(k + 1)) return 1;
    const int b = (int)(k & 1);
    HIPCHK(c, hipStreamSynchronize(h->st[b]));
    const size_t first = k * per, cnt = nc - first < per ? nc - first : per;
    char *dst = (char *)host + first * stride; const char *src = h->pin[b];
    if (nseg == 1) {
      const Seg s0 = seg[0];
      h->pool->run(cnt, [=](size_t a, size_t e) { for (size_t i = a; i < e; i++) memcpy(dst + i * stride + s0.uoff, src + i * cstride + s0.coff, (size_t)s0.len); });
    } else {
      Seg sg[6]; for (int q = 0; q < nseg; q++) sg[q] = seg[q];
      const int ns_ = nseg;
      h->pool->run(cnt, [=](size_t a, size_t e) {
        for (size_t i = a; i < e; i++)
          for (int q = 0; q < ns_; q++) memcpy(dst + i * stride + sg[q].uoff, src + i * cstride + sg[q].coff, (size_t)sg[q].len);
      });
    }
  }
  return 0;
}

// first stage of fragmentation on the device (pf_select_sort.hip)
extern "C" int pf_select_sorted(pf_ctx *c, float flast, size_t capacity, unsigned int *cell_index, float *fmax, size_t *count) {
  if (!c || !count) return pf_fail(0, "pf_select_sorted: null argument");
  if (!c->products_init) return pf_fail(c->rank, "pf_select_sorted: products not computed");
  if (ncell(c) > 0xFFFFFFFFull) return pf_fail(c->rank, "pf_select_sorted: more than 2^32 cells on one rank");
  unsigned int *d_idx = nullptr; float *d_f = nullptr;
  if (c->pb != 4) return pf_fail(c->rank, "pf_select_sorted: fp32 Fmax only (not with PF_FLAG_DOUBLE_PRODUCTS)");
  if (pf_select_sort_device((const float *)c->fmax, ncell(c), flast, &d_idx, &d_f, count, c->stream)) return pf_fail(c->rank, "pf_select_sorted: device sort failed (out of memory?)");
  const size_t m = *count < capacity ? *count : capacity;
  if (m && cell_index) HIPCHK(c, hipMemcpyAsync(cell_index, d_idx, m * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  if (m && fmax) HIPCHK(c, hipMemcpyAsync(fmax, d_f, m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(d_idx);
  return 0;
}

// per-particle payload of one block of the timeless snapshot (src/write_snapshot.c:207-342, 620-855)
extern "C" int pf_get_block(pf_ctx *c, const char *name, int id_bytes, void *host) {
  if (!c || !name || !host) return pf_fail(0, "pf_get_block: null argument");
  if (!c->products_init) return pf_fail(c->rank, "pf_get_block: products not computed");
  if (c->pb != 4) return pf_fail(c->rank, "pf_get_block: the fp32 snapshot blocks are served from fp32 products only (not with PF_FLAG_DOUBLE_PRODUCTS)");
  PFCHK(c, velocities_ready(c));
  const size_t nc = ncell(c);
  if (!strncmp(name, "FMAX", 4)) return handoff_d2h(c, (char *)host, (const char *)c->fmax, nc * sizeof(float), 4, no_fill);
  if (!strncmp(name, "RMAX", 4)) return handoff_d2h(c, (char *)host, (const char *)c->rmax, nc * sizeof(int), 4, no_fill);
  static const char *vec[4] = {"ZEL ", "2LPT", "31PT", "32PT"};
  int o = -1;
  for (int i = 0; i < 4; i++) if (!strncmp(name, vec[i], 4)) o = i;
  const bool id = !strncmp(name, "ID  ", 4);
  if (o < 0 && !id) return pf_fail(c->rank, "pf_get_block: unknown block '%.4s'", name);
  if (id && id_bytes != 4 && id_bytes != 8) return pf_fail(c->rank, "pf_get_block: MYIDTYPE is 4 or 8 bytes");
  const size_t rec = id ? (size_t)id_bytes : 3 * sizeof(float);
  const unsigned long long gfirst = (unsigned long long)c->rank * nc;  // x-slabs: global index = rank * ncell + local
  auto fill = [&](size_t off, size_t len, char *stage, hipStream_t st) -> int {
    if (id) PFCHK(c, pf_launch_block_id(id_bytes, gfirst + off / rec, len / rec, stage, st));
    else PFCHK(c, pf_launch_block_vec3((const float *)c->vel12, nc, o, off / rec, len / rec, (float *)stage, st));
    return 0;
  };
  return handoff_d2h(c, (char *)host, nullptr, nc * rec, rec, fill);
}

extern "C" int pf_get_second_derivative(pf_ctx *c, int i, double *host) {
  if (!c || !host || i < 0 || i > 5) return pf_fail(0, "pf_get_second_derivative: bad argument");
  if (!c->have_hessian) return pf_fail(c->rank, "pf_get_second_derivative: not computed");
  const long long nrows = (long long)c->nxl * c->n;
  PFCHK(c, pf_launch_real_export(c->fb, c->B[i], (double *)staging(c), nrows, c->n, rpitch(c), c->stream));
  return handoff_d2h(c, (char *)host, staging(c), (size_t)nrows * c->n * sizeof(double), 1, no_fill);
}

extern "C" int pf_get_kvector(pf_ctx *c, int which, double *host) {
  if (!c || !host || which < 0 || which > 2) return pf_fail(0, "pf_get_kvector: bad argument");
  if (!c->have_sources) return pf_fail(c->rank, "pf_get_kvector: LPT sources not computed");
  return export_spec(c, c->S[which], host);
}
// test tap: rows kx0 .. kx0 + nkx - 1 of the replicated spectrum this rank transforms, [nkx][n (ky)][n/2+1] complex fp64 (gathered or
// generated first if it is not in place yet)
extern "C" int pf_debug_replicated_rows(pf_ctx *c, int kx0, int nkx, double *host) {
  if (!c || !host || kx0 < 0 || nkx < 1 || kx0 + nkx > c->n) return pf_fail(0, "pf_debug_replicated_rows: bad argument");
  if (!c->replicate) return pf_fail(c->rank, "pf_debug_replicated_rows: this context does not keep the whole spectrum (PF_REPLICATE_DK)");
  if (!c->have_density) return pf_fail(c->rank, "pf_debug_replicated_rows: density not set");
  PFCHK(c, ensure_dk_full(c));
  const long long nrows = (long long)nkx * c->n;
  if ((size_t)nrows * c->nzh * 2 * sizeof(double) > 2 * c->field_bytes) return pf_fail(c->rank, "pf_debug_replicated_rows: %d rows exceed the staging area", nkx);
  const char *src = (const char *)c->dk_full + (size_t)kx0 * c->n * c->nzp * 2 * c->fb;
  PFCHK(c, pf_launch_spec_export(c->fb, src, (double *)staging(c), nrows, c->nzh, c->nzp, c->stream));
  return handoff_d2h(c, (char *)host, staging(c), (size_t)nrows * c->nzh * 2 * sizeof(double), 1, no_fill);
}
extern "C" int pf_get_density(pf_ctx *c, double *host) {
  if (!c || !host) return 1;
  if (!c->have_density) return pf_fail(c->rank, "pf_get_density: density not set");
  return export_spec(c, c->dk, host);
}

static int import_real(pf_ctx *c, const double *host, void *dst) {
  const long long nrows = (long long)c->nxl * c->n;
  if (handoff_h2d(c, staging(c), (const char *)host, (size_t)nrows * c->n * sizeof(double))) return 1;
  PFCHK(c, pf_launch_real_import(c->fb, (const double *)staging(c), dst, nrows, c->n, rpitch(c), c->stream));
  return 0;
}
static int export_real(pf_ctx *c, const void *src, double *host) {
  const long long nrows = (long long)c->nxl * c->n;
  PFCHK(c, pf_launch_real_export(c->fb, src, (double *)staging(c), nrows, c->n, rpitch(c), c->stream));
  return handoff_d2h(c, (char *)host, staging(c), (size_t)nrows * c->n * sizeof(double), 1, no_fill);
}
extern "C" int pf_forward_transform(pf_ctx *c, const double *real_host, double *spec_host) {
  if (!c || !real_host || !spec_host) return 1;
  // P > 1: the y-pass writes the send blocks into A[0], so the field itself lives in receive field 0
  void *f = c->P > 1 ? recv_field(c, 0, 0) : c->A[0];
  PFCHK(c, import_real(c, real_host, f));
  PFCHK(c, forward_of(c, f));
  return export_spec(c, f, spec_host);
}
extern "C" int pf_reverse_transform(pf_ctx *c, const double *spec_host, double *real_host) {
  if (!c || !real_host || !spec_host) return 1;
  PFCHK(c, import_spec(c, spec_host, c->A[0]));
  PFCHK(c, reverse_of(c, c->A[0]));
  return export_real(c, c->A[0], real_host);
}

// compute_derivative (src/fmax-pfft.c:255-441) on a caller-held spectrum: one component, host in and out.  The
// multiplier k_a k_b / k^2 (or i k_a / k^2, or -1/k^2) factorises over the three passes like the shared-pass
// transforms of the sweep: x-pass applies the window, 1/k^2, the growth of ScaleDep.order and its own k power.
extern "C" int pf_derivative(pf_ctx *c, const double *spec_host, int first_derivative, int second_derivative, double rs_cells,
                             int order, double *real_host) {
  if (!c || !spec_host || !real_host) return pf_fail(0, "pf_derivative: null argument");
  const int a = first_derivative, b = second_derivative;
  if (a < 0 || a > 3 || b < 0 || b > 3) return pf_fail(c->rank, "pf_derivative: components (%d,%d) not in 0..3", a, b);
  if (order < 0 || order > 4) return pf_fail(c->rank, "pf_derivative: ScaleDep.order %d not in 0..4", order);
  const bool swap = (a == 0) != (b == 0);  // first derivative: multiply by i (src/fmax-pfft.c:299-300, 379-384)
  int mul[3] = {PF_MUL_ONE, PF_MUL_ONE, PF_MUL_ONE};
  for (int ax = 1; ax <= 3; ax++) {
    const int cnt = (a == ax) + (b == ax);
    mul[ax - 1] = cnt == 2 ? PF_MUL_K2 : cnt == 1 ? (swap ? PF_MUL_IK : PF_MUL_K) : PF_MUL_ONE;
  }
  double growth = order ? c->growth[order - 1] : 1.0;
  if (a == 0 && b == 0) growth = -growth;     // greens_function: -1/k^2 (src/fmax-pfft.c:449-450)
  void *f = c->A[0];
  PFCHK(c, import_spec(c, spec_host, f));
  if (c->general) {
    PFCHK(c, g_filter(c, f, c->W, a, b, rs_cells, order, true));
    PFCHK(c, g_c2r(c, c->W, f));
    return export_real(c, f, real_host);
  }
  PFCHK(c, dc_of_host_spec(c, spec_host, SC_DC_TMP));
  if (order && c->gt_n[order - 1]) {
    const int o = order - 1;
    PFCHK(c, pf_launch_apply_growth(c->fb, f, f, c->n, c->nyl, c->nzh, c->nzp, c->rank * c->nyl, c->gtab + o * PF_KBIN_CAP, c->gt_n[o],
                                    c->gt_logkmin[o], c->gt_dlogk[o], c->gt_sign[o], c->stream));
    growth = (a == 0 && b == 0) ? -1.0 : 1.0;
  }
  const Job xj[1] = {{f, f, mul[0]}};
  PFCHK(c, xpass(c, KS_XPASS_PLAIN, +1, 1, xj, 1, rs_cells, growth, 1));
  const void *R = f;
  if (c->P > 1) { PFCHK(c, exchange(c, f, c->recvA)); R = c->recvA; }
  const Job yj[1] = {{R, f, mul[1]}};
  PFCHK(c, ypass(c, KS_YPASS_PLAIN, +1, 1, yj, true, false, 1));
  const ZJob zj[1] = {{f, f, mul[2], 0}};
  // the untouched k = 0 mode survives only without the swap (then Im of the DC mode is ignored by the c2r)
  PFCHK(c, zpass_c2r(c, KS_ZPASS_PLAIN, 1, zj, swap ? nullptr : c->scal + SC_DC_TMP));
  return export_real(c, f, real_host);
}

extern "C" int pf_collapse_cells(pf_ctx *c, int ismooth, const double *d, size_t count, double *F) {
  if (!c || !d || !F) return 1;
  if (count * 7 * sizeof(double) > 2 * c->field_bytes) return pf_fail(c->rank, "pf_collapse_cells: %zu cells exceed the staging area", count);
  PfSplineDev s;
  if (spline_for(c, ismooth, &s)) return 1;
  double *dd = (double *)staging(c), *df = dd + 6 * count;
  HIPCHK(c, hipMemcpyAsync(dd, d, 6 * count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PFCHK(c, pf_launch_collapse_cells(dd, count, s, df, c->fast_libm ? 1 : 0, c->stream));
  HIPCHK(c, hipMemcpyAsync(F, df, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// test tap: elementary functions of the solver's fast flavour on the device, which = 0 a/b, 1 sqrt a, 2 acos a, 3 log10 a,
// 4 sin (b != 0) or cos (b == 0) of a in [0, pi/3], 5 a^0.333333333333333, 6 a/9, 7 exp a, 8 10^a, 9 / 10 the hardware seeds rcp a / rsq a
extern "C" int pf_debug_math(pf_ctx *c, int which, const double *a, const double *b, size_t count, double *out) {
  if (!c || !a || !b || !out) return 1;
  if (count * 3 * sizeof(double) > 2 * c->field_bytes) return pf_fail(c->rank, "pf_debug_math: %zu values exceed the staging area", count);
  double *da = (double *)staging(c), *db = da + count, *dout = db + count;
  HIPCHK(c, hipMemcpyAsync(da, a, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(db, b, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PFCHK(c, pf_launch_debug_math(which, da, db, count, dout, c->stream));
  HIPCHK(c, hipMemcpyAsync(out, dout, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Streaming yardsticks on this context's own buffers: GB/s of a kernel that only reads (kind 0: delta(k)), only writes (1) or copies
// (2) one field, plain 16-byte accesses; the written field is a send buffer of the x-pass (scratch between transforms).
extern "C" int pf_debug_stream_rate(pf_ctx *c, int kind, int reps, double *gbps) {
  if (!c || !gbps || kind < 0 || kind > 2 || reps < 1) return 1;
  if (c->general) return pf_fail(c->rank, "pf_debug_stream_rate: power-of-two grids only");
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  float *sink = (float *)(c->scal + SC_POWER);  // a device scalar nobody reads between generator runs (never written: the condition is never met)
  PFCHK(c, pf_launch_stream(kind, c->dk, c->A[0], c->field_bytes, sink, c->stream));
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; r++) PFCHK(c, pf_launch_stream(kind, c->dk, c->A[0], c->field_bytes, sink, c->stream));
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  *gbps = (kind == 2 ? 2.0 : 1.0) * (double)c->field_bytes * reps / (ms * 1e-3) / 1e9;
  return 0;
}

// ------------------------------------------------------------- measurement --
extern "C" int pf_get_cputime(pf_ctx *c, pf_cputime *t) {
  if (!c || !t) return 1;
  resolve_events(c);
  *t = c->cpu;
  // deriv and coll are spans on their streams.  With the solve stream on (PF_SOLVE_BESIDE_Z) the collapse pass of radius i runs
  // beside the z-pass of radius i + 1, both spans cover that time, and the reference's report adds the two (src/fmax.c:160-170):
  // coll keeps what the collapse passes ADD to the sweep -- the part of their span the derivative passes do not cover
  if (c->solve_ran_beside && t->fmax > 0.0 && t->deriv + t->coll > t->fmax) t->coll = t->fmax > t->deriv ? t->fmax - t->deriv : 0.0;
  // FFT share: every pass kernel (src/fmax-pfft.c:195-199 times pfft_execute only)
  resolve_events(c);
  double fft = 0;
  for (int k = 0; k < KS_COUNT; k++)
    if (k != KS_COLLAPSE && k != KS_COLLAPSE_INV && k != KS_COLLAPSE_SRC && k != KS_LPT_SRC && k != KS_LPT_ACC && k != KS_MISC) fft += 1e-3 * c->ks_ms[k];
  t->fft = fft;
  return 0;
}
extern "C" int pf_reset_cputime(pf_ctx *c) { if (!c) return 1; resolve_events(c); memset(&c->cpu, 0, sizeof(c->cpu)); return 0; }
extern "C" int pf_kernel_stats(pf_ctx *c, pf_kernel_stat *out, int max, int *n) {
  if (!c || !out || !n) return 1;
  resolve_events(c);
  int k = 0;
  for (int i = 0; i < KS_COUNT && k < max; i++) {
    if (!c->ks_n[i]) continue;
    memset(&out[k], 0, sizeof(out[k]));
    strncpy(out[k].name, ks_names[i], sizeof(out[k].name) - 1);
    out[k].launches = c->ks_n[i]; out[k].total_ms = c->ks_ms[i]; out[k].alg_bytes = c->ks_bytes[i];
    k++;
  }
  *n = k;
  return 0;
}
extern "C" int pf_reset_kernel_stats(pf_ctx *c) {
  if (!c) return 1;
  resolve_events(c);
  memset(c->ks_ms, 0, sizeof(c->ks_ms)); memset(c->ks_bytes, 0, sizeof(c->ks_bytes)); memset(c->ks_n, 0, sizeof(c->ks_n));
  return 0;
}

// Exchange self-test: fills the send blocks with a rank/block pattern, runs the installed all-to-all (also on one
// rank, where the path itself never calls it) and checks what arrived.  Returns 0 when every word is right.
extern "C" int pf_debug_exchange(pf_ctx *c, size_t bytes_per_peer) {
  if (!c) return 1;
  if (!c->a2a) return pf_fail(c->rank, "pf_debug_exchange: no exchange installed");
  if (bytes_per_peer % 8 || bytes_per_peer * c->P > c->field_bytes) return pf_fail(c->rank, "pf_debug_exchange: bad size");
  const size_t nw = bytes_per_peer / 8;
  std::vector<unsigned long long> h(nw * c->P);
  for (int q = 0; q < c->P; q++)
    for (size_t i = 0; i < nw; i++) h[q * nw + i] = ((unsigned long long)c->rank << 48) | ((unsigned long long)q << 40) | i;
  void *recv = c->P > 1 ? c->recvA : c->A[1];
  HIPCHK(c, hipMemcpyAsync(c->A[0], h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(recv, 0, h.size() * 8, c->stream));
  if (c->a2a(c->a2a_user, c->A[0], recv, bytes_per_peer, (void *)c->stream)) return pf_fail(c->rank, "pf_debug_exchange: all-to-all failed");
  HIPCHK(c, hipMemcpyAsync(h.data(), recv, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int p = 0; p < c->P; p++)
    for (size_t i = 0; i < nw; i++)
      if (h[p * nw + i] != (((unsigned long long)p << 48) | ((unsigned long long)c->rank << 40) | i))
        return pf_fail(c->rank, "pf_debug_exchange: wrong word from rank %d at %zu", p, i);
  if (c->a2av) {  // the row-range form: every rank a different range, some empty
    std::vector<size_t> off(c->P), len(c->P);
    for (int p = 0; p < c->P; p++) { off[p] = 8 * (size_t)((p % 3) * 37); len[p] = 8 * (size_t)(((p + 1) % 4) * 129); if (off[p] + len[p] > bytes_per_peer) len[p] = 0; }
    for (int q = 0; q < c->P; q++)
      for (size_t i = 0; i < nw; i++) h[q * nw + i] = ((unsigned long long)c->rank << 48) | ((unsigned long long)q << 40) | i;
    HIPCHK(c, hipMemcpyAsync(c->A[0], h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(recv, 0, h.size() * 8, c->stream));
    if (c->a2av(c->a2av_user, c->A[0], recv, bytes_per_peer, off[c->rank], len[c->rank], off.data(), len.data(), (void *)c->stream))
      return pf_fail(c->rank, "pf_debug_exchange: row-range all-to-all failed");
    HIPCHK(c, hipMemcpyAsync(h.data(), recv, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int p = 0; p < c->P; p++)
      for (size_t i = 0; i < nw; i++) {
        const bool in = 8 * i >= off[p] && 8 * i < off[p] + len[p];
        const unsigned long long want = in ? (((unsigned long long)p << 48) | ((unsigned long long)c->rank << 40) | i) : 0ull;
        if (h[p * nw + i] != want) return pf_fail(c->rank, "pf_debug_exchange: row-range exchange, wrong word from rank %d at %zu", p, i);
      }
  }
  if (c->ared) {
    double v[2] = {1.0 + c->rank, 2.0};
    HIPCHK(c, hipMemcpyAsync(c->scal, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
    if (c->ared(c->ared_user, c->scal, 2, 0, (void *)c->stream)) return pf_fail(c->rank, "pf_debug_exchange: all-reduce failed");
    HIPCHK(c, hipMemcpyAsync(v, c->scal, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (v[0] != 0.5 * c->P * (c->P + 1) || v[1] != 2.0 * c->P) return pf_fail(c->rank, "pf_debug_exchange: wrong all-reduce result");
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 2 * sizeof(double), c->stream));
  }
  return 0;
}

// Test tap without a context: one of the three pass kernels on a batch of lines, host in and host out (fp64 on the host,
// converted to the field type on the way).  It exists so that every instantiation -- in particular N = 2048 (BASELINE
// config 5), whose 3-D box does not fit one GPU -- can be compared with an independent transform line by line.
//   pass 0 / 1: k_strided inverse (+1) / forward (-1): in, out complex [nouter][n][ncols]; `mul`, `band` (loads with
//               |wavenumber| > band are zeros; >= n/2: none), pre / rs / growth / outer_offset as in the x-pass
//   pass 2: z-pass c2r (k_c2r_persistent, or k_c2r with PF_ZPASS_PERSIST=0): in complex [nouter][n/2+1], out real [nouter][n],
//           unnormalised, `mul` in kz, columns kz > band not read
//   pass 3: k_r2c: in real [nouter][n], out complex [nouter][n/2+1]
//   pass 4: k_c2r_invariants (n <= 1024): in complex [6][nouter][n/2+1], out real [3][nouter][n] = mu1, mu2, mu3 of the six rows
extern "C" int pf_debug_lines(int field_bytes, int n, int pass, int mul, int band, int nouter, int ncols, int pre, double rs,
                              double growth, int outer_offset, const double *in, double *out) {
  double *flag_out = (pass == 4 && out) ? out + (size_t)3 * nouter * n : nullptr;  // pass 4: one more double after the rows, the q == 0 flag
  if (!in || !out || nouter < 1 || (field_bytes != 8 && field_bytes != 4) || n < 8 || n > 2048 || pass < 0 || pass > 4)
    return pf_fail(0, "pf_debug_lines: bad argument");
  if (n & (n - 1)) {  // not a power of two: the run-time stage plans (pf_mixed_kernels.hip), where they apply
    PfMixedPlan pl;
    if (!(pass <= 1 ? pf_mixed_plan(n, false, &pl) : (n % 2 == 0 && pf_mixed_plan(n / 2, true, &pl))))
      return pf_fail(0, "pf_debug_lines: no stage plan for %d points in pass %d", n, pass);
  } else if (n < 16) return pf_fail(0, "pf_debug_lines: bad argument");
  if (pass == 4 && !pf_c2r_invariants_supported(field_bytes, n))
    return pf_fail(0, "pf_debug_lines: the invariant z-pass takes rows whose six lines fit a workgroup (fp64 rows of at most 1024 points, fp32 rows of at most 2048; n = 8 m as the grids)");
  const int fb = field_bytes, nzh = n / 2 + 1;
  pf_ctx *nc = nullptr;  // for the error macros
  size_t n_in, n_out;    // scalars of type F
  // strided passes: rows of `ncols` complex on the host, padded to whole tiles on the device as the library's fields are (nzp):
  // the two-column fp32 kernels move a column pair per 16-byte access, the unpredicated kernels whole tiles
  const int pc = (ncols + 15) & ~15;  // (whole tiles of every instantiation: 8 fp64, 16 fp32 columns -- the unpredicated form of the pass runs too)
  if (pass <= 1) { if (ncols < 1) return pf_fail(0, "pf_debug_lines: ncols"); n_in = n_out = (size_t)nouter * n * ncols * 2; }
  else if (pass == 2) { n_in = (size_t)nouter * nzh * 2; n_out = (size_t)nouter * n; }
  else if (pass == 3) { n_in = (size_t)nouter * n; n_out = (size_t)nouter * nzh * 2; }
  else { n_in = (size_t)6 * nouter * nzh * 2; n_out = (size_t)3 * nouter * n; }
  // pass 3 runs in place like forward_r2c: the buffer must hold the longer of the two rows
  const size_t n_in_alloc = pass == 3 ? (n_in > n_out ? n_in : n_out) : n_in;
  void *d_in = nullptr, *d_out = nullptr, *d_tw = nullptr; double *d_etab = nullptr, *d_flag = nullptr;
  PfTuning tune;
  read_tuning(&tune);
  int dev = 0, ncu = 0;
  {
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return pf_fail(0, "pf_debug_lines: no HIP device");
    ncu = prop.multiProcessorCount;
  }
  std::vector<char> tw;
  host_twiddles(n, fb, tw);
  int rc = 0;
  auto body = [&]() -> int {
    const size_t nrows01 = (size_t)nouter * n;                        // rows of a strided pass
    const size_t dev01 = nrows01 * (size_t)pc * 2;                     // ... and its scalars on the device
    HIPCHK(nc, hipMalloc(&d_in, (pass <= 1 ? dev01 : n_in_alloc) * fb));
    const size_t out_b = (pass == 4) ? 8 : (size_t)fb;  // the invariants are fp64 whatever the fields are
    HIPCHK(nc, hipMalloc(&d_out, (pass <= 1 ? dev01 : n_out) * out_b));
    HIPCHK(nc, hipMalloc(&d_tw, tw.size()));
    HIPCHK(nc, hipMalloc((void **)&d_etab, (size_t)n * sizeof(double)));
    HIPCHK(nc, hipMalloc((void **)&d_flag, sizeof(double)));
    HIPCHK(nc, hipMemset(d_flag, 0, sizeof(double)));
    HIPCHK(nc, hipMemcpy(d_tw, tw.data(), tw.size(), hipMemcpyHostToDevice));
    const size_t hrow = (size_t)ncols * 2 * fb, drow = (size_t)pc * 2 * fb;  // bytes of a host / device row (strided passes)
    if (pass <= 1) HIPCHK(nc, hipMemset(d_in, 0, dev01 * fb));
    if (fb == 8) {
      if (pass <= 1) HIPCHK(nc, hipMemcpy2D(d_in, drow, in, hrow, hrow, nrows01, hipMemcpyHostToDevice));
      else HIPCHK(nc, hipMemcpy(d_in, in, n_in * 8, hipMemcpyHostToDevice));
    } else {
      std::vector<float> h(n_in);
      for (size_t i = 0; i < n_in; i++) h[i] = (float)in[i];
      if (pass <= 1) HIPCHK(nc, hipMemcpy2D(d_in, drow, h.data(), hrow, hrow, nrows01, hipMemcpyHostToDevice));
      else HIPCHK(nc, hipMemcpy(d_in, h.data(), n_in * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(nc, hipMemset(d_out, 0, (pass <= 1 ? dev01 : n_out) * out_b));
    if (pass <= 1) {
      PfStridedParams p; memset(&p, 0, sizeof(p));
      p.njobs = 1; p.job[0].in = d_in; p.job[0].out = d_out; p.job[0].mul = mul;
      p.ain.os = (long long)n * pc; p.ain.el_shift = ilog2i(n); p.ain.el_len = n; p.ain.ehs = 0; p.ain.els = pc; p.aout = p.ain;
      p.ncols = ncols; p.nouter = nouter; p.pre = pre; p.outer_offset = outer_offset; p.rs = rs; p.growth = growth; p.tw = d_tw; p.etab = d_etab;
      p.band_e = p.band_outer = n; p.dev = dev;
      if (band < n / 2) p.band_e = band;
      if (pre && rs != 0.0) PFCHK0(pf_launch_exp_table(d_etab, n, rs, nullptr));
      PFCHK0(pf_launch_strided(fb, n, pass == 0 ? +1 : -1, p, nullptr));
    } else if (pass == 2 || pass == 4) {
      PfC2RParams p; memset(&p, 0, sizeof(p));
      p.njobs = pass == 2 ? 1 : 6;
      static const int mul6[6] = {PF_MUL_ONE, PF_MUL_ONE, PF_MUL_K2, PF_MUL_ONE, PF_MUL_K, PF_MUL_K};
      for (int j = 0; j < p.njobs; j++) {
        p.job[j].in = (char *)d_in + (size_t)j * nouter * nzh * 2 * fb;
        p.job[j].out = pass == 2 ? d_out : (char *)d_out + (size_t)(j % 3) * nouter * n * 8;
        if (pass == 4 && j < 3) p.inv_out[j] = (double *)((char *)d_out + (size_t)j * nouter * n * 8);
        p.job[j].mul = pass == 2 ? mul : mul6[j]; p.job[j].out_f32 = 0;
      }
      p.nlines = nouter; p.in_pitch = nzh; p.out_pitch = n; p.norm = 1.0; p.dc = nullptr; p.tw = d_tw; p.band_k = band < n / 2 ? band : n;
      p.ncu = ncu; p.dev = dev; p.persist_per_cu = tune.zpass_persist; p.inv_per_cu = tune.zpass_inv_wg_per_cu; p.flag = d_flag;
      p.inv_pitch = n;
      if (pass == 2) PFCHK0(pf_launch_c2r(fb, n, p, nullptr));
      else PFCHK0(pf_launch_c2r_invariants(fb, n, p, nullptr, 0));
    } else {
      PfR2CParams p; p.in = d_in; p.out = d_out; p.nlines = nouter; p.in_pitch = n; p.out_pitch = nzh; p.tw = d_tw;
      PFCHK0(pf_launch_r2c(fb, n, p, nullptr));
    }
    HIPCHK(nc, hipDeviceSynchronize());
    if (pass == 4 && flag_out) HIPCHK(nc, hipMemcpy(flag_out, d_flag, sizeof(double), hipMemcpyDeviceToHost));
    if (fb == 8 || pass == 4) {
      if (pass <= 1) HIPCHK(nc, hipMemcpy2D(out, hrow, d_out, drow, hrow, nrows01, hipMemcpyDeviceToHost));
      else HIPCHK(nc, hipMemcpy(out, d_out, n_out * 8, hipMemcpyDeviceToHost));
    } else {
      std::vector<float> h(n_out);
      if (pass <= 1) HIPCHK(nc, hipMemcpy2D(h.data(), hrow, d_out, drow, hrow, nrows01, hipMemcpyDeviceToHost));
      else HIPCHK(nc, hipMemcpy(h.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n_out; i++) out[i] = (double)h[i];
    }
    return 0;
  };
  rc = body();
  hipFree(d_in); hipFree(d_out); hipFree(d_tw); hipFree(d_etab); hipFree(d_flag);
  return rc;
}

// Test tap without a context: ONE strided launch with several jobs, as the passes of the sweep issue them -- job j transforms input
// in_of[j] (one of `nin` complex fields [nouter][n][ncols]) with the factor mul[j] into its own output field; jobs that share an input
// are adjacent (the launch's contract).  What pf_debug_lines cannot reach: tiles that serve several jobs (kept in registers, or read
// again by every job / by one workgroup per job).  out: [njobs][nouter][n][ncols].
extern "C" int pf_debug_strided_jobs(int field_bytes, int n, int njobs, int nin, const int *in_of, const int *mul, int nouter, int ncols,
                                     const double *in, double *out) {
  if (!in || !out || !in_of || !mul || njobs < 1 || njobs > PF_MAX_JOBS || nin < 1 || nin > njobs || nouter < 1 || ncols < 1 || (field_bytes != 8 && field_bytes != 4) ||
      n < 16 || n > 2048 || ((n & (n - 1)) && !pf_mixed_supported(n)))
    return pf_fail(0, "pf_debug_strided_jobs: bad argument");
  const int fb = field_bytes, pc = (ncols + 15) & ~15;
  pf_ctx *nc = nullptr;
  const size_t nrows = (size_t)nouter * n, field = nrows * (size_t)pc * 2;  // scalars of one device field
  void *d_in = nullptr, *d_out = nullptr, *d_tw = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return pf_fail(0, "pf_debug_strided_jobs: no HIP device");
  std::vector<char> tw;
  host_twiddles(n, fb, tw);
  auto body = [&]() -> int {
    HIPCHK(nc, hipMalloc(&d_in, (size_t)nin * field * fb));
    HIPCHK(nc, hipMalloc(&d_out, (size_t)njobs * field * fb));
    HIPCHK(nc, hipMalloc(&d_tw, tw.size()));
    HIPCHK(nc, hipMemcpy(d_tw, tw.data(), tw.size(), hipMemcpyHostToDevice));
    HIPCHK(nc, hipMemset(d_in, 0, (size_t)nin * field * fb));
    HIPCHK(nc, hipMemset(d_out, 0, (size_t)njobs * field * fb));
    const size_t hrow = (size_t)ncols * 2 * fb, drow = (size_t)pc * 2 * fb, hfield = nrows * (size_t)ncols * 2;
    std::vector<float> hf;
    for (int i = 0; i < nin; i++) {
      const void *src = in + (size_t)i * hfield;
      if (fb == 4) { hf.resize(hfield); for (size_t k = 0; k < hfield; k++) hf[k] = (float)in[(size_t)i * hfield + k]; src = hf.data(); }
      HIPCHK(nc, hipMemcpy2D((char *)d_in + (size_t)i * field * fb, drow, src, hrow, hrow, nrows, hipMemcpyHostToDevice));
    }
    PfStridedParams p; memset(&p, 0, sizeof(p));
    p.njobs = njobs;
    for (int j = 0; j < njobs; j++) {
      if (in_of[j] < 0 || in_of[j] >= nin) return pf_fail(0, "pf_debug_strided_jobs: in_of");
      p.job[j].in = (char *)d_in + (size_t)in_of[j] * field * fb; p.job[j].out = (char *)d_out + (size_t)j * field * fb; p.job[j].mul = mul[j];
    }
    p.ain.os = (long long)n * pc; p.ain.el_shift = (n & (n - 1)) ? 0 : ilog2i(n); p.ain.el_len = n; p.ain.ehs = 0; p.ain.els = pc; p.aout = p.ain;
    p.ncols = ncols; p.nouter = nouter; p.tw = d_tw; p.band_e = p.band_outer = n; p.dev = dev; p.growth = 1.0;
    PFCHK0(pf_launch_strided(fb, n, +1, p, nullptr));
    HIPCHK(nc, hipDeviceSynchronize());
    for (int j = 0; j < njobs; j++) {
      if (fb == 8) HIPCHK(nc, hipMemcpy2D(out + (size_t)j * hfield, hrow, (char *)d_out + (size_t)j * field * fb, drow, hrow, nrows, hipMemcpyDeviceToHost));
      else {
        hf.resize(hfield);
        HIPCHK(nc, hipMemcpy2D(hf.data(), hrow, (char *)d_out + (size_t)j * field * fb, drow, hrow, nrows, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < hfield; k++) out[(size_t)j * hfield + k] = (double)hf[k];
      }
    }
    return 0;
  };
  const int rc = body();
  hipFree(d_in); hipFree(d_out); hipFree(d_tw);
  return rc;
}

// used by pf_rccl.cpp
extern "C" int pf_ctx_set_rccl(pf_ctx *c, void *link) {
  if (!c) return 1;
  pf_rccl_release(c->rccl);
  c->rccl = link;
  return 0;
}
// drops the built-in RCCL exchange (communicator destroyed, callbacks cleared): a caller that falls back to another
// exchange kind after pf_init_rccl leaves nothing of the first one behind
extern "C" int pf_release_rccl(pf_ctx *c) {
  if (!c) return 1;
  if (c->rccl) {
    hipStreamSynchronize(c->stream); hipStreamSynchronize(c->cstream);
    pf_rccl_release(c->rccl); c->rccl = nullptr;
    c->a2a = nullptr; c->a2av = nullptr; c->ared = nullptr; c->a2a_user = c->a2av_user = c->ared_user = nullptr;
  }
  return 0;
}
extern "C" int pf_rccl_comm_count(pf_ctx *c) {
  if (!c) return -1;
  return c->rccl ? pf_rccl_link_count(c->rccl) : 0;
}
extern "C" int pf_ctx_rank_size(pf_ctx *c, int *rank, int *nranks) {
  if (!c) return 1;
  *rank = c->rank; *nranks = c->P;
  return 0;
}
