// pf_internal.h -- launch interface between the translation units of libpinfmax_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/pinfmax.h"

#define PF_MAX_JOBS 6
#define PF_MAX_DEVICES 64

// run-time switches, read from the environment once per context in pf_create (pf_api.hip: read_tuning);
// launch paths see only this struct
struct PfTuning {
  int zpass_persist;        // PF_ZPASS_PERSIST: workgroups per CU of the persistent z-pass, 0 = one-shot workgroups
  int zpass_inv_wg_per_cu;  // PF_ZPASS_INV_WG_PER_CU
  int collapse_wg_per_cu;   // PF_COLLAPSE_WG_PER_CU
  bool spline_lut, exchange_rows, general, pipeline, exact_libm;
  int invariants, lpt_fuse;  // PF_INVARIANTS / PF_LPT_FUSE: 0 off, 1 where the invariant z-pass is the faster way (pf_c2r_invariants_preferred), 2 wherever it exists
  int replicate;  // PF_REPLICATE_DK: -1 by rank count, 0 off, 1 on
  double prune_eps;
  int debug_fault;          // PF_DEBUG_PIPELINE_FAULT (tests only): 1 / 2 = one wait of the exchange pipeline left out
  int solve_beside_z;       // PF_SOLVE_BESIDE_Z: the solve of sweep radius i runs on its own stream beside the z-pass of radius i + 1 (1), every
                            // kernel in line (0), or -- the default, -1 -- beside with fp32 fields and in line with fp64 fields (pf_sweep)
  int handoff_chunk_mb, handoff_threads;  // PF_HANDOFF_CHUNK_MB, PF_HANDOFF_THREADS: pieces and host threads of the hand-off (pf_api.hip PfHandoff)
  bool host_register;       // PF_HOST_REGISTER: register the caller's product array with the driver, DMA straight into it
  bool preflight;           // PF_PREFLIGHT: pf_create holds the plan's bytes against the device's free memory before it allocates
  bool gtab;                // PF_GTAB: the inverse growing mode of the fast flavour from the polynomial table (pf_gtab.h)
};

// multiplier applied along the transformed axis before the 1-D transform
// (k = 2 pi s / N, s the signed wavenumber of src/fmax-pfft.c:306-339)
enum { PF_MUL_ONE = 0, PF_MUL_K = 1, PF_MUL_K2 = 2, PF_MUL_IK = 3 };

// element address of (outer, e, col) in a strided pass:
//   outer*os + (e >> el_shift)*ehs + (e & (el-1))*els + col      (units: complex elements)
// (e / el, e % el) lets the y-pass read the P received all-to-all blocks in place.
struct PfAddr {
  long long os;
  int el_shift;  // el = 1 << el_shift (slab thickness; a power of two where n and P are: what the power-of-two kernels split by)
  long long ehs, els;
  int el_len;    // the slab thickness itself (n, n / P): what the mixed-radix kernels split by (multiply-high), any length
};

struct PfStridedJob {
  const void *in;
  void *out;
  int mul;
};

struct PfStridedParams {
  int njobs;
  PfStridedJob job[PF_MAX_JOBS];
  PfAddr ain, aout;
  int ncols;         // valid columns (n/2+1)
  int nouter;        // lines of tiles along the non-transformed slow axis
  int pre;           // 1: multiply by exp(-k^2 rs^2/2) * growth / k^2 (0 at k = 0) on load
  int outer_offset;  // global index of outer = 0 (k-space y-slab start)
  double rs, growth;
  // pre: the Gaussian window is separable, exp(-k^2 rs^2/2) = E(k_e) * exp(-(k_o^2 + k_c^2) rs^2/2): etab[e] = E of the
  // transformed axis (n doubles, pf_launch_exp_table) leaves one exp per thread instead of eight; rs == 0 needs none
  const double *etab;
  const void *tw;    // exp(+2 pi i j / n), n entries of complex F
  // pruned transform of a band-limited (Gaussian-smoothed) spectrum: wavenumbers |s| > band carry a
  // weight below 2^-60 and are treated as exact zeros.  band_e masks the loads along the transformed
  // axis, band_outer skips whole workgroups; columns are pruned by ncols.  band >= n/2 disables.
  int band_e, band_outer;
  int dev;           // device the launch goes to (the LDS-size attribute of an instantiation is raised once per device)
  // out_ne > 0: only the outputs e in [out_e0, out_e0 + out_ne) are stored (a rank that holds the whole spectrum and
  // transforms every line, but keeps its own slab of the transformed axis: no exchange afterwards)
  int out_e0, out_ne;
};

// one x- or y-pass: for every job, out = FFT_e[ in * pre * mul ]  (dir = +1 inverse, -1 forward)
int pf_launch_strided(int field_bytes, int n, int dir, const PfStridedParams &p, hipStream_t st);
int pf_launch_exp_table(double *etab, int n, double rs, hipStream_t st);
// 2048-point fp32 lines, sixteen points per thread (pf_fft16_kernels.hip); -1: not a case of that kernel
int pf_launch_strided16(int field_bytes, int n, int dir, const PfStridedParams &p, hipStream_t st);

struct PfC2RJob {
  const void *in;   // complex rows
  void *out;        // real rows (type F, pitch out_pitch); out_f32 == 1: float rows of pitch n; == 2: rows of type F, pitch n
  int mul;          // multiplier in kz
  int out_f32;
};

struct PfC2RParams {
  int njobs;
  PfC2RJob job[PF_MAX_JOBS];
  long long nlines;     // rows = nx_local * n
  long long in_pitch;   // complex elements per input row
  long long out_pitch;  // real elements per output row (F outputs)
  double norm;          // 1/N^3 applied to the result (src/fmax-pfft.c:220-225)
  const double *dc;     // device scalar added after normalisation (DC mode of 2nd derivatives), or null
  const void *tw;       // exp(+2 pi i j / n)
  int band_k;           // input columns kz > band_k are zero (pruned), not read
  void *acc;            // k_c2r_invariants MODE 1: the real field updated in place (3LPT(b) source); job[c].out = first-order Hessian
  double *flag;         // k_c2r_invariants MODE 0: set to 1 when a cell has q == 0 without being exactly isotropic (pf_sweep repeats)
  int ncu, dev;         // CUs of the device, device index
  int persist_per_cu;   // workgroups per CU of the persistent z-pass (0: one-shot kernel)
  int inv_per_cu;       // workgroups per CU of the invariant z-pass
  double *inv_out[3];   // k_c2r_invariants<float> MODE 0: the fp64 invariant rows (pitch inv_pitch) -- fp32 rows cannot hold them
  long long inv_pitch;
};
int pf_launch_c2r(int field_bytes, int n, const PfC2RParams &p, hipStream_t st);
// six components (njobs == 6) -> the three invariants of the tensor: fp64 fields into job[0..2].out, fp32 fields into inv_out
// mode 1: nothing stored, p.acc -= 2 phi2_ab h_ab with h_ab read from job[c].out (the 3LPT(b) source, src/LPT.c:134-137)
int pf_launch_c2r_invariants(int field_bytes, int n, const PfC2RParams &p, hipStream_t st, int mode = 0);

struct PfR2CParams {
  const void *in;       // real rows, pitch in_pitch reals
  void *out;            // complex rows, pitch out_pitch complex (may alias in)
  long long nlines, in_pitch, out_pitch;
  const void *tw;
};
int pf_launch_r2c(int field_bytes, int n, const PfR2CParams &p, hipStream_t st);

// ---- grid sizes that are not a power of two (pf_mixed_kernels.hip): the same three passes with a run-time stage plan ----
#define PF_MIXED_MAX_STAGES 12
struct PfMixedPlan { int n, nstages; int radix[PF_MIXED_MAX_STAGES]; unsigned magic[PF_MIXED_MAX_STAGES]; unsigned magic_in, magic_out; /* ceil(2^32 / el_len) of the launch's input and output layouts (strided passes) */ };
bool pf_mixed_plan(int n, bool allow4, PfMixedPlan *pl);
bool pf_mixed_supported(int n);   // n = 8 m with m = 2^a 3^b 5^c, up to 2048: strided passes on n, z-passes on n / 2
int pf_launch_mixed_strided(int field_bytes, int n, int dir, const PfStridedParams &p, hipStream_t st);
int pf_launch_mixed_c2r(int field_bytes, int n, const PfC2RParams &p, hipStream_t st);
int pf_launch_mixed_r2c(int field_bytes, int n, const PfR2CParams &p, hipStream_t st);
bool pf_mixed_invariants_supported(int field_bytes, int n);  // six lines of a row fit one workgroup
int pf_launch_mixed_c2r_invariants(int field_bytes, int n, const PfC2RParams &p, hipStream_t st, int mode);
bool pf_c2r_invariants_preferred(int field_bytes, int n);   // ... and the sweep uses it (pf_fft_kernels.hip)
bool pf_mixed_plan_compiled_in(int n);                       // n is one of PF_MIXED_CT_SIZES (pf_mixed_kernels.hip)
bool pf_c2r_invariants_supported(int field_bytes, int n);    // pf_launch_c2r_invariants takes rows of n points (either kind of plan)

// ---- per-cell kernels (pf_cell_kernels.hip) ----
#define PF_KNOT_CAP 512  // knots per spline at most; the five arrays of a spline lie PF_KNOT_CAP doubles apart (y = x + PF_KNOT_CAP ...: spline_for)
struct PfSplineDev {
  const double *x, *y, *c, *b, *d;  // knots, GSL cspline c_i, and the per-interval b_i, d_i (pf_spline_bd)
  int n;
  // pf_gtab.h: header + records of the polynomial table of this spline and its start table; null: none (table refused or PF_GTAB=0)
  const double *gt;
  const unsigned short *gt_lut;
};
// TABULATED_CT table of one radius on the device (pf_collapse_core.h pf_ct_view)
struct PfCtDev {
  double *delta, *y, *b, *c, *d;   // [100], 4 x [50*50*100]
  const double *alpha, *gamma;     // [98] factors of the shared tridiagonal system
  double ampl;
  int model;                       // what fills the table: 0 ell_classic + InverseGrowingMode, 1 ell_sng (pf_sng_core.h)
  int flavour;                     // how the solve reads it: 0 BILINEAR_SPLINE, 1 TRILINEAR, 2 ALL_SPLINE (pf_collapse_core.h)
  double sng_cosmo[7], sng_Din;    // ELL_SNG: Omega0, OmegaLambda, OmegaRad, OmegaK, FR0, H_over_c, size; GrowingMode at a = 1e-5 for this radius
};
struct PfCollapseParams {
  const void *h[6];     // Hessian fields, type F, rows of pitch reals
  long long pitch;
  long long nrows;      // nx_local * n
  int n;                // row length
  void *fmax;           // products.Fmax: float, or double when prod_f64 (-DDOUBLE_PRECISION_PRODUCTS, fp64 fields only)
  int prod_f64;
  int *rmax;
  int ismooth;
  PfSplineDev spline;
  double *partials;     // [2*nblocks]: sum delta, sum delta^2 per block
  int nblocks;
  int fast;             // 1: sincos/cbrt/exp10 forms of the transcendental hot spots (pf_collapse_core.h)
  int no_lut;           // 1: plain bisection in the spline lookup (PF_SPLINE_LUT=0)
  int invariants;       // 1: h[0..2] hold mu1, mu2, mu3 (k_c2r_invariants), h[3..5] unused
  int tabulated;        // 1: F from the collapse-time table `ct` (TABULATED_CT build) instead of the direct solve
  int sng;              // 1: ELL_SNG without a table -- one RKF45 integration per cell, cosmology and D_in in ct.sng_*
  // sources: the same pass also forms the 2LPT / 3LPT sources of the cell from the six components it holds (what
  // k_lpt_sources computes, src/LPT.c:64-93) -- fields of type F in src[0..2], per-block sums of S2 in src_partials
  int sources;
  void *src[3];
  double *src_partials;
  PfCtDev ct;
};
int pf_launch_ct_build(const PfSplineDev &sp, const PfCtDev &ct, int fast, int compute_table, hipStream_t st);
int pf_launch_collapse(int field_bytes, const PfCollapseParams &p, hipStream_t st);
int pf_launch_final_sum(const double *partials, int nblocks, double *out2, hipStream_t st);
int pf_launch_collapse_cells(const double *d, size_t count, PfSplineDev s, double *F, int fast, hipStream_t st);

struct PfLptSrcParams {
  const void *h[6];
  void *s2, *s3a, *s3b;  // real rows, same pitch
  long long pitch, nrows;
  int n;
  double *partials;      // per-block sum of s2 (its spectrum's DC), [nblocks]
  int nblocks;
};
int pf_launch_lpt_sources(int field_bytes, const PfLptSrcParams &p, hipStream_t st);

struct PfLptAccParams {
  const void *h[6], *phi2[6];
  void *s3b;
  long long pitch, nrows;
  int n;
};
int pf_launch_lpt_accum(int field_bytes, const PfLptAccParams &p, hipStream_t st);

int pf_launch_sum1(const double *partials, int nblocks, double scale, double *out, hipStream_t st);
// pb: bytes of a PRODFLOAT (4, or 8 for -DDOUBLE_PRECISION_PRODUCTS): the type of fmax[] and of the twelve vel12 columns
int pf_launch_fill_products(void *fmax, int *rmax, void *vel12, size_t ncell, int pb, hipStream_t st);
int pf_launch_apply_growth(int fb, const void *in, void *out, int n, int nyl, int nzh, int nzp, int y0, const double *T, int nk,
                           double logkmin, double dlogk, double sign, hipStream_t st);
int pf_launch_pack_products(int pb, const void *fmax, const int *rmax, const void *vel12, size_t ncell_total,
                            size_t first, size_t count, char *aos, size_t stride, int off_rmax, int off_fmax,
                            const int off_vel[4], hipStream_t st);
int pf_launch_fmax_pdf(const void *fmax, size_t ncell, unsigned long long *hist, int pb, hipStream_t st);
// copy + pitch/precision conversion of a half-spectrum between the boundary layout
// (fp64, rows of nzh complex) and the internal one (F, rows of nzp complex)
int pf_launch_spec_import(int field_bytes, const double *src, void *dst, long long nrows, int nzh, int nzp, hipStream_t st);
int pf_launch_spec_export(int field_bytes, const void *src, double *dst, long long nrows, int nzh, int nzp, hipStream_t st);
int pf_launch_spec_import_t(int fb, const double *src, void *dst, int n, int nyl, int nzh, int nzp, hipStream_t st);
int pf_launch_spec_export_t(int fb, const void *src, double *dst, int n, int nyl, int nzh, int nzp, hipStream_t st);
int pf_launch_real_import(int field_bytes, const double *src, void *dst, long long nrows, int n, long long pitch, hipStream_t st);
int pf_launch_real_export(int field_bytes, const void *src, double *dst, long long nrows, int n, long long pitch, hipStream_t st);
// general grid sizes (library-FFT path)
int pf_launch_gen_filter(const void *in, void *out, int n, int a, int b, double rs, double growth, const double *T, int nk, double logkmin,
                         double dlogk, double sign, double norm, hipStream_t st);
int pf_launch_real_to_col(const double *src, void *dst, size_t ncell, int pb, hipStream_t st);
int pf_launch_scale_real(double *f, size_t ncell, double s, hipStream_t st);
// pf_gfft.hip: chirp-z (Bluestein) 3-D transforms for any even n (double precision, natural layouts: spectrum [n][n][n/2+1], real [n][n][n])
int pf_gfft_create(int n, hipStream_t st, void **c2r, void **r2c);
int pf_gfft_c2r(void *plan, void *spec, void *real);   // unnormalised, out of place; may destroy spec
int pf_gfft_r2c(void *plan, void *real, void *spec);   // unnormalised, out of place
void pf_gfft_destroy(void *plan);
int pf_launch_debug_math(int which, const double *a, const double *b, size_t count, double *out, hipStream_t st);
// pf_select_sort.hip
int pf_select_sort_device(const float *fmax, size_t ncell, float flast, unsigned int **d_idx, float **d_f, size_t *count, hipStream_t st);
int pf_launch_block_vec3(const float *vel12, size_t ncell, int o, size_t first, size_t count, float *out, hipStream_t st);
int pf_launch_block_id(int id_bytes, unsigned long long global_first, size_t count, void *out, hipStream_t st);
int pf_launch_to_blocks(int field_bytes, const void *src, void *dst, int nxl, int n, int nyl, int nzp, int back, hipStream_t st);
int pf_launch_extract_dc(int field_bytes, const void *spec, double scale, double *out, hipStream_t st);

// ---- synthetic density (pf_synth.hip) ----
int pf_launch_white(int field_bytes, void *real, long long nrows, long long row0, int n, long long pitch,
                    uint64_t seed, hipStream_t st);
struct PfShapeParams {
  void *spec;            // internal layout, rows (x, y_local) of nzp complex
  int n, nzp, nyl, y0;   // y-slab
  double slope, scale;   // amplitude k^(slope/2) * scale ; scale applied only when apply != 0
  double *partials;      // per-block weighted sum |dk|^2
  int nblocks;
  int mode;              // 0: apply shape + accumulate power; 1: multiply by *dscale
  const double *dscale;
};
int pf_launch_shape(int field_bytes, const PfShapeParams &p, hipStream_t st);
int pf_launch_stream(int kind, const void *src, void *dst, size_t bytes, float *sink, hipStream_t st);  // 0 read, 1 write, 2 copy
int pf_launch_sigma_scale(const double *power_sum, double sigma0, double n3, double *dscale, hipStream_t st);

// ---- GenIC on the device (pf_genic.hip) ----
int pf_genic_launch(int field_bytes, void *dk, int n, int nzp, int nyl, int y0, const pf_genic_params *p, hipStream_t st,
                    unsigned int **seed_dev_out);
