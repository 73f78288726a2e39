// pf_cell_kernels.hip -- per-cell kernels of the path (gfx950): collapse-time
// solve with running max (src/collapse_times.c:431-673), LPT sources
// (src/LPT.c:64-93, :134-137), products init/pack (src/collapse_times.c:461-492,
// src/fmax-pfft.c:563-631), Fmax PDF (src/fmax.c:509-550), layout converters.
#include <type_traits>

#include "pf_internal.h"
#include "pf_fft_core.h"
#include "pf_collapse_core.h"
#include "pf_sng_core.h"

#define PF_CELL_BLOCK 256
// the solve on the invariants (k_collapse_inv) has its own workgroup size and wave budget: its LDS (the 38 KB polynomial table) lets
// four workgroups share a CU, its registers decide how many waves those may hold (A/B knobs: -DPF_SOLVE_INV_BLOCK=320 -DPF_SOLVE_INV_WAVES=5)
#ifndef PF_SOLVE_INV_BLOCK
#define PF_SOLVE_INV_BLOCK 512  // (round 5: 15.9 -> 15.4 ms per launch against 256; 1024: 17.0; 640 threads at five waves per SIMD: 21.0)
#endif
#ifndef PF_SOLVE_INV_WAVES
#define PF_SOLVE_INV_WAVES 4
#endif
#define PF_MAX_KNOTS 512
#ifndef PF_C3_TABLE
#define PF_C3_TABLE 1  // (0 in an A/B build: the cosine triple from its single degree-22 polynomial)
#endif

// deterministic block reduction of two doubles: wave shuffle, then thread 0 sums the waves in order
__device__ __forceinline__ void pf_block_sum2(double &a, double &b, double *sh /* 2*nwaves */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { sh[2 * w] = a; sh[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
    for (int i = 0; i < nw; i++) { sa += sh[2 * i]; sb += sh[2 * i + 1]; }
    a = sa; b = sb;
  }
}
// The same by GROUPS of PF_CELL_BLOCK threads: a workgroup of several such groups leaves one pair of partial sums per group, in the
// first thread of the group -- the very sums, in the very order, that workgroups of PF_CELL_BLOCK threads on a grid with as many
// threads leave (the grid-stride walk hands a thread the same cells either way): TrueVariance does not depend on the workgroup size
__device__ __forceinline__ void pf_group_sum2(double &a, double &b, double *sh /* 2*nwaves */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[2 * w] = a; sh[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x % 256 == 0) {
    double sa = 0, sb = 0;
    for (int i = w; i < w + 4; i++) { sa += sh[2 * i]; sb += sh[2 * i + 1]; }
    a = sa; b = sb;
  }
}

// K6: one thread per cell, grid-stride.  fp64 math whatever the field precision.
// INV: h[0..2] hold the invariants mu1, mu2, mu3 of the tensor (written by k_c2r_invariants) instead of its components
// SNG: ell() of an ELL_SNG build without TABULATED_CT (src/collapse_times.c:416-426): the ellipsoid of every cell is
// integrated on its own (pf_sng_core.h)
// K7 (src/LPT.c:64-93): the 2LPT source and the two 3LPT sources that need only the first-order Hessian, from one
// cell's six components {11,22,33,12,13,23}; one definition for k_lpt_sources and for the solve that forms them in passing
PF_HD void pf_lpt_sources_cell(const double d[6], double &src2, double &src31, double &src32) {
  src2 = d[0] * d[1] + d[0] * d[2] + d[1] * d[2] - d[3] * d[3] - d[4] * d[4] - d[5] * d[5];
  src31 = 3.0 * (d[0] * (d[1] * d[2] - d[5] * d[5]) - d[3] * (d[3] * d[2] - d[4] * d[5]) + d[4] * (d[3] * d[5] - d[4] * d[1]));
  src32 = 2.0 * (d[0] + d[1] + d[2]) * src2;
}
// SRC: the pass of the last radius of a sweep that is followed by compute_LPT_displacements -- the cell's six components are
// in registers anyway, so its three LPT sources are written here and k_lpt_sources (six more field reads) is not run
// PR: PRODFLOAT, the type of products.Fmax (float; double in a -DDOUBLE_PRECISION_PRODUCTS build, src/pinocchio.h:219-225)
template <typename F, bool FAST, bool TAB = false, bool INV = false, bool SNG = false, bool SRC = false, int FLAV = 0, typename PR = float>
__device__ __forceinline__ void pf_collapse_body(const PfCollapseParams &p) {
  // GT: the fast flavour's inverse growing mode from the polynomial table of the spline (pf_gtab.h), staged in LDS in place
  // of the knot arrays; D outside the table (or no table: refused knots, PF_GTAB=0) takes the series forms on the knot arrays --
  // in global memory (L2) when the table has their place in LDS, staged as before otherwise, then with plain bisection (the
  // start table of the knots has no room beside the larger array)
  constexpr bool GT = FAST && !TAB && !SNG;
  constexpr int SK = TAB ? PF_CT_NBINS_D : GT ? (PF_GT_MAX_INT + 1) * PF_GT_REC : 5 * PF_MAX_KNOTS;
  static_assert(!GT || SK >= 5 * PF_MAX_KNOTS, "the knot arrays fit the table's place");
  __shared__ double sk[SK];
  __shared__ double red[2 * ((PF_SOLVE_INV_BLOCK > PF_CELL_BLOCK ? PF_SOLVE_INV_BLOCK : PF_CELL_BLOCK) / 64)];
  __shared__ unsigned short slut[TAB ? 1 : GT ? PF_GT_MAX_BINS : PF_SPLINE_LUT_BINS];
  // C3: the fast flavour's cosine triple from its table of short polynomials (pf_c3tab.h; 2 KB: with the 34.6 + 4 KB above a
  // workgroup stays under the 40 KB that let four share a CU)
  constexpr bool C3 = FAST && !SNG && PF_C3_TABLE;
  __shared__ double sc3[C3 ? PF_C3_DOUBLES : 1];
  if (C3) for (int i = threadIdx.x; i < PF_C3_DOUBLES; i += blockDim.x) sc3[i] = pf_c3_tab[i];
  const int nk = p.spline.n;
  const bool gt = GT && p.spline.gt != nullptr;  // uniform
  pf_spline_view sv;
  if (gt) {
    const double *g = p.spline.gt;
    const int nint = (int)g[0], nbins = (int)g[1];
    for (int i = threadIdx.x; i < (nint + 1) * PF_GT_REC; i += blockDim.x) sk[i] = g[PF_GT_HEADER + i];
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) slut[i] = p.spline.gt_lut[i];
    sv.gt.bin0 = (unsigned)g[2]; sv.gt.lo_all = g[3]; sv.gt.hi_all = g[4];
  } else if (TAB) {
    for (int i = threadIdx.x; i < PF_CT_NBINS_D; i += blockDim.x) sk[i] = p.ct.delta[i];
  } else {
    for (int i = threadIdx.x; i < nk; i += blockDim.x) {
      sk[i] = p.spline.x[i];
      sk[PF_MAX_KNOTS + i] = p.spline.y[i];
      sk[2 * PF_MAX_KNOTS + i] = p.spline.c[i];
      sk[3 * PF_MAX_KNOTS + i] = p.spline.b[i];
      sk[4 * PF_MAX_KNOTS + i] = p.spline.d[i];
    }
  }
  __syncthreads();
  double lut_x0 = 0.0, lut_inv_w = 0.0;
  int lut_direct = 0;
  if (!TAB && !GT) {  // interval-search start table over the knots now in LDS: the direct form if no bin holds two knots, else the walk form
    pf_spline_lut_geometry(sk, nk, true, lut_x0, lut_inv_w);
    for (int b = threadIdx.x; b < PF_SPLINE_LUT_BINS; b += blockDim.x) slut[b] = pf_spline_lut_entry(sk, nk, b, lut_x0, lut_inv_w, true);
    __syncthreads();
    int crowded = 0;
    for (int b = threadIdx.x; b + 1 < PF_SPLINE_LUT_BINS; b += blockDim.x) crowded |= (int)slut[b + 1] - (int)slut[b] > 1;
    lut_direct = __syncthreads_or(crowded) ? 0 : 1;
    if (!lut_direct) {
      pf_spline_lut_geometry(sk, nk, false, lut_x0, lut_inv_w);
      for (int b = threadIdx.x; b < PF_SPLINE_LUT_BINS; b += blockDim.x) slut[b] = pf_spline_lut_entry(sk, nk, b, lut_x0, lut_inv_w, false);
      __syncthreads();
    }
  }
  pf_ct_view tv;
  tv.delta = sk; tv.y = p.ct.y; tv.b = p.ct.b; tv.c = p.ct.c; tv.d = p.ct.d; tv.ampl = p.ct.ampl;
  if (GT) {
    // Scalar registers are what this kernel is short of (the fp64 constants of the series live there): with the table's place in
    // LDS a compile-time address, "no table" an empty range [inf, 0) instead of a flag, the range's ends in vector registers and
    // the five knot arrays of the rare series path addressed from one pointer (spline_for lays them out PF_KNOT_CAP apart), the
    // loop keeps what is left without the ~27 v_readlane / v_writelane per cell that reloading spilled scalars cost (round 4).
    sv.gt.rec = sk; sv.gt.lut = slut;
    if (!gt) { sv.gt.lo_all = HUGE_VAL; sv.gt.hi_all = 0.0; sv.gt.bin0 = 0; }
    asm volatile("" : "+v"(sv.gt.lo_all), "+v"(sv.gt.hi_all));
  }
  if (gt) { sv.x = p.spline.x; sv.y = sv.x + PF_KNOT_CAP; sv.c = sv.x + 2 * PF_KNOT_CAP; sv.b = sv.x + 3 * PF_KNOT_CAP; sv.d = sv.x + 4 * PF_KNOT_CAP; }
  else if (!TAB) { sv.x = sk; sv.y = sk + PF_MAX_KNOTS; sv.c = sk + 2 * PF_MAX_KNOTS; sv.b = sk + 3 * PF_MAX_KNOTS; sv.d = sk + 4 * PF_MAX_KNOTS; }
  else { sv.x = sk; sv.y = sv.c = sv.b = sv.d = sk; }
  sv.n = nk;
  const double *c3tab = C3 ? sc3 : nullptr;
  if (C3) __builtin_assume(c3tab != nullptr);  // (an LDS address is not known to differ from null: without this both forms are kept, behind a branch)
  sv.c3tab = c3tab;
  pf_sng_cosmo sc;
  if (SNG) {
    sc.Omega0 = p.ct.sng_cosmo[0]; sc.OmegaLambda = p.ct.sng_cosmo[1]; sc.OmegaRad = p.ct.sng_cosmo[2]; sc.OmegaK = p.ct.sng_cosmo[3];
    sc.FR0 = p.ct.sng_cosmo[4]; sc.H_over_c = p.ct.sng_cosmo[5]; sc.size = p.ct.sng_cosmo[6];
  }
  if (!TAB && !GT && !p.no_lut) { sv.lut = slut; sv.lut_inv_w = lut_inv_w; sv.lut_x0 = lut_x0; sv.lut_direct = lut_direct; sv.x_first = sk[0]; sv.x_last = sk[nk - 1]; }

  const F *__restrict__ h0 = (const F *)p.h[0], *__restrict__ h1 = (const F *)p.h[1],
          *__restrict__ h2 = (const F *)p.h[2], *__restrict__ h3 = (const F *)p.h[3],
          *__restrict__ h4 = (const F *)p.h[4], *__restrict__ h5 = (const F *)p.h[5];
  const long long ncell = p.nrows * p.n;
  double sum = 0.0, sum2 = 0.0, sum_src = 0.0;
  F *__restrict__ s2 = (F *)p.src[0], *__restrict__ s3a = (F *)p.src[1], *__restrict__ s3b = (F *)p.src[2];
  // grid-stride walk over the cells with (row, column) carried along: one 64-bit division per thread instead of one per
  // cell (the division and the address arithmetic behind it were 36 of the ~650 vector instructions a cell costs)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int srow = (int)(stride / p.n), scol = (int)(stride - (long long)srow * p.n);
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int row = (int)(i0 / p.n), col = (int)(i0 - (long long)row * p.n);
#ifndef PF_SOLVE_PREFETCH
#define PF_SOLVE_PREFETCH 0  // (1 in an A/B build: see below -- measured: no gain, 16.2 against 15.9 ms per launch)
#endif
  // PF_SOLVE_PREFETCH=1 (A/B, round 5): the inputs of a thread's NEXT cell are requested before it solves the current one.  No gain
  // (16.2 against 15.9 ms per launch at 1024^3): the wait at the top of the next iteration then also covers this iteration's two
  // conditional stores -- the counter is in order and the compiler cannot count stores behind a branch -- which costs what the
  // early request saves.  products[].Fmax is float, compared after promotion (quirk Q2); init -10 / -1 at ismooth 0.
  PR *__restrict__ fmax = (PR *)p.fmax;
  double dn[6] = {0, 0, 0, 0, 0, 0};
  PR foldn = (PR)-10.0f;
  long long an = 0;
  auto fetch = [&](long long i, long long a) {
#if PF_NT  // (read once per radius: streaming loads, pf_fft_core.h)
    dn[0] = (double)__builtin_nontemporal_load(&h0[a]); dn[1] = (double)__builtin_nontemporal_load(&h1[a]); dn[2] = (double)__builtin_nontemporal_load(&h2[a]);
    if (!INV) { dn[3] = (double)__builtin_nontemporal_load(&h3[a]); dn[4] = (double)__builtin_nontemporal_load(&h4[a]); dn[5] = (double)__builtin_nontemporal_load(&h5[a]); }
#else
    dn[0] = (double)h0[a]; dn[1] = (double)h1[a]; dn[2] = (double)h2[a];
    if (!INV) { dn[3] = (double)h3[a]; dn[4] = (double)h4[a]; dn[5] = (double)h5[a]; }
#endif
    foldn = p.ismooth ? fmax[i] : (PR)-10.0f;
    an = a;
  };
  if (PF_SOLVE_PREFETCH && i0 < ncell) {
    if (col >= p.n) { col -= p.n; row++; }
    fetch(i0, (long long)row * p.pitch + col);
  }
  for (long long i = i0; i < ncell; i += stride) {
    if (!PF_SOLVE_PREFETCH) {
      if (col >= p.n) { col -= p.n; row++; }
      fetch(i, (long long)row * p.pitch + col);
    }
    double d[6];
#pragma unroll
    for (int k = 0; k < 6; k++) d[k] = dn[k];
    const PR fold = foldn;
    const long long a = an;
    row += srow; col += scol;
    if (PF_SOLVE_PREFETCH && i + stride < ncell) {
      if (col >= p.n) { col -= p.n; row++; }
      fetch(i + stride, (long long)row * p.pitch + col);
    }
    if (SRC) {
      double src2, src31, src32;
      pf_lpt_sources_cell(d, src2, src31, src32);
      s2[a] = (F)src2;
      s3a[a] = (F)src31;
      s3b[a] = (F)src32;
      sum_src += (double)(F)src2;
    }
    const double delta = INV ? d[0] : d[0] + d[1] + d[2];
    sum += delta;
    sum2 += delta * delta;
    double lam[3];
    bool have_lam;
    if (INV) {
      const double third = d[0] * (1.0 / 3.0);  // an exactly isotropic tensor: its diagonal is not stored
      const double diag[3] = {third, third, third};
      have_lam = pf_eigen_from_invariants<FAST>(d[0], d[1], d[2], diag, lam, c3tab);
    } else {
      have_lam = pf_ordered_eigenvalues<FAST>(d, lam, c3tab);
    }
    // TABULATED_CT: the same eigenvalues, then the table instead of ell() (src/collapse_times.c:749)
    const double Fnew = !have_lam ? -10.0
                        : TAB   ? pf_interpolate_collapse_time_as<FLAV>(tv, lam[0], lam[1], lam[2])
                        : SNG   ? pf_ell_sng_F(lam[0], lam[1], lam[2], p.ct.sng_Din, sc)
                                : pf_ell<FAST>(sv, lam[0], lam[1], lam[2]);
    if ((double)fold < Fnew) {
      fmax[i] = (PR)Fnew;
      p.rmax[i] = p.ismooth;
    } else if (!p.ismooth) {
      fmax[i] = (PR)-10.0f;
      p.rmax[i] = -1;
    }
  }
  static_assert(PF_CELL_BLOCK == 256, "pf_group_sum2 sums groups of four waves");
  pf_group_sum2(sum, sum2, red);
  if (threadIdx.x % PF_CELL_BLOCK == 0) {
    const int slot = blockIdx.x * (blockDim.x / PF_CELL_BLOCK) + threadIdx.x / PF_CELL_BLOCK;
    p.partials[2 * slot] = sum;
    p.partials[2 * slot + 1] = sum2;
  }
  if (SRC) {  // the same grid and walk as k_lpt_sources: the same partial sums
    double dummy = 0.0;
    __syncthreads();
    pf_block_sum2(sum_src, dummy, red);
    if (threadIdx.x == 0) p.src_partials[blockIdx.x] = sum_src;
  }
}

// (four waves per SIMD = four 256-thread workgroups per CU, which the grid of 8 per CU is sized for: at most 128 VGPRs)
template <typename F, bool FAST, typename PR = float>
__global__ void __launch_bounds__(PF_CELL_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) k_collapse(const PfCollapseParams p) { pf_collapse_body<F, FAST, false, false, false, false, 0, PR>(p); }
// FLAV: the table interpolation of the build (0 BILINEAR_SPLINE, 1 -DTRILINEAR, 2 -DALL_SPLINE)
template <typename F, bool FAST, int FLAV, typename PR = float>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_collapse_tab(const PfCollapseParams p) { pf_collapse_body<F, FAST, true, false, false, false, FLAV, PR>(p); }
template <typename F, bool FAST, typename PR> static void pf_launch_collapse_tab(const PfCollapseParams &p, hipStream_t st) {
  switch (p.ct.flavour) {
    case 1: hipLaunchKernelGGL((k_collapse_tab<F, FAST, 1, PR>), dim3(p.nblocks), dim3(PF_CELL_BLOCK), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_collapse_tab<F, FAST, 2, PR>), dim3(p.nblocks), dim3(PF_CELL_BLOCK), 0, st, p); break;
    default: hipLaunchKernelGGL((k_collapse_tab<F, FAST, 0, PR>), dim3(p.nblocks), dim3(PF_CELL_BLOCK), 0, st, p); break;
  }
}
template <typename F, bool FAST, typename PR = float>
__global__ void __launch_bounds__(PF_CELL_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) k_collapse_src(const PfCollapseParams p) { pf_collapse_body<F, FAST, false, false, false, true, 0, PR>(p); }
// ELL_SNG per cell: thousands of dependent steps per thread, lanes of a wave finish at different times -- small workgroups
template <typename F, bool FAST, typename PR = float>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_collapse_sng(const PfCollapseParams p) { pf_collapse_body<F, FAST, false, false, true, false, 0, PR>(p); }
// the solve on the three invariants per cell that k_c2r_invariants leaves in h[0..2] (fp64 fields)
template <bool FAST, typename PR = float>
__global__ void __launch_bounds__(PF_SOLVE_INV_BLOCK) __attribute__((amdgpu_waves_per_eu(PF_SOLVE_INV_WAVES))) k_collapse_inv(const PfCollapseParams p) { pf_collapse_body<double, FAST, false, true, false, false, 0, PR>(p); }

// ---- the solve on the invariants, default arithmetic, with the one-root lanes run compacted (round 5) ----
// The reduced cubic of ell_classic has one real root for about one cell in eight (pf_ell_setup: kind 1) and three for the rest; the
// two branches share nothing, and practically every wave holds cells of both kinds, so every wave used to run both (~60 of its ~400
// vector instructions per cell for one lane in eight).  Here a lane whose cell needs the one-root branch puts the cell aside -- its
// eigenvalues and index into a queue of its wave in LDS -- and goes on as if it had no cell; when 64 cells have gathered the wave
// runs the one-root branch, the spherical correction, the inverse growing mode and the comparison for all of them at once, every
// lane busy.  The same per-cell functions on the same inputs in another lane at another time: bit for bit the results of
// pf_collapse_body.  One workgroup of 1024 threads per CU: ONE copy of the polynomial tables (40.8 KB) beside the sixteen queues
// (16 x 3.5 KB), sixteen waves per CU as before.
#ifndef PF_SOLVE_QUEUE
#define PF_SOLVE_QUEUE 0  // 1 (A/B build): this kernel for the default arithmetic.  MEASURED SLOWER: 18.6 against 15.7 ms per launch at 1024^3 (profiles/r05_notes.md), not used
#endif
#ifndef PF_SQ_BLOCK
#define PF_SQ_BLOCK 512   // two workgroups per CU, each with its tables (40.8 KB) and eight queues (28 KB): eight waves start and end together instead of sixteen (A/B: 1024)
#endif
#define PF_SQ_SLOTS 128
template <typename PR>
__global__ void __launch_bounds__(PF_SQ_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) k_collapse_invq(const PfCollapseParams p) {
  constexpr int SK = (PF_GT_MAX_INT + 1) * PF_GT_REC;
  __shared__ double sk[SK];
  __shared__ double red[2 * (PF_SQ_BLOCK / 64)];
  __shared__ unsigned short slut[PF_GT_MAX_BINS];
  __shared__ double sc3[PF_C3_DOUBLES];
  __shared__ double ql[PF_SQ_BLOCK / 64][3][PF_SQ_SLOTS];     // eigenvalues of the cells put aside, per wave
  __shared__ unsigned int qi[PF_SQ_BLOCK / 64][PF_SQ_SLOTS];  // ... and their index
  for (int i = threadIdx.x; i < PF_C3_DOUBLES; i += blockDim.x) sc3[i] = pf_c3_tab[i];
  const bool gt = p.spline.gt != nullptr;  // uniform
  pf_spline_view sv;
  if (gt) {
    const double *g = p.spline.gt;
    const int nint = (int)g[0], nbins = (int)g[1];
    for (int i = threadIdx.x; i < (nint + 1) * PF_GT_REC; i += blockDim.x) sk[i] = g[PF_GT_HEADER + i];
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) slut[i] = p.spline.gt_lut[i];
    sv.gt.bin0 = (unsigned)g[2]; sv.gt.lo_all = g[3]; sv.gt.hi_all = g[4];
  }
  __syncthreads();
  sv.gt.rec = sk; sv.gt.lut = slut;
  if (!gt) { sv.gt.lo_all = HUGE_VAL; sv.gt.hi_all = 0.0; sv.gt.bin0 = 0; }  // no table: an empty range (the series forms on the knot arrays)
  asm volatile("" : "+v"(sv.gt.lo_all), "+v"(sv.gt.hi_all));
  sv.x = p.spline.x; sv.y = sv.x + PF_KNOT_CAP; sv.c = sv.x + 2 * PF_KNOT_CAP; sv.b = sv.x + 3 * PF_KNOT_CAP; sv.d = sv.x + 4 * PF_KNOT_CAP;
  sv.n = p.spline.n;
  const double *c3tab = sc3;
  __builtin_assume(c3tab != nullptr);
  sv.c3tab = c3tab;

  const double *__restrict__ h0 = (const double *)p.h[0], *__restrict__ h1 = (const double *)p.h[1], *__restrict__ h2 = (const double *)p.h[2];
  PR *__restrict__ fmax = (PR *)p.fmax;
  const long long ncell = p.nrows * p.n;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *q1 = ql[wv][0], *q2 = ql[wv][1], *q3 = ql[wv][2];
  unsigned int *qx = qi[wv];
  int qcount = 0;  // cells waiting in this wave's queue (uniform)
  double sum = 0.0, sum2 = 0.0;
  // everything behind the root of the cubic, for one cell per lane: src/collapse_times.c:395-427 and the update of :749-773
  auto finish = [&](bool live, double ell, double l1, double l2, double l3, long long i, PR fold) {
    if (!live) return;
    const double bc = pf_ell_finish<true>(ell, l1, l2, l3);
    const double Fnew = bc > 0.0 ? 1. + pf_inverse_growing_mode<true>(sv, bc) : 0.0;
    if ((double)fold < Fnew) {
      fmax[i] = (PR)Fnew;
      p.rmax[i] = p.ismooth;
    } else if (!p.ismooth) {
      fmax[i] = (PR)-10.0f;
      p.rmax[i] = -1;
    }
  };
  // the cells put aside, 64 at a time (or what is left at the end): lane L takes entry first + L
  auto drain = [&](int first, int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool live = lane < count;
    const int e = first + (live ? lane : 0);
    const double l1 = q1[e], l2 = q2[e], l3 = q3[e];
    const long long i = (long long)qx[e];
    const PR fold = (live && p.ismooth) ? fmax[i] : (PR)-10.0f;
    double ell = 0.0;
    pf_cubic c;
    const int kind = pf_ell_setup<true>(l1, l2, l3, ell, c);   // (the same setup again: kind 1, the same cubic)
    if (kind == 1) ell = pf_ell_one_root<true>(c);
    finish(live, ell, l1, l2, l3, i, fold);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // grid-stride walk, a wave at a time (every lane of a wave makes the same number of trips: the queue is the wave's)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int srow = (int)(stride / p.n), scol = (int)(stride - (long long)srow * p.n);
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int row = (int)(i0 / p.n), col = (int)(i0 - (long long)row * p.n);
  for (long long i = i0; i - lane < ncell; i += stride, row += srow, col += scol) {
    if (col >= p.n) { col -= p.n; row++; }
    const bool active = i < ncell;
    const long long a = (long long)row * p.pitch + col;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    PR fold = (PR)-10.0f;
    if (active) {
#if PF_NT
      d0 = __builtin_nontemporal_load(&h0[a]); d1 = __builtin_nontemporal_load(&h1[a]); d2 = __builtin_nontemporal_load(&h2[a]);
#else
      d0 = h0[a]; d1 = h1[a]; d2 = h2[a];
#endif
      if (p.ismooth) fold = fmax[i];
      sum += d0;
      sum2 += d0 * d0;
    }
    double lam[3] = {0.0, 0.0, 0.0};
    const double third = d0 * (1.0 / 3.0);  // an exactly isotropic tensor: its diagonal is not stored
    const double diag[3] = {third, third, third};
    const bool have_lam = active && pf_eigen_from_invariants<true>(d0, d1, d2, diag, lam, c3tab);
    double ell = 0.0;
    pf_cubic c;
    int kind = 0;
    if (have_lam) kind = pf_ell_setup<true>(lam[0], lam[1], lam[2], ell, c);
    const bool aside = have_lam && kind == 1;
    const unsigned long long m = __ballot(aside);
    if (m) {  // (uniform)
      if (aside) {
        const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        q1[pos] = lam[0]; q2[pos] = lam[1]; q3[pos] = lam[2]; qx[pos] = (unsigned int)i;
      }
      qcount += __popcll(m);
    }
    if (active && !have_lam) {  // the -10 sentinel of inverse_collapse_time: never above a stored Fmax, only the first radius writes
      if (!p.ismooth) { fmax[i] = (PR)-10.0f; p.rmax[i] = -1; }
    }
    if (have_lam && kind == 2) ell = pf_ell_three_roots<true>(c, c3tab);
    finish(have_lam && !aside, ell, lam[0], lam[1], lam[2], i, fold);
    if (qcount >= 64) { qcount -= 64; drain(qcount, 64); }
  }
  if (qcount > 0) drain(0, qcount);
  pf_group_sum2(sum, sum2, red);
  if (threadIdx.x % PF_CELL_BLOCK == 0) {
    const int slot = blockIdx.x * (blockDim.x / PF_CELL_BLOCK) + threadIdx.x / PF_CELL_BLOCK;
    p.partials[2 * slot] = sum;
    p.partials[2 * slot + 1] = sum2;
  }
}

// initialize_collapse_times (src/collapse_times.c:956-972): CT_table[i] = ell(ismooth, l1, l2, l3) on the
// (delta, x, y) grid, i = id + 100 * (ix + 50 * iy)
template <bool FAST>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_ct_table(PfSplineDev s, PfCtDev ct) {
  pf_spline_view sv;
  sv.x = s.x; sv.y = s.y; sv.c = s.c; sv.b = s.b; sv.d = s.d; sv.n = s.n;
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const int total = PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int id = i % PF_CT_NBINS_D;
    const int ix = (i / PF_CT_NBINS_D) % PF_CT_NBINS_XY;
    const int iy = i / PF_CT_NBINS_D / PF_CT_NBINS_XY;
    const double x = ix * bin_x, y = iy * bin_x, del = ct.delta[id];
    const double l1 = (del + 2. * x + y) / 3.0 * ct.ampl;
    const double l2 = (del - x + y) / 3.0 * ct.ampl;
    const double l3 = (del - x - 2. * y) / 3.0 * ct.ampl;
    ct.y[i] = pf_ell<FAST>(sv, l1, l2, l3);
  }
}
// the same table filled by the ELL_SNG model: one adaptive RKF45 integration of the nine-equation system per node
__global__ void __launch_bounds__(64) k_ct_table_sng(PfCtDev ct) {
  const double bin_x = PF_CT_RANGE_X / (double)(PF_CT_NBINS_XY);
  const int total = PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY;
  pf_sng_cosmo c;
  c.Omega0 = ct.sng_cosmo[0]; c.OmegaLambda = ct.sng_cosmo[1]; c.OmegaRad = ct.sng_cosmo[2]; c.OmegaK = ct.sng_cosmo[3];
  c.FR0 = ct.sng_cosmo[4]; c.H_over_c = ct.sng_cosmo[5]; c.size = ct.sng_cosmo[6];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int id = i % PF_CT_NBINS_D;
    const int ix = (i / PF_CT_NBINS_D) % PF_CT_NBINS_XY;
    const int iy = i / PF_CT_NBINS_D / PF_CT_NBINS_XY;
    const double x = ix * bin_x, y = iy * bin_x, del = ct.delta[id];
    const double l1 = (del + 2. * x + y) / 3.0 * ct.ampl;
    const double l2 = (del - x + y) / 3.0 * ct.ampl;
    const double l3 = (del - x - 2. * y) / 3.0 * ct.ampl;
    ct.y[i] = pf_ell_sng_F(l1, l2, l3, ct.sng_Din, c);
  }
}
// gsl_spline_init of the 50 x 50 node splines (src/collapse_times.c:1037-1041): GSL's cspline_init per node -- right-hand
// side, forward and back substitution with the shared factors -- then the b, d of pf_spline_bd.  One thread per node.
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_ct_splines(PfCtDev ct) {
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= PF_CT_NBINS_XY * PF_CT_NBINS_XY) return;
  const size_t o = (size_t)node * PF_CT_NBINS_D;
  pf_ct_node_spline(ct.delta, ct.alpha, ct.gamma, ct.y + o, ct.c + o, ct.b + o, ct.d + o);
}

// one wave, fixed association order: bitwise reproducible for a given launch geometry
__global__ void k_final_sum2(const double *partials, int nblocks, double *out2) {
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { a += partials[2 * i]; b += partials[2 * i + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  if (threadIdx.x == 0) { out2[0] = a; out2[1] = b; }
}

__global__ void k_sum1(const double *partials, int nblocks, double scale, double *out) {
  double a = 0;
  for (int i = threadIdx.x; i < nblocks; i += 64) a += partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if (threadIdx.x == 0) out[0] = a * scale;
}

template <bool FAST>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_collapse_cells(const double *d, size_t count, PfSplineDev s, double *F) {
  pf_spline_view sv;
  sv.x = s.x; sv.y = s.y; sv.c = s.c; sv.b = s.b; sv.d = s.d; sv.n = s.n;
  if (FAST && s.gt) { sv.gt.rec = s.gt + PF_GT_HEADER; sv.gt.lut = s.gt_lut; sv.gt.bin0 = (unsigned)s.gt[2]; sv.gt.lo_all = s.gt[3]; sv.gt.hi_all = s.gt[4]; }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double t[6], lam[3];
    for (int k = 0; k < 6; k++) t[k] = d[6 * i + k];
    F[i] = pf_inverse_collapse_time<FAST>(t, sv, lam);
  }
}

// test tap: the elementary functions of the fast flavour evaluated on the device (pf_debug_math)
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_debug_math(int which, const double *a, const double *b, size_t count, double *out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double r = 0.0;
    switch (which) {
      case 0: r = pf_div_fast(a[i], b[i]); break;
      case 1: r = pf_sqrt_fast(a[i]); break;
      case 2: { double c1, c2, c3; pf_cos3_of_acos(a[i], c1, c2, c3); r = b[i] == 0.0 ? c1 : (b[i] == 1.0 ? c2 : c3); } break;
      case 3: r = pf_log10_pos(a[i]); break;
      case 5: r = pf_pow_third<true>(a[i]); break;
      case 6: r = pf_div_const<9>(a[i]); break;
      case 7: r = pf_exp_series(a[i]); break;
      case 8: r = pf_exp10_series(a[i]); break;
      case 9: r = __builtin_amdgcn_rcp(a[i]); break;
      case 10: r = __builtin_amdgcn_rsq(a[i]); break;
      default: break;
    }
    out[i] = r;
  }
}
int pf_launch_debug_math(int which, const double *a, const double *b, size_t count, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_math, dim3(256), dim3(PF_CELL_BLOCK), 0, st, which, a, b, count, out);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// K7 (src/LPT.c:64-93)
template <typename F>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_lpt_sources(const PfLptSrcParams p) {
  __shared__ double red[2 * (PF_CELL_BLOCK / 64)];
  const F *__restrict__ h0 = (const F *)p.h[0], *__restrict__ h1 = (const F *)p.h[1],
          *__restrict__ h2 = (const F *)p.h[2], *__restrict__ h3 = (const F *)p.h[3],
          *__restrict__ h4 = (const F *)p.h[4], *__restrict__ h5 = (const F *)p.h[5];
  F *__restrict__ s2 = (F *)p.s2, *__restrict__ s3a = (F *)p.s3a, *__restrict__ s3b = (F *)p.s3b;
  const long long ncell = p.nrows * p.n;
  double acc = 0.0, dummy = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / p.n;
    const long long a = row * p.pitch + (i - row * p.n);
    const double d[6] = {(double)h0[a], (double)h1[a], (double)h2[a], (double)h3[a], (double)h4[a], (double)h5[a]};
    double src2, src31, src32;
    pf_lpt_sources_cell(d, src2, src31, src32);
    s2[a] = (F)src2;
    s3a[a] = (F)src31;
    s3b[a] = (F)src32;
    acc += (double)(F)src2;
  }
  pf_block_sum2(acc, dummy, red);
  if (threadIdx.x == 0) p.partials[blockIdx.x] = acc;
}

// K9 (src/LPT.c:134-137), all six components in one sweep
template <typename F>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_lpt_accum(const PfLptAccParams p) {
  F *__restrict__ s3b = (F *)p.s3b;
  const long long ncell = p.nrows * p.n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / p.n;
    const long long a = row * p.pitch + (i - row * p.n);
    double ph[6], hh[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { ph[c] = (double)((const F *)p.phi2[c])[a]; hh[c] = (double)((const F *)p.h[c])[a]; }
    s3b[a] = (F)pf_lpt3b_accumulate((double)s3b[a], ph, hh);
  }
}

template <typename PR>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_fill_products(PR *fmax, int *rmax, PR *vel12, size_t ncell) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) {
    fmax[i] = (PR)-10.0f;
    rmax[i] = -1;
#pragma unroll
    for (int k = 0; k < 12; k++) vel12[(size_t)k * ncell + i] = (PR)0;
  }
}

// K2's growth multiplier when it depends on |k| (SCALE_DEPENDENT build: src/fmax-pfft.c:339-364 calling
// GrowingMode*(z, k_module), i.e. InterpolateGrowth src/cosmo.c:1728-1755 and +-pow(10., .) :1789-1819):
// out(k) = in(k) * sign * 10^{lerp(T; log10 |k|)}, |k| in rad/cell as in the reference (quirk Q3).  KY layout.
template <typename F>
__global__ void __launch_bounds__(PF_CELL_BLOCK)
    k_apply_growth(const pfc<F> *__restrict__ in, pfc<F> *__restrict__ out, int n, int nyl, int nzh, int nzp, int y0,
                   const double *__restrict__ T, int nk, double logkmin, double dlogk, double sign) {
  const double knorm = 2. * PF_PI / (double)n;
  const double kmin = pow(10., logkmin), kmax = pow(10., logkmin + (nk - 1) * dlogk);
  const size_t total = (size_t)n * nyl * nzh;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int iz = (int)(i % nzh);
    const size_t r = i / nzh;
    const int yl = (int)(r % nyl), ix = (int)(r / nyl);
    int sx = ix, sy = yl + y0;
    if (sx > n / 2) sx -= n;
    if (sy > n / 2) sy -= n;
    const double kx = knorm * sx, ky = knorm * sy, kz = knorm * iz;
    const double k2_1 = kx * kx + ky * ky;
    const double k = sqrt(k2_1 + kz * kz);
    double v;
    if (k < kmin) v = T[0];
    else if (k > kmax) v = T[nk - 1];
    else {
      double dk = (log10(k) - logkmin) / dlogk;
      const int kk = (int)dk;
      dk -= kk;
      v = (kk >= nk - 1) ? T[nk - 1] : dk * T[kk + 1] + (1 - dk) * T[kk];
    }
    const double g = sign * pow(10., v);
    const size_t a = r * (size_t)nzp + iz;
    out[a] = pf_mk<F>((F)((double)in[a].x * g), (F)((double)in[a].y * g));
  }
}

// ---- general grid sizes (library-FFT path, pf_api.hip "general path") --------------------------------------------
// compute_derivative's k-space loop (src/fmax-pfft.c:303-386) on the natural layout [n][n][n/2+1]:
// out(k) = in(k) * G(k) * exp(-k^2 rs^2 / 2) * growth * norm with G = k_a k_b / k^2, k_a / k^2 (then the re/im swap),
// -1/k^2 for (0,0), 1 for (-1,-1); the k = 0 mode only takes norm (and the swap).  norm = 1/N^3 is applied here
// because the library c2r that follows is unnormalised (the reference scales after its c2r, :220-225).
__global__ void __launch_bounds__(PF_CELL_BLOCK)
    k_gen_filter(const pfc<double> *__restrict__ in, pfc<double> *__restrict__ out, int n, int a, int b, double rs, double growth,
                 const double *__restrict__ T, int nk, double logkmin, double dlogk, double sign, double norm) {
  const int nzh = n / 2 + 1;
  const double knorm = 2. * PF_PI / (double)n;
  const size_t total = (size_t)n * n * nzh;
  const bool swap = (a == 0 && b > 0) || (a > 0 && b == 0);
  double kmin = 0, kmax = 0;
  if (nk) { kmin = pow(10., logkmin); kmax = pow(10., logkmin + (nk - 1) * dlogk); }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int iz = (int)(i % nzh);
    const size_t r = i / nzh;
    int iy = (int)(r % n), ix = (int)(r / n);
    if (ix > n / 2) ix -= n;
    if (iy > n / 2) iy -= n;
    const double kx = knorm * ix, ky = knorm * iy, kz = knorm * iz;
    const double k2_0 = kx * kx, k2_1 = k2_0 + ky * ky;
    const double k_squared = k2_1 + kz * kz;
    double re = in[i].x, im = in[i].y;
    if (k_squared != 0.) {
      double g = growth;
      if (nk) {  // InterpolateGrowth, |k| in rad/cell (k_apply_growth)
        const double k = sqrt(k_squared);
        double v;
        if (k < kmin) v = T[0];
        else if (k > kmax) v = T[nk - 1];
        else {
          double dk = (log10(k) - logkmin) / dlogk;
          const int kk = (int)dk;
          dk -= kk;
          v = (kk >= nk - 1) ? T[nk - 1] : dk * T[kk + 1] + (1 - dk) * T[kk];
        }
        g = sign * pow(10., v);
      }
      const double comp[4] = {1.0, kx, ky, kz};
      double green = 1.0;
      if (a >= 0 && b >= 0) green = (a == 0 && b == 0) ? -comp[a] * comp[b] / k_squared : comp[a] * comp[b] / k_squared;
      const double w = green * exp(-0.5 * k_squared * rs * rs) * g;
      re *= w; im *= w;
    }
    if (swap) { const double t = im; im = re; re = -t; }
    out[i] = pf_mk<double>(re * norm, im * norm);
  }
}
// real field -> one fp32 column of the products (write_from_rvector_to_products, src/fmax-pfft.c:563-631)
template <typename PR>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_real_to_col(const double *__restrict__ src, PR *__restrict__ dst, size_t ncell) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) dst[i] = (PR)src[i];
}
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_scale_real(double *__restrict__ f, size_t ncell, double s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) f[i] *= s;
}

// device SoA -> the caller's AoS product_data (src/pinocchio.h:233-259)
template <typename PR>
__global__ void __launch_bounds__(PF_CELL_BLOCK)
    k_pack_products(const PR *fmax, const int *rmax, const PR *vel12, size_t ncell_total, size_t first, size_t count,
                    char *aos, size_t stride, int off_rmax, int off_fmax, int ov0, int ov1, int ov2, int ov3) {
  const int ov[4] = {ov0, ov1, ov2, ov3};
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) {
    const size_t i = first + j;
    char *rec = aos + j * stride;
    if (off_rmax >= 0) *reinterpret_cast<int *>(rec + off_rmax) = rmax[i];
    if (off_fmax >= 0) *reinterpret_cast<PR *>(rec + off_fmax) = fmax[i];
#pragma unroll
    for (int o = 0; o < 4; o++)
      if (ov[o] >= 0) {
        PR *v = reinterpret_cast<PR *>(rec + ov[o]);
        v[0] = vel12[(size_t)(3 * o + 0) * ncell_total + i];
        v[1] = vel12[(size_t)(3 * o + 1) * ncell_total + i];
        v[2] = vel12[(size_t)(3 * o + 2) * ncell_total + i];
      }
  }
}

// K11 (src/fmax.c:517-525)
template <typename PR>
__global__ void __launch_bounds__(PF_CELL_BLOCK) k_fmax_pdf(const PR *fmax, size_t ncell, unsigned long long *hist) {
  __shared__ unsigned int sh[210];
  for (int i = threadIdx.x; i < 210; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  // a block handles < 2^32 cells by construction (grid >= ncell / 2^31)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) {
    int xF = (int)((double)fmax[i] * 10.);
    if (xF < 0) xF = 0;
    if (xF >= 210) xF = 209;
    atomicAdd(&sh[xF], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 210; i += blockDim.x)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

template <typename F>
__global__ void k_spec_import(const double *src, F *dst, long long nrows, int nzh, int nzp) {
  const long long total = nrows * nzh * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (2 * nzh);
    const long long r = i - row * 2 * nzh;
    dst[row * 2 * nzp + r] = (F)src[i];
  }
}
template <typename F>
__global__ void k_spec_export(const F *src, double *dst, long long nrows, int nzh, int nzp) {
  const long long total = nrows * nzh * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (2 * nzh);
    const long long r = i - row * 2 * nzh;
    dst[i] = (double)src[row * 2 * nzp + r];
  }
}
// the same for a caller whose k-space is PFFT_TRANSPOSED_OUT on slabs (params.use_transposed_fft, src/fmax-pfft.c:92,
// 271-281): host order [ky_local][kx][kz] <-> device KY layout [kx][ky_local][kz] -- the two leading indices swap places
template <typename F>
__global__ void k_spec_import_t(const double *src, F *dst, int n, int nyl, int nzh, int nzp) {
  const long long total = (long long)nyl * n * nzh * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (2 * nzh);
    const long long r = i - row * 2 * nzh;
    const long long yl = row / n, x = row - yl * n;
    dst[(x * nyl + yl) * 2 * nzp + r] = (F)src[i];
  }
}
template <typename F>
__global__ void k_spec_export_t(const F *src, double *dst, int n, int nyl, int nzh, int nzp) {
  const long long total = (long long)nyl * n * nzh * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (2 * nzh);
    const long long r = i - row * 2 * nzh;
    const long long yl = row / n, x = row - yl * n;
    dst[i] = (double)src[(x * nyl + yl) * 2 * nzp + r];
  }
}
template <typename F>
__global__ void k_real_import(const double *src, F *dst, long long nrows, int n, long long pitch) {
  const long long total = nrows * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / n;
    dst[row * pitch + (i - row * n)] = (F)src[i];
  }
}
template <typename F>
__global__ void k_real_export(const F *src, double *dst, long long nrows, int n, long long pitch) {
  const long long total = nrows * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / n;
    dst[i] = (double)src[row * pitch + (i - row * n)];
  }
}
// x-slab rows [xl][y][nzp] -> the P all-to-all blocks [q][xl][yl][nzp], y = q*nyl + yl (back: the inverse gather)
template <typename F>
__global__ void k_to_blocks(const F *src, F *dst, int nxl, int n, int nyl, int nzp, int back) {
  const long long total = (long long)nxl * n * nzp * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i % (2 * nzp);
    const long long row = i / (2 * nzp);
    const int y = (int)(row % n), xl = (int)(row / n);
    const int q = y / nyl, yl = y - q * nyl;
    const long long b = (((long long)q * nxl + xl) * nyl + yl) * 2 * nzp + r;
    if (back) dst[i] = src[b]; else dst[b] = src[i];
  }
}

template <typename F>
__global__ void k_extract_dc(const F *spec, double scale, double *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)spec[0] * scale;
}

// ------------------------------------------------------------ launchers ----
static inline int pf_grid_for(size_t n, int cap = 256 * 8) {
  size_t b = (n + PF_CELL_BLOCK - 1) / PF_CELL_BLOCK;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}
#define PF_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : 1)

int pf_launch_ct_build(const PfSplineDev &sp, const PfCtDev &ct, int fast, int compute_table, hipStream_t st) {
  if (compute_table && ct.model == 1) {
    hipLaunchKernelGGL(k_ct_table_sng, dim3((PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY + 63) / 64), dim3(64), 0, st, ct);
  } else if (compute_table) {
    const int g = pf_grid_for((size_t)PF_CT_NBINS_D * PF_CT_NBINS_XY * PF_CT_NBINS_XY);
    if (fast) hipLaunchKernelGGL(k_ct_table<true>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, sp, ct);
    else hipLaunchKernelGGL(k_ct_table<false>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, sp, ct);
  }
  hipLaunchKernelGGL(k_ct_splines, dim3((PF_CT_NBINS_XY * PF_CT_NBINS_XY + PF_CELL_BLOCK - 1) / PF_CELL_BLOCK), dim3(PF_CELL_BLOCK), 0, st, ct);
  return PF_CHECK_LAUNCH();
}
// one kernel per (field type, libm flavour, variant, PRODFLOAT); double products exist with fp64 fields only
template <typename PR> static int pf_launch_collapse_as(int fb, const PfCollapseParams &p, hipStream_t st) {
  constexpr bool F32_FIELDS_TOO = std::is_same<PR, float>::value;
  if (p.spline.n > PF_MAX_KNOTS) return 2;
  if (fb != 8 && !F32_FIELDS_TOO) return 2;
  const dim3 g(p.nblocks), b(PF_CELL_BLOCK);
#define PF_BY_FIELD_AND_LIBM(K)                                                            \
  do {                                                                                     \
    if (fb == 8) {                                                                         \
      if (p.fast) hipLaunchKernelGGL((K<double, true, PR>), g, b, 0, st, p);               \
      else hipLaunchKernelGGL((K<double, false, PR>), g, b, 0, st, p);                     \
    } else if constexpr (F32_FIELDS_TOO) {                                                 \
      if (p.fast) hipLaunchKernelGGL((K<float, true, PR>), g, b, 0, st, p);                \
      else hipLaunchKernelGGL((K<float, false, PR>), g, b, 0, st, p);                      \
    }                                                                                      \
  } while (0)
  if (p.tabulated) {
    if (fb == 8) {
      if (p.fast) pf_launch_collapse_tab<double, true, PR>(p, st);
      else pf_launch_collapse_tab<double, false, PR>(p, st);
    } else if constexpr (F32_FIELDS_TOO) {
      if (p.fast) pf_launch_collapse_tab<float, true, PR>(p, st);
      else pf_launch_collapse_tab<float, false, PR>(p, st);
    }
    return PF_CHECK_LAUNCH();
  }
  if (p.sources) {
    if (p.invariants || p.tabulated || p.sng) return 2;
    PF_BY_FIELD_AND_LIBM(k_collapse_src);
    return PF_CHECK_LAUNCH();
  }
  if (p.sng) {
    if (p.invariants) return 2;
    PF_BY_FIELD_AND_LIBM(k_collapse_sng);
    return PF_CHECK_LAUNCH();
  }
  if (p.invariants) {
    if (fb != 8 || p.tabulated) return 2;
    // (workgroups of several groups of PF_CELL_BLOCK threads: as many threads in the grid, one pair of partial sums per group; a
    //  block count that does not divide -- tiny grids -- runs workgroups of one group)
    static_assert(PF_SOLVE_INV_BLOCK % PF_CELL_BLOCK == 0 && PF_SQ_BLOCK % PF_CELL_BLOCK == 0, "whole groups");
    const int gi = p.nblocks % (PF_SOLVE_INV_BLOCK / PF_CELL_BLOCK) == 0 ? PF_SOLVE_INV_BLOCK / PF_CELL_BLOCK : 1;
    const int gq = PF_SQ_BLOCK / PF_CELL_BLOCK;
    if (p.fast && PF_SOLVE_QUEUE && p.nrows * p.n < (1ll << 32) && p.nblocks % gq == 0)  // (the queue holds 32-bit cell indices)
      hipLaunchKernelGGL((k_collapse_invq<PR>), dim3(p.nblocks / gq), dim3(PF_SQ_BLOCK), 0, st, p);
    else if (p.fast) hipLaunchKernelGGL((k_collapse_inv<true, PR>), dim3(p.nblocks / gi), dim3(PF_CELL_BLOCK * gi), 0, st, p);
    else hipLaunchKernelGGL((k_collapse_inv<false, PR>), dim3(p.nblocks / gi), dim3(PF_CELL_BLOCK * gi), 0, st, p);
    return PF_CHECK_LAUNCH();
  }
  PF_BY_FIELD_AND_LIBM(k_collapse);
#undef PF_BY_FIELD_AND_LIBM
  return PF_CHECK_LAUNCH();
}
int pf_launch_collapse(int fb, const PfCollapseParams &p, hipStream_t st) {
  return p.prod_f64 ? pf_launch_collapse_as<double>(fb, p, st) : pf_launch_collapse_as<float>(fb, p, st);
}
int pf_launch_final_sum(const double *partials, int nblocks, double *out2, hipStream_t st) {
  hipLaunchKernelGGL(k_final_sum2, dim3(1), dim3(64), 0, st, partials, nblocks, out2);
  return PF_CHECK_LAUNCH();
}
int pf_launch_sum1(const double *partials, int nblocks, double scale, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(64), 0, st, partials, nblocks, scale, out);
  return PF_CHECK_LAUNCH();
}
int pf_launch_collapse_cells(const double *d, size_t count, PfSplineDev s, double *F, int fast, hipStream_t st) {
  if (fast) hipLaunchKernelGGL(k_collapse_cells<true>, dim3(pf_grid_for(count)), dim3(PF_CELL_BLOCK), 0, st, d, count, s, F);
  else hipLaunchKernelGGL(k_collapse_cells<false>, dim3(pf_grid_for(count)), dim3(PF_CELL_BLOCK), 0, st, d, count, s, F);
  return PF_CHECK_LAUNCH();
}
int pf_launch_lpt_sources(int fb, const PfLptSrcParams &p, hipStream_t st) {
  if (fb == 8) hipLaunchKernelGGL(k_lpt_sources<double>, dim3(p.nblocks), dim3(PF_CELL_BLOCK), 0, st, p);
  else hipLaunchKernelGGL(k_lpt_sources<float>, dim3(p.nblocks), dim3(PF_CELL_BLOCK), 0, st, p);
  return PF_CHECK_LAUNCH();
}
int pf_launch_lpt_accum(int fb, const PfLptAccParams &p, hipStream_t st) {
  const int g = pf_grid_for((size_t)(p.nrows * p.n));
  if (fb == 8) hipLaunchKernelGGL(k_lpt_accum<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, p);
  else hipLaunchKernelGGL(k_lpt_accum<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, p);
  return PF_CHECK_LAUNCH();
}
int pf_launch_fill_products(void *fmax, int *rmax, void *vel12, size_t ncell, int pb, hipStream_t st) {
  const dim3 g(pf_grid_for(ncell)), b(PF_CELL_BLOCK);
  if (pb == 8) hipLaunchKernelGGL(k_fill_products<double>, g, b, 0, st, (double *)fmax, rmax, (double *)vel12, ncell);
  else hipLaunchKernelGGL(k_fill_products<float>, g, b, 0, st, (float *)fmax, rmax, (float *)vel12, ncell);
  return PF_CHECK_LAUNCH();
}
int pf_launch_pack_products(int pb, const void *fmax, const int *rmax, const void *vel12, size_t ncell_total, size_t first,
                            size_t count, char *aos, size_t stride, int off_rmax, int off_fmax, const int ov[4],
                            hipStream_t st) {
  const dim3 g(pf_grid_for(count)), b(PF_CELL_BLOCK);
  if (pb == 8)
    hipLaunchKernelGGL(k_pack_products<double>, g, b, 0, st, (const double *)fmax, rmax, (const double *)vel12, ncell_total, first, count, aos,
                       stride, off_rmax, off_fmax, ov[0], ov[1], ov[2], ov[3]);
  else
    hipLaunchKernelGGL(k_pack_products<float>, g, b, 0, st, (const float *)fmax, rmax, (const float *)vel12, ncell_total, first, count, aos,
                       stride, off_rmax, off_fmax, ov[0], ov[1], ov[2], ov[3]);
  return PF_CHECK_LAUNCH();
}
int pf_launch_apply_growth(int fb, const void *in, void *out, int n, int nyl, int nzh, int nzp, int y0, const double *T, int nk,
                           double logkmin, double dlogk, double sign, hipStream_t st) {
  const int g = pf_grid_for((size_t)n * nyl * nzh);
  if (fb == 8) hipLaunchKernelGGL(k_apply_growth<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const pfc<double> *)in, (pfc<double> *)out, n, nyl, nzh, nzp, y0, T, nk, logkmin, dlogk, sign);
  else hipLaunchKernelGGL(k_apply_growth<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const pfc<float> *)in, (pfc<float> *)out, n, nyl, nzh, nzp, y0, T, nk, logkmin, dlogk, sign);
  return PF_CHECK_LAUNCH();
}
int pf_launch_gen_filter(const void *in, void *out, int n, int a, int b, double rs, double growth, const double *T, int nk, double logkmin,
                         double dlogk, double sign, double norm, hipStream_t st) {
  hipLaunchKernelGGL(k_gen_filter, dim3(pf_grid_for((size_t)n * n * (n / 2 + 1))), dim3(PF_CELL_BLOCK), 0, st, (const pfc<double> *)in,
                     (pfc<double> *)out, n, a, b, rs, growth, T, nk, logkmin, dlogk, sign, norm);
  return PF_CHECK_LAUNCH();
}
int pf_launch_real_to_col(const double *src, void *dst, size_t ncell, int pb, hipStream_t st) {
  if (pb == 8) hipLaunchKernelGGL(k_real_to_col<double>, dim3(pf_grid_for(ncell)), dim3(PF_CELL_BLOCK), 0, st, src, (double *)dst, ncell);
  else hipLaunchKernelGGL(k_real_to_col<float>, dim3(pf_grid_for(ncell)), dim3(PF_CELL_BLOCK), 0, st, src, (float *)dst, ncell);
  return PF_CHECK_LAUNCH();
}
int pf_launch_scale_real(double *f, size_t ncell, double s, hipStream_t st) {
  hipLaunchKernelGGL(k_scale_real, dim3(pf_grid_for(ncell)), dim3(PF_CELL_BLOCK), 0, st, f, ncell, s);
  return PF_CHECK_LAUNCH();
}
int pf_launch_fmax_pdf(const void *fmax, size_t ncell, unsigned long long *hist, int pb, hipStream_t st) {
  if (pb == 8) hipLaunchKernelGGL(k_fmax_pdf<double>, dim3(pf_grid_for(ncell)), dim3(PF_CELL_BLOCK), 0, st, (const double *)fmax, ncell, hist);
  else hipLaunchKernelGGL(k_fmax_pdf<float>, dim3(pf_grid_for(ncell)), dim3(PF_CELL_BLOCK), 0, st, (const float *)fmax, ncell, hist);
  return PF_CHECK_LAUNCH();
}
int pf_launch_spec_import(int fb, const double *src, void *dst, long long nrows, int nzh, int nzp, hipStream_t st) {
  const int g = pf_grid_for((size_t)(nrows * nzh * 2));
  if (fb == 8) hipLaunchKernelGGL(k_spec_import<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (double *)dst, nrows, nzh, nzp);
  else hipLaunchKernelGGL(k_spec_import<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (float *)dst, nrows, nzh, nzp);
  return PF_CHECK_LAUNCH();
}
int pf_launch_spec_export(int fb, const void *src, double *dst, long long nrows, int nzh, int nzp, hipStream_t st) {
  const int g = pf_grid_for((size_t)(nrows * nzh * 2));
  if (fb == 8) hipLaunchKernelGGL(k_spec_export<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const double *)src, dst, nrows, nzh, nzp);
  else hipLaunchKernelGGL(k_spec_export<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const float *)src, dst, nrows, nzh, nzp);
  return PF_CHECK_LAUNCH();
}
int pf_launch_spec_import_t(int fb, const double *src, void *dst, int n, int nyl, int nzh, int nzp, hipStream_t st) {
  const int g = pf_grid_for((size_t)nyl * n * nzh * 2);
  if (fb == 8) hipLaunchKernelGGL(k_spec_import_t<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (double *)dst, n, nyl, nzh, nzp);
  else hipLaunchKernelGGL(k_spec_import_t<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (float *)dst, n, nyl, nzh, nzp);
  return PF_CHECK_LAUNCH();
}
int pf_launch_spec_export_t(int fb, const void *src, double *dst, int n, int nyl, int nzh, int nzp, hipStream_t st) {
  const int g = pf_grid_for((size_t)nyl * n * nzh * 2);
  if (fb == 8) hipLaunchKernelGGL(k_spec_export_t<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const double *)src, dst, n, nyl, nzh, nzp);
  else hipLaunchKernelGGL(k_spec_export_t<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const float *)src, dst, n, nyl, nzh, nzp);
  return PF_CHECK_LAUNCH();
}
int pf_launch_real_import(int fb, const double *src, void *dst, long long nrows, int n, long long pitch, hipStream_t st) {
  const int g = pf_grid_for((size_t)(nrows * n));
  if (fb == 8) hipLaunchKernelGGL(k_real_import<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (double *)dst, nrows, n, pitch);
  else hipLaunchKernelGGL(k_real_import<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, src, (float *)dst, nrows, n, pitch);
  return PF_CHECK_LAUNCH();
}
int pf_launch_real_export(int fb, const void *src, double *dst, long long nrows, int n, long long pitch, hipStream_t st) {
  const int g = pf_grid_for((size_t)(nrows * n));
  if (fb == 8) hipLaunchKernelGGL(k_real_export<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const double *)src, dst, nrows, n, pitch);
  else hipLaunchKernelGGL(k_real_export<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const float *)src, dst, nrows, n, pitch);
  return PF_CHECK_LAUNCH();
}
int pf_launch_to_blocks(int fb, const void *src, void *dst, int nxl, int n, int nyl, int nzp, int back, hipStream_t st) {
  const int g = pf_grid_for((size_t)nxl * n * nzp * 2);
  if (fb == 8) hipLaunchKernelGGL(k_to_blocks<double>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const double *)src, (double *)dst, nxl, n, nyl, nzp, back);
  else hipLaunchKernelGGL(k_to_blocks<float>, dim3(g), dim3(PF_CELL_BLOCK), 0, st, (const float *)src, (float *)dst, nxl, n, nyl, nzp, back);
  return PF_CHECK_LAUNCH();
}
int pf_launch_extract_dc(int fb, const void *spec, double scale, double *out, hipStream_t st) {
  if (fb == 8) hipLaunchKernelGGL(k_extract_dc<double>, dim3(1), dim3(64), 0, st, (const double *)spec, scale, out);
  else hipLaunchKernelGGL(k_extract_dc<float>, dim3(1), dim3(64), 0, st, (const float *)spec, scale, out);
  return PF_CHECK_LAUNCH();
}
