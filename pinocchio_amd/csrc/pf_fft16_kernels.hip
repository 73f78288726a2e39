// pf_fft16_kernels.hip -- the strided passes (x, y) of 2048-point fp32 lines: BASELINE config 5 (2048^3, fp32 fields, eight ranks),
// whose passes replace the transposed 1-D transforms inside pfft_execute (src/fmax-pfft.c:203-228) and the k-space filter of
// compute_derivative (src/fmax-pfft.c:306-397) under the per-rank limit of src/pinocchio.h:297.
//
//   k_strided16 : a workgroup owns eight 16-byte elements (sixteen fp32 columns: 128 contiguous bytes) of every row of a line of
//                 tiles and all 2048 points along the transformed axis -- 256 KB, held in the registers of its 1024 threads
//                 (sixteen points x two columns each).  Plan 16 x 16 x 8, arithmetic on (re, im) register pairs in packed fp32
//                 instructions, LDS for the two exchanges only (pf_fft16.h).  A tile that serves several jobs is READ AGAIN for
//                 each of them -- nothing can hold a second copy of it on the CU -- which costs nothing that shows: the second and
//                 third read come from the L2 / MALL (profiles/r05_notes.md, seg_probe: 1 -> 3 jobs at 5.2 TB/s of algorithmic
//                 bytes with the tile re-read, 3.4 TB/s for the 64-byte segments of the eight-point kernel that keeps it).
// Same parameter block, filter, k multipliers, band limits and layouts as k_strided (pf_fft_kernels.hip), which stays the kernel of
// the cases this one does not take (addresses that do not split into a scalar and a 32-bit lane part, slab stores of a replicated
// spectrum).
#include <atomic>
#include <cstdio>

#include "pf_internal.h"
#include "pf_fft16.h"
#include "pf_fft_stages.h"  // pf_wave_sync
#include "pf_strided_addr.h"

typedef float pf_f4 __attribute__((ext_vector_type(4)));

#ifndef PF_STRIDED16
#define PF_STRIDED16 1  // (0 in an A/B build: 2048-point fp32 lines on k_strided<pf_f32x2, 2048, 4>, 64-byte row segments)
#endif

// one 16-byte element (a column pair of a row): scalar 64-bit part + 32-bit part per lane (pf_strided_addr.h)
__device__ __forceinline__ pf_f4 pf16_ld(const pfc<float> *base, long long u, unsigned lane) {
  return *reinterpret_cast<const pf_f4 *>(reinterpret_cast<const char *>(base + u) + (size_t)(lane * 8u));
}
__device__ __forceinline__ void pf16_st4(pfc<float> *base, long long u, unsigned lane, pf_f2 a, pf_f2 b) {
  const pf_f4 t = (pf_f4){a.x, a.y, b.x, b.y};
  pf_f4 *q = reinterpret_cast<pf_f4 *>(reinterpret_cast<char *>(base + u) + (size_t)(lane * 8u));
#if PF_NT
  __builtin_nontemporal_store(t, q);
#else
  *q = t;
#endif
}
__device__ __forceinline__ void pf16_st2(pfc<float> *base, long long u, unsigned lane, pf_f2 a) {
  pf_f2 *q = reinterpret_cast<pf_f2 *>(reinterpret_cast<char *>(base + u) + (size_t)(lane * 8u));
#if PF_NT
  __builtin_nontemporal_store(a, q);
#else
  *q = a;
#endif
}

// PRE: the launch carries the filter of the first pass (p.pre); BAND: it is band-limited (p.band_e or p.band_outer < N / 2)
#ifndef PF_S16_SPLIT
#define PF_S16_SPLIT 0  // (1 in an A/B build: ONE job per workgroup, see below.  Measured slower: x-pass 1 -> 3 52.5 against 49.0 ms per step,
                        //  y-pass 3 -> 6 87 against 84 on BASELINE config 5's slab: profiles/r05_notes.md)
#endif
template <int DIR, bool PRE, bool BAND, bool SPLIT = PF_S16_SPLIT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) k_strided16(const PfStridedParams p, const long long nwork, const int ntiles) {
  using A = PfCxPk;
  using C = pf_f2;
  constexpr int N = PF16_N, T = 8, NT = N / 16;  // 128 threads per column pair
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);  // [2048 slots][T] (re, im) pairs: ONE column of every pair at a time (128 KB)
  // PF_S16_SPLIT (A/B): ONE job per workgroup, the jobs of a tile on consecutive work items -- which the hardware deals to the CUs of
  // ONE XCD back to back (pf_xcd_swizzle) -- in the hope that the tile they all read comes from that XCD's L2 for all but the first.
  // (A workgroup running its tile's jobs in series reads it again 25 us later, when it has long left the L2 -- 32 CUs x 256 KB of tiles
  // in flight per 4 MB: counted at the L2's exit, the x-pass 1 -> 3 moves 11.4 GB in for 4.3 GB of input per launch,
  // profiles/r05_2048_pmc_traffic.json.)  It does not pay: three workgroups that miss together are not served as one miss, and a
  // workgroup's start-up is paid three times.
  const long long wid = pf_xcd_swizzle(blockIdx.x, (nwork + 7) >> 3);
  if (wid >= nwork) return;
  const int jsplit = SPLIT ? p.njobs : 1;
  const long long w = wid / jsplit;
  const int j_first = SPLIT ? (int)(wid - w * jsplit) : 0, j_end = SPLIT ? j_first + 1 : p.njobs;
  const int tid = threadIdx.x;
  const int c = tid % T, tl = tid / T;
  const int tile = (int)(w % ntiles);
  const int outer = (int)(w / ntiles);
  const int col = 2 * (tile * T + c);  // first column of this thread
  const pf_f2 *__restrict__ tw = reinterpret_cast<const pf_f2 *>(p.tw);
  const double kfd = 2.0 * 3.14159265358979323846 / (double)N;
  if (BAND && p.band_outer < N / 2) {  // whole line of tiles outside the band of the smoothed spectrum: its output is never read
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    if (so > p.band_outer || so < -p.band_outer) return;
  }
  // first pass: the window of the two untransformed axes times the growth factor, one exp per thread and column, and their part of k^2
  float wocf[2] = {1.f, 1.f}, ko2kc2[2] = {0.f, 0.f};
  if (PRE) {
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    const double ko = kfd * so;
#pragma unroll
    for (int l = 0; l < 2; l++) {
      const double kc = kfd * (col + l);
      const double s = ko * ko + kc * kc;
      wocf[l] = (float)((p.rs != 0.0 ? exp(-0.5 * s * p.rs * p.rs) : 1.0) * p.growth);
      ko2kc2[l] = (float)s;
    }
  }
#pragma unroll 1
  for (int j = j_first; j < j_end; j++) {
    // (an opaque copy of the thread coordinates per iteration keeps the index math inside the loop: see k_strided)
    int tlj = tl, cj = c;
    asm volatile("" : "+v"(tlj), "+v"(cj));
    const int wv = __builtin_amdgcn_readfirstlane(tlj >> 3);  // this wave's place in the workgroup = the t it takes after exchange 1
    // ---- the tile of this job's input: sixteen rows x two columns per thread.  No load is predicated per lane: a block of 128 rows
    // (register m of every thread) is read if any of its rows lies inside the band of the smoothed spectrum -- a uniform test -- and
    // the rows beyond the band that come with it are then cleared; column pairs beyond ncols are read from the rows' padding (whole
    // tiles fit the pitch of every layout) and never stored.  A tile that serves several jobs is read by each of them (see the
    // head of this file).
    C va[16], vb[16];
    {
      const pfc<float> *__restrict__ in = reinterpret_cast<const pfc<float> *>(p.job[j].in);
      const unsigned lane = pf_addr_lane(p.ain, tlj, 2 * cj);
#pragma unroll
      for (int m = 0; m < 16; m++) {
        // rows 128 m .. 128 m + 127: signed wavenumbers from 128 m up (m < 8), or down to 128 m + 127 - N (m >= 8)
        const bool inband = !BAND || (m < 8 ? 128 * m <= p.band_e : N - (128 * m + 127) <= p.band_e);
        pf_f4 t = (pf_f4){0.f, 0.f, 0.f, 0.f};
        if (inband) {
          long long u = pf_addr_uniform<NT>(p.ain, outer, m, 2 * tile * T);
          asm volatile("" : "+s"(u));  // (formed here, one after the other: all sixteen at once, ahead of the loads, spill the scalar registers)
          t = pf16_ld(in, u, lane);
          if (BAND) {  // the rows of the block that lie beyond the band: exact zeros, as for every kernel of the pass
            const int e = pf16_line_index(tlj, m), se = e > N / 2 ? e - N : e;
            if (se > p.band_e || se < -p.band_e) t = (pf_f4){0.f, 0.f, 0.f, 0.f};
          }
        }
        va[m] = (C){t.x, t.y}; vb[m] = (C){t.z, t.w};
      }
    }
    // twiddles of the last stage (one table value per butterfly, per lane): requested here, used after the two exchanges
    const C w20 = tw[pf16_tw2(tlj, 0)], w21 = tw[pf16_tw2(tlj, 1)];
    // ---- filter of the first pass and the multiplier along the transformed axis, in the precision of the fields
    const int mul = p.job[j].mul;
    const float kff = (float)kfd;
    if (PRE) {
      const bool windowed = p.rs != 0.0;
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const int e = pf16_line_index(tlj, m);
        const float ke = kff * (float)(e > N / 2 ? e - N : e);
        float wk = windowed ? (float)p.etab[e] : 1.f;  // the window along the transformed axis ...
        if (mul == PF_MUL_K || mul == PF_MUL_IK) wk *= ke;  // ... times this job's power of k
        else if (mul == PF_MUL_K2) wk *= ke * ke;
        // 1 / k^2 by the hardware reciprocal (1 ulp of fp32: the precision of the field).  k = 0 -- one element of the whole grid,
        // left untouched by the reference (src/fmax-pfft.c:368) and zero here -- is dealt with below, not by a test per element
        const float fa = wk * wocf[0] * __builtin_amdgcn_rcpf(fmaf(ke, ke, ko2kc2[0]));
        const float fb = wk * wocf[1] * __builtin_amdgcn_rcpf(fmaf(ke, ke, ko2kc2[1]));
        va[m] = va[m] * fa;
        vb[m] = vb[m] * fb;
      }
      if (ko2kc2[0] == 0.f && tlj == 0) va[0] = (C){0.f, 0.f};  // k = 0: row 0 of the thread whose first column and outer wavenumber are zero
    } else if (mul != PF_MUL_ONE) {
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const int e = pf16_line_index(tlj, m);
        const float ke = kff * (float)(e > N / 2 ? e - N : e);
        const float km = mul == PF_MUL_K2 ? ke * ke : ke;
        va[m] = va[m] * km;
        vb[m] = vb[m] * km;
      }
    }
    if (mul == PF_MUL_IK) {
#pragma unroll
      for (int m = 0; m < 16; m++) { va[m] = A::muli<+1>(va[m]); vb[m] = A::muli<+1>(vb[m]); }
    }
    // ---- stage 0
    pfx_bfly16<A, DIR>(va);
    pfx_bfly16<A, DIR>(vb);
    // ---- exchange 1, one column of the pairs at a time (pf_fft16.h): every thread writes its sixteen outputs of the column, a
    // barrier, and reads the sixteen inputs of its stage-1 butterfly back into the same registers
    __syncthreads();  // the exchange 2 of the job before is over in every wave
#pragma unroll
    for (int t = 0; t < 16; t++) lds[pf16_x1_write(tlj, t) * T + cj] = va[t];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) va[r] = lds[pf16_x1_read(tlj, r) * T + cj];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) lds[pf16_x1_write(tlj, t) * T + cj] = vb[t];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) vb[r] = lds[pf16_x1_read(tlj, r) * T + cj];
    // ---- stage 1: twiddle W_256^(wv r), the same for the whole wave: scalar loads, scalar-register operands
#pragma unroll
    for (int r = 1; r < 16; r++) {
      const C w1 = tw[pf16_tw1(wv) * r];
      va[r] = A::cmul_s<DIR>(va[r], w1);
      vb[r] = A::cmul_s<DIR>(vb[r], w1);
    }
    pfx_bfly16<A, DIR>(va);
    pfx_bfly16<A, DIR>(vb);
    // ---- exchange 2, inside the wave: its own 128 slots (which only it has read since the last barrier), outputs s = s8 + 8 b in
    // two halves b
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
      for (int s8 = 0; s8 < 8; s8++) { lds[pf16_x2_write(tlj, s8, 0) * T + cj] = va[8 * b + s8]; lds[pf16_x2_write(tlj, s8, 1) * T + cj] = vb[8 * b + s8]; }
      pf_wave_sync();
#pragma unroll
      for (int r = 0; r < 8; r++) { va[8 * b + r] = lds[pf16_x2_read(tlj, r, 0) * T + cj]; vb[8 * b + r] = lds[pf16_x2_read(tlj, r, 1) * T + cj]; }
      pf_wave_sync();
    }
    // ---- stage 2: two radix-8 butterflies per column, twiddle W_2048^(j r) with j = wv + 16 s0 + 128 b
#pragma unroll
    for (int b = 0; b < 2; b++) {
      C wp[7];
      pfx_powers7<A>(b ? w21 : w20, wp);
      C ua[8], ub[8];
      ua[0] = va[8 * b]; ub[0] = vb[8 * b];
#pragma unroll
      for (int r = 1; r < 8; r++) { ua[r] = A::cmul<DIR>(va[8 * b + r], wp[r - 1]); ub[r] = A::cmul<DIR>(vb[8 * b + r], wp[r - 1]); }
      pfx_bfly8<A, DIR>(ua);
      pfx_bfly8<A, DIR>(ub);
#pragma unroll
      for (int r = 0; r < 8; r++) { va[8 * b + r] = ua[r]; vb[8 * b + r] = ub[r]; }
    }
    // ---- store: register 8 b + s1 is element lane + 128 (b + 2 s1) of the line.  Every tile but the last of a line of tiles is whole
    // (a uniform test): sixteen unpredicated 16-byte stores per thread
    {
      pfc<float> *__restrict__ out = reinterpret_cast<pfc<float> *>(p.job[j].out);
      const unsigned lane = pf_addr_lane(p.aout, pf16_out_lane(tlj), 2 * cj);
      if (2 * (tile + 1) * T <= p.ncols) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
          long long u = pf_addr_uniform<NT>(p.aout, outer, (m >> 3) + 2 * (m & 7), 2 * tile * T);
          asm volatile("" : "+s"(u));
          pf16_st4(out, u, lane, va[m], vb[m]);
        }
      } else {
        const int colj = 2 * (tile * T + cj);
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const long long u = pf_addr_uniform<NT>(p.aout, outer, (m >> 3) + 2 * (m & 7), 2 * tile * T);
          if (colj + 1 < p.ncols) pf16_st4(out, u, lane, va[m], vb[m]);
          else if (colj < p.ncols) pf16_st2(out, u, lane, va[m]);
        }
      }
    }
  }
}

// ---- 1024-point fp32 lines: EIGHT points x two columns per thread (the tile, 128 KB, fits LDS and a second copy of it fits the
// registers: the input of several jobs is kept, as in k_strided), in the same packed (re, im) arithmetic.  Plan 16 x 8 x 8 with the
// paired first stage of pf_fft_core.h: each thread its radix 8, the odd one of a pair the twiddle W16^k, then one radix-2 step
// across the pair through the lanes of the wave (row_ror:8).  What k_strided<pf_f32x2, 1024, 8> spends beside its butterflies --
// lane shuffles on every load and store (its pairs are (re, re), (im, im)), twiddle powers and constants in separate registers
// per half, ~940 vector instructions per job -- is gone: ~420.
#ifndef PF_STRIDED_PK8
#define PF_STRIDED_PK8 1  // (0 in an A/B build: 1024-point fp32 lines on k_strided<pf_f32x2, 1024, 8>)
#endif
__device__ __forceinline__ pf_f2 pf_pk_xor8(pf_f2 v) { pf_f2 r; r.x = pf_lane_xor8(v.x); r.y = pf_lane_xor8(v.y); return r; }

template <int DIR, bool PRE, bool BAND>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) k_strided_pk8(const PfStridedParams p, const long long nwork, const int ntiles) {
  using A = PfCxPk;
  using C = pf_f2;
  constexpr int N = 1024, T = 8, NT = N / 8;  // 128 threads per column pair
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pf_f4 *lds = reinterpret_cast<pf_f4 *>(smem);  // [N][T] column pairs (128 KB)
  const long long w = pf_xcd_swizzle(blockIdx.x, (nwork + 7) >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x;
  const int c = tid % T, tl = tid / T;
  const int tile = (int)(w % ntiles);
  const int outer = (int)(w / ntiles);
  const int col = 2 * (tile * T + c);
  const pf_f2 *__restrict__ tw = reinterpret_cast<const pf_f2 *>(p.tw);
  const double kfd = 2.0 * 3.14159265358979323846 / (double)N;
  const float kff = (float)kfd;
  if (BAND && p.band_outer < N / 2) {
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    if (so > p.band_outer || so < -p.band_outer) return;
  }
  float wocf[2] = {1.f, 1.f}, ko2kc2[2] = {0.f, 0.f};
  if (PRE) {
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    const double ko = kfd * so;
#pragma unroll
    for (int l = 0; l < 2; l++) {
      const double kc = kfd * (col + l);
      const double s = ko * ko + kc * kc;
      wocf[l] = (float)((p.rs != 0.0 ? exp(-0.5 * s * p.rs * p.rs) : 1.0) * p.growth);
      ko2kc2[l] = (float)s;
    }
  }
  pf_f4 src[8];
  // a tile of input `in` into src: rows pf_line_index<N, true>(tl, m) = lane part + 128 m, a block of 128 rows at a time (uniform band
  // test per block, the rows beyond the band inside a block cleared; nothing predicated by column: see k_strided16)
  auto load_tile = [&](const void *inp, int tlj, int cj) {
    const pfc<float> *__restrict__ in = reinterpret_cast<const pfc<float> *>(inp);
    const int e0 = pf_line_index<N, true>(tlj, 0);
    const unsigned lane = pf_addr_lane(p.ain, e0, 2 * cj);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const bool inband = !BAND || (m < 4 ? 128 * m <= p.band_e : N - (128 * m + 127) <= p.band_e);
      pf_f4 t = (pf_f4){0.f, 0.f, 0.f, 0.f};
      if (inband) {
        long long u = pf_addr_uniform<NT>(p.ain, outer, m, 2 * tile * T);
        asm volatile("" : "+s"(u));
        t = pf16_ld(in, u, lane);
        if (BAND) {
          const int e = e0 + m * NT, se = e > N / 2 ? e - N : e;
          if (se > p.band_e || se < -p.band_e) t = (pf_f4){0.f, 0.f, 0.f, 0.f};
        }
      }
      src[m] = t;
    }
  };
  load_tile(p.job[0].in, tl, c);
#pragma unroll 1
  for (int j = 0; j < p.njobs; j++) {
    int tlj = tl, cj = c;
    asm volatile("" : "+v"(tlj), "+v"(cj));
    const int e0 = pf_line_index<N, true>(tlj, 0);
    // jobs are grouped by input: a tile is read once, filtered once, and transformed for every job that uses it
    if (PRE && (j == 0 || p.job[j].in != p.job[j - 1].in)) {
      const bool windowed = p.rs != 0.0;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int e = e0 + m * NT;
        const float ke = kff * (float)(e > N / 2 ? e - N : e);
        const float wk = windowed ? (float)p.etab[e] : 1.f;
        const float fa = wk * wocf[0] * __builtin_amdgcn_rcpf(fmaf(ke, ke, ko2kc2[0]));
        const float fb = wk * wocf[1] * __builtin_amdgcn_rcpf(fmaf(ke, ke, ko2kc2[1]));
        src[m] = (pf_f4){src[m].x * fa, src[m].y * fa, src[m].z * fb, src[m].w * fb};
      }
      if (ko2kc2[0] == 0.f && e0 == 0) { src[0].x = 0.f; src[0].y = 0.f; }  // k = 0 (see k_strided16)
    }
    const int mul = p.job[j].mul;
    C va[8], vb[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      va[m] = (C){src[m].x, src[m].y}; vb[m] = (C){src[m].z, src[m].w};
      if (mul != PF_MUL_ONE) {
        const int e = e0 + m * NT;
        const float ke = kff * (float)(e > N / 2 ? e - N : e);
        const float km = mul == PF_MUL_K2 ? ke * ke : ke;
        va[m] = va[m] * km; vb[m] = vb[m] * km;
        if (mul == PF_MUL_IK) { va[m] = A::muli<+1>(va[m]); vb[m] = A::muli<+1>(vb[m]); }
      }
    }
    // src is free once the last job on this input has taken its copy: the next input's tile travels during the stages
    if (j + 1 < p.njobs && p.job[j + 1].in != p.job[j].in) load_tile(p.job[j + 1].in, tlj, cj);
    // twiddles of the two later stages: w^1 of this thread's butterfly (pf_stage_twiddles: k = tl & (NS - 1), table step N / (NS R))
    const C w1 = tw[(tlj & 15) * 8], w2 = tw[tlj & 127];
    // ---- stage 0: the pair (tl, tl ^ 1) = (h = 0, 1) of butterfly tl >> 1
    pfx_bfly8<A, DIR>(va);
    pfx_bfly8<A, DIR>(vb);
    {
      const bool odd = tlj & 1;
      if (odd) {
        const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#define PF_W16(k, cc, ss) va[k] = A::cmulc<DIR>(va[k], cc, ss); vb[k] = A::cmulc<DIR>(vb[k], cc, ss);
        PF_W16(1, c1, s1) PF_W16(2, h, h) PF_W16(3, s1, c1) PF_W16(5, -s1, c1) PF_W16(6, -h, h) PF_W16(7, -c1, s1)
#undef PF_W16
        va[4] = A::muli<DIR>(va[4]); vb[4] = A::muli<DIR>(vb[4]);
      }
      const float sg = odd ? -1.f : 1.f;  // X[ta] = Y0 + Y1 stays with h = 0, X[ta + 8] = Y0 - Y1 with h = 1: o + sgn v
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const C oa = pf_pk_xor8(va[m]), ob = pf_pk_xor8(vb[m]);
        va[m] = __builtin_elementwise_fma(va[m], (C){sg, sg}, oa);
        vb[m] = __builtin_elementwise_fma(vb[m], (C){sg, sg}, ob);
      }
    }
    // ---- exchange, stage 1 (NS = 16), exchange, stage 2 (NS = 128): positions of pf_stage_pos<N, S, true>
#pragma unroll
    for (int S = 0; S < 2; S++) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int pos = S == 0 ? pf_stage_pos<N, 0, true>(tlj, m) : pf_stage_pos<N, 1, true>(tlj, m);
        lds[pos * T + cj] = (pf_f4){va[m].x, va[m].y, vb[m].x, vb[m].y};
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; m++) { const pf_f4 t = lds[(tlj + m * NT) * T + cj]; va[m] = (C){t.x, t.y}; vb[m] = (C){t.z, t.w}; }
      __syncthreads();
      C wp[7];
      pfx_powers7<A>(S == 0 ? w1 : w2, wp);
#pragma unroll
      for (int r = 1; r < 8; r++) { va[r] = A::cmul<DIR>(va[r], wp[r - 1]); vb[r] = A::cmul<DIR>(vb[r], wp[r - 1]); }
      pfx_bfly8<A, DIR>(va);
      pfx_bfly8<A, DIR>(vb);
    }
    // ---- store: register m is element tl + 128 m
    {
      pfc<float> *__restrict__ out = reinterpret_cast<pfc<float> *>(p.job[j].out);
      const unsigned lane = pf_addr_lane(p.aout, tlj, 2 * cj);
      if (2 * (tile + 1) * T <= p.ncols) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
          long long u = pf_addr_uniform<NT>(p.aout, outer, m, 2 * tile * T);
          asm volatile("" : "+s"(u));
          pf16_st4(out, u, lane, va[m], vb[m]);
        }
      } else {
        const int colj = 2 * (tile * T + cj);
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const long long u = pf_addr_uniform<NT>(p.aout, outer, m, 2 * tile * T);
          if (colj + 1 < p.ncols) pf16_st4(out, u, lane, va[m], vb[m]);
          else if (colj < p.ncols) pf16_st2(out, u, lane, va[m]);
        }
      }
    }
  }
}

// 0: launched; 1: launch failed; 3: LDS opt-in refused; -1: not a case of these kernels (the caller runs k_strided)
int pf_launch_strided16(int fb, int n, int dir, const PfStridedParams &p, hipStream_t st) {
  if (fb != 4 || p.out_ne > 0) return -1;
  if (!((n == PF16_N && PF_STRIDED16) || (n == 1024 && PF_STRIDED_PK8))) return -1;
  constexpr int T = 8, NT = 128;  // both kernels: 128 threads per column pair, elements lane part (< 128) + 128 m
  // split addresses (pf_addr_uniform): a slab holds whole multiples of 128 elements of the transformed axis on both sides and the
  // per-lane part -- at most 127 rows plus a tile's columns -- fits 32 bits in bytes
  auto lane_fits = [&](const PfAddr &a) {
    return (1 << a.el_shift) >= NT && a.els > 0 && ((unsigned long long)(NT - 1) * (unsigned long long)a.els + (unsigned long long)(2 * T)) * sizeof(pfc<float>) < (1ull << 32);
  };
  if (!lane_fits(p.ain) || !lane_fits(p.aout)) return -1;
  const int ntiles = (p.ncols + 2 * T - 1) / (2 * T);
  const long long nwork = (long long)ntiles * p.nouter;
  const size_t shm = 128 * 1024;
  // instantiations: inverse passes with and without the first-pass filter, band-limited or not; forward passes carry neither
  const bool band = p.band_e < n / 2 || p.band_outer < n / 2;
  if (dir < 0 && (p.pre || band)) return -1;
  const int k = (n == 1024 ? 5 : 0) + (dir < 0 ? 4 : (p.pre ? 2 : 0) + (band ? 1 : 0));
  typedef void (*kern_t)(const PfStridedParams, const long long, const int);
  static const kern_t kern[10] = {k_strided16<+1, false, false>, k_strided16<+1, false, true>, k_strided16<+1, true, false>, k_strided16<+1, true, true>, k_strided16<-1, false, false>,
                                  k_strided_pk8<+1, false, false>, k_strided_pk8<+1, false, true>, k_strided_pk8<+1, true, false>, k_strided_pk8<+1, true, true>, k_strided_pk8<-1, false, false>};
  static std::atomic<bool> raised[10][PF_MAX_DEVICES];
  const int d = p.dev >= 0 && p.dev < PF_MAX_DEVICES ? p.dev : 0;
  if (!raised[k][d].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern[k]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
      fprintf(stderr, "ERROR on task 0: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) refused for the packed fp32 strided pass on device %d\n", shm, d);
      return 3;
    }
    raised[k][d].store(true, std::memory_order_release);
  }
  const long long nitems = (n == PF16_N && PF_S16_SPLIT) ? nwork * p.njobs : nwork;  // k_strided16: one job per workgroup
  dim3 grid((unsigned)(((nitems + 7) >> 3) << 3), 1, 1), block(1024, 1, 1);
  hipLaunchKernelGGL(kern[k], grid, block, shm, st, p, nitems, ntiles);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// test tap: the operations of the packed (re, im) algebra, one per `which`, on arrays of (re, im) pairs (tests/test_gpu_lines.py
// holds them against their definitions: the op_sel / neg modifiers of the inline assembly are what it is there to catch)
__global__ void k_debug_pk(int which, const pf_f2 *a, const pf_f2 *b, pf_f2 *out, int count) {
  using A = PfCxPk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const pf_f2 x = a[i], y = b[i];
    pf_f2 r = x;
    switch (which) {
      case 0: r = A::addi<+1>(x, y); break;
      case 1: r = A::addi<-1>(x, y); break;
      case 2: r = A::muli<+1>(x); break;
      case 3: r = A::muli<-1>(x); break;
      case 4: r = A::cmul<+1>(x, y); break;
      case 5: r = A::cmul<-1>(x, y); break;
      case 6: { const pf_f2 w = b[0]; r = A::cmul_s<+1>(x, w); } break;
      case 7: { const pf_f2 w = b[0]; r = A::cmul_s<-1>(x, w); } break;
      case 8: r = A::cmulc<+1>(x, 0.92387953251128675613, 0.38268343236508977173); break;
      case 9: r = A::cmulc<-1>(x, 0.92387953251128675613, 0.38268343236508977173); break;
      case 10: r = A::subi<+1>(x, y); break;
      default: break;
    }
    out[i] = r;
  }
}
extern "C" int pf_debug_pk(int which, const float *a, const float *b, float *out, int count) {
  pf_f2 *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t bytes = (size_t)count * sizeof(pf_f2);
  int rc = 1;
  if (hipMalloc((void **)&da, bytes) == hipSuccess && hipMalloc((void **)&db, bytes) == hipSuccess && hipMalloc((void **)&dout, bytes) == hipSuccess &&
      hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(k_debug_pk, dim3(64), dim3(256), 0, nullptr, which, da, db, dout, count);
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  hipFree(da); hipFree(db); hipFree(dout);
  return rc;
}
