// pf_fft_kernels.hip -- the three 1-D passes of the slab 3-D FFT (gfx950).
//
//   k_strided : x- or y-pass on the half-spectrum.  A workgroup owns T adjacent
//               kz columns (T*sizeof(complex) contiguous bytes per row) of one
//               line of tiles and all N points along the transformed axis; the
//               k-space filter of compute_derivative (src/fmax-pfft.c:306-397) is
//               applied on load: the Gaussian window, 1/k^2 and growth factor in
//               the first pass, the k_a / k_a^2 / i k_a factors in the pass along
//               axis a.  HBM bound: reads and writes each element once.
//   k_c2r     : z-pass, Hermitian rows of n/2+1 -> n reals through a half-length
//               complex FFT; applies the kz factor, 1/N^3 (src/fmax-pfft.c:220-225)
//               and the DC mode; writes fp64/fp32 fields or fp32 product columns.
//   k_r2c     : forward z-pass (real rows -> half-spectrum rows), LPT sources.
//
// A strided launch carries several (input, multiplier, output) jobs grouped by
// input: a workgroup reads its tile once and transforms it for every job on it.
#include <atomic>
#include <cstdio>

#include "pf_internal.h"
#include "pf_fft_core.h"

#include "pf_fft_stages.h"
#include "pf_fft16.h"   // pfx_bfly4 / pfx_bfly8 on the packed algebra (the two-row invariant z-pass)
#include "pf_strided_addr.h"
#include "pf_collapse_core.h"  // pf_invariants

#ifndef PF_SLAB_STORE
#define PF_SLAB_STORE 1  // (0 in an A/B build, profiles/tools/ab.sh: the kernel without the slab-store branch)
#endif
// How a thread of the strided passes reaches its column(s) in HBM.  One column per thread (fp64 fields; fp32 lines below 1024
// points): an element is one complex.  Two neighbouring columns per thread (F = pf_f32x2: fp32 lines of 1024 points and more,
// where the thread budget of a workgroup, not LDS, limits the tile): an element is 16 bytes -- both columns of a row in one
// load / store -- so that a tile's row segments are 128 bytes (64 at 2048 points) as in the fp64 passes, and the arithmetic is
// packed (pf_fft_core.h).  Offsets are in complex numbers of the lane type; `v1`: the second column exists (the number of
// columns, n/2 + 1 or band + 1, may be odd).
template <typename F> struct PfCols {
  using S = F;
  static constexpr int n = 1;
  static __device__ __forceinline__ pfc<F> load(const pfc<S> *base, long long off) { return pf_ld_stream(base + off); }
  static __device__ __forceinline__ void store(pfc<S> *base, long long off, pfc<F> v, bool) { pf_st_stream(base + off, v); }
  // uniform base (complex elements) + per-lane offset (complex elements, 32 bits)
  static __device__ __forceinline__ pfc<F> load_ul(const pfc<S> *base, long long u, unsigned lane) {
    return pf_ld_stream(reinterpret_cast<const pfc<S> *>(reinterpret_cast<const char *>(base + u) + (size_t)(lane * (unsigned)sizeof(pfc<S>))));
  }
  static __device__ __forceinline__ void store_ul(pfc<S> *base, long long u, unsigned lane, pfc<F> v, bool) {
    pf_st_stream(reinterpret_cast<pfc<S> *>(reinterpret_cast<char *>(base + u) + (size_t)(lane * (unsigned)sizeof(pfc<S>))), v);
  }
};
template <> struct PfCols<pf_f32x2> {
  using S = float;
  static constexpr int n = 2;
  typedef float f4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ pfc<pf_f32x2> load(const pfc<float> *base, long long off) {
#if PF_NT
    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(base + off));
#else
    const f4 t = *reinterpret_cast<const f4 *>(base + off);
#endif
    pfc<pf_f32x2> r;
    r.x = (pf_f32x2){t.x, t.z}; r.y = (pf_f32x2){t.y, t.w};
    return r;
  }
  static __device__ __forceinline__ void store(pfc<float> *base, long long off, pfc<pf_f32x2> v, bool v1) {
    if (v1) {
      const f4 t = (f4){v.x.x, v.y.x, v.x.y, v.y.y};
#if PF_NT
      __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(base + off));
#else
      *reinterpret_cast<f4 *>(base + off) = t;
#endif
    } else {
      pf_st_stream(base + off, pf_mk<float>(v.x.x, v.y.x));
    }
  }
  static __device__ __forceinline__ pfc<pf_f32x2> load_ul(const pfc<float> *base, long long u, unsigned lane) {
    return load(reinterpret_cast<const pfc<float> *>(reinterpret_cast<const char *>(base + u) + (size_t)(lane * (unsigned)sizeof(pfc<float>))), 0);
  }
  static __device__ __forceinline__ void store_ul(pfc<float> *base, long long u, unsigned lane, pfc<pf_f32x2> v, bool v1) {
    store(reinterpret_cast<pfc<float> *>(reinterpret_cast<char *>(base + u) + (size_t)(lane * (unsigned)sizeof(pfc<float>))), 0, v, v1);
  }
};

// FA: addresses as a uniform 64-bit part plus a 32-bit part per lane (pf_addr_uniform; the launcher checks the conditions)
template <typename F, int N, int T, int DIR, bool FA>
__global__ void __launch_bounds__(T *N / 8) __attribute__((amdgpu_waves_per_eu(4))) k_strided(const PfStridedParams p, const long long nwork, const int ntiles) {
  using C = pfc<F>;
  using COLS = PfCols<F>;
  using S = typename COLS::S;  // scalar of the field in memory
  constexpr int NL = COLS::n;  // columns per thread
  constexpr int NT = N / 8;
#ifndef PF_PAIR16
#define PF_PAIR16 1  // (0 in an A/B build: the plain plan, whose 1024-point line ends on a radix-2 stage behind a third LDS exchange)
#endif
  // the paired first stage (pf_fft_core.h): lengths 16 * 8^k in tiles of eight columns -- the partner thread tl ^ 1 is then
  // eight lanes away, in the same row of sixteen lanes.  A thread's register m then holds point pf_line_index(tl, m).
  constexpr bool P16 = PF_PAIR16 && pf_pair16_length(N) && T == 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);  // [N][T]
  const long long w = pf_xcd_swizzle(blockIdx.x, (nwork + 7) >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x;
  const int c = tid % T, tl = tid / T;
  const int tile = (int)(w % ntiles);
  const int outer = (int)(w / ntiles);
  const int col = NL * (tile * T + c);    // (first) column of this thread
  const bool valid = col < p.ncols, valid1 = NL > 1 && col + 1 < p.ncols;
  const pfc<S> *__restrict__ tw = reinterpret_cast<const pfc<S> *>(p.tw);
  const double kf = 2.0 * 3.14159265358979323846 / (double)N;
  if (p.band_outer < N / 2) {  // whole line of tiles outside the band of the smoothed spectrum: its output is never read
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    if (so > p.band_outer || so < -p.band_outer) return;
  }

  C src[8], v[8];
  // a tile of input `in` into src: issued as early as src is free, consumed at the top of the job that uses it
  // (two columns: the pair is loaded whole when its first column exists -- a row's padding holds the odd one out -- and the
  //  second lane is cleared where it does not, so that nothing undefined enters the arithmetic)
  auto load_tile = [&](const void *inp, int tlj, int colj) {
    const pfc<S> *__restrict__ in = reinterpret_cast<const pfc<S> *>(inp);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = pf_line_index<N, P16>(tlj, m);  // (= e0 + m NT with e0 = pf_line_index(tl, 0) < NT in either plan)
      const int se = e > N / 2 ? e - N : e;
      const bool inband = se <= p.band_e && se >= -p.band_e;
      if constexpr (FA) src[m] = (valid && inband) ? COLS::load_ul(in, pf_addr_uniform<NT>(p.ain, outer, m, NL * tile * T), pf_addr_lane(p.ain, pf_line_index<N, P16>(tlj, 0), colj - NL * tile * T)) : pf_zero<F>();
      else src[m] = (valid && inband) ? COLS::load(in, pf_addr(p.ain, outer, e, colj)) : pf_zero<F>();
      if constexpr (NL > 1) { if (!valid1) { src[m].x.y = 0.f; src[m].y.y = 0.f; } }
    }
  };
  load_tile(p.job[0].in, tl, col);
  // first pass: window of the two untransformed axes times the growth factor, one exp per thread and column (none without
  // smoothing).  Formed here, outside the loop over jobs: inside it the constants of exp() would live in ~24 registers
  // through every job of the kernel.
  double ko = 0.0, woc[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) woc[l] = 1.0;
  if (p.pre) {
    int so = outer + p.outer_offset;
    if (so > N / 2) so -= N;
    ko = kf * so;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double kc = kf * (col + l);
      const double ko2kc2 = ko * ko + kc * kc;
      woc[l] = (p.rs != 0.0 ? exp(-0.5 * ko2kc2 * p.rs * p.rs) : 1.0) * p.growth;
    }
  }
#pragma unroll 1
  for (int j = 0; j < p.njobs; j++) {
    // Everything below is loop invariant except the job; left to LICM the compiler hoists every
    // twiddle, LDS position and address out of the loop (~250 VGPRs, half the occupancy).  An
    // opaque copy of the thread coordinates per iteration keeps the index math where it is used.
    int tlj = tl, cj = c;
    asm volatile("" : "+v"(tlj), "+v"(cj));
    const int colj = NL * (tile * T + cj);
    // jobs are grouped by input: a tile is read from HBM once and transformed for every job that uses it
    if (p.pre && (j == 0 || p.job[j].in != p.job[j - 1].in)) {
      // the window along the transformed axis: the thread's eight table values requested together (as `p.rs != 0.0 ? p.etab[e] : 1.0`
      // inside the loop below each load sat in a branch of its own, behind its own wait: eight L2 latencies in a row, with one
      // workgroup per CU and nothing else to run)
      double we[8];
      if (p.rs != 0.0) {
#pragma unroll
        for (int m = 0; m < 8; m++) we[m] = p.etab[pf_line_index<N, P16>(tlj, m)];
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) we[m] = 1.0;
      }
      // 1 / k^2 and the window, column after column (one set of fp64 temporaries alive at a time)
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double kc = kf * (colj + l);
        const double ko2kc2 = ko * ko + kc * kc;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int e = pf_line_index<N, P16>(tlj, m);
          const double ke = kf * (e > N / 2 ? e - N : e);
          const double k2 = ke * ke + ko2kc2;
          // (the quotient from the hardware reciprocal, one Newton step and the residual correction: the correctly rounded one but for
          //  rare 1-ulp cases, in 6 operations where the IEEE division takes 12 -- eight per thread and input tile)
          const S fac = (S)((k2 != 0.0) ? pf_div_fast(we[m] * woc[l], k2) : 0.0);
          if constexpr (NL > 1) { src[m].x[l] = src[m].x[l] * fac; src[m].y[l] = src[m].y[l] * fac; }
          else src[m] = pf_scale(src[m], (F)fac);
        }
      }
    }
    const int mul = p.job[j].mul;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = pf_line_index<N, P16>(tlj, m);
      const F ke = (F)(S)(kf * (e > N / 2 ? e - N : e));
      C x = src[m];
      if (mul == PF_MUL_K) x = pf_scale(x, ke);
      else if (mul == PF_MUL_K2) x = pf_scale(x, ke * ke);
      else if (mul == PF_MUL_IK) x = pf_mul_i<+1>(pf_scale(x, ke));
      v[m] = x;
    }
    // src is free once the last job on this input has taken its copy: the next input's tile travels during the stages
    if (j + 1 < p.njobs && p.job[j + 1].in != p.job[j].in) load_tile(p.job[j + 1].in, tlj, colj);
    PfStages<F, N, DIR, 1, 0, false, P16>::run(
        v, tlj, tw, [&](int pos, C val) { lds[pos * T + cj] = val; }, [&](int pos) { return lds[pos * T + cj]; });
#if defined(PF_DUMMY_VALU) && PF_DUMMY_VALU > 0  // (A/B probe: how much of the arithmetic of a job is hidden behind its memory traffic)
    if constexpr (NL == 1) {
      double a = (double)v[0].x;
      for (int i = 0; i < PF_DUMMY_VALU; i++) a = __builtin_fma(a, 1.0000001, 1e-30);
      v[0].x = (F)a;
    }
#endif
    if (valid) {
      pfc<S> *__restrict__ out = reinterpret_cast<pfc<S> *>(p.job[j].out);
      if (PF_SLAB_STORE && p.out_ne > 0) {  // uniform: the slab of the transformed axis this rank keeps
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int e = tlj + m * NT;
          if ((unsigned)(e - p.out_e0) < (unsigned)p.out_ne) COLS::store(out, pf_addr(p.aout, outer, e, colj), v[m], valid1);
        }
      } else if constexpr (FA) {
        const unsigned lane = pf_addr_lane(p.aout, tlj, colj - NL * tile * T);
#pragma unroll
        for (int m = 0; m < 8; m++) COLS::store_ul(out, pf_addr_uniform<NT>(p.aout, outer, m, NL * tile * T), lane, v[m], valid1);
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int e = tlj + m * NT;
          COLS::store(out, pf_addr(p.aout, outer, e, colj), v[m], valid1);
        }
      }
    }
  }
}

template <typename F, int N, int TL>
__global__ void __launch_bounds__(TL *(N / 16)) k_c2r(const PfC2RParams p) {
  using C = pfc<F>;
  constexpr int M = N / 2, NT = M / 8;
  constexpr int LPL = (M + 1 > M + M / 8) ? M + 1 : M + M / 8;  // LDS complex per line
  constexpr int NTHR = TL * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tid = threadIdx.x;
  const int l = tid / NT, tl = tid % NT;
  const int job = blockIdx.x % p.njobs;
  const long long line0 = (long long)(blockIdx.x / p.njobs) * TL;
  const C *__restrict__ in = reinterpret_cast<const C *>(p.job[job].in);
  const int mul = p.job[job].mul;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);

  // phase A: rows -> LDS, coalesced
  for (int idx = tid; idx < TL * (M + 1); idx += NTHR) {
    const int ll = idx / (M + 1), k = idx % (M + 1);
    const long long row = line0 + ll;
    lds[ll * LPL + k] = (row < p.nlines && k <= p.band_k) ? in[row * p.in_pitch + k] : pf_mk<F>(0, 0);
  }
  __syncthreads();

  // phase B: kz factor + Hermitian fold into the half-length complex line
  C *L = lds + l * LPL;
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)N);
  C v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int e = tl + m * NT;
    v[m] = pf_zfold<F>(L[e], L[M - e], e, M, mul, kf, tw[e], m == 0);
  }
  __syncthreads();

  PfStages<F, M, +1, 2>::run(
      v, tl, tw, [&](int pos, C val) { L[pf_lpad(pos)] = val; }, [&](int pos) { return L[pf_lpad(pos)]; });

  const long long row = line0 + l;
  if (row < p.nlines) {
    const F norm = (F)p.norm;
    const F dcv = p.dc ? (F)(*p.dc) : (F)0;
    if (p.job[job].out_f32 == 1) {
      float2 *o = reinterpret_cast<float2 *>(reinterpret_cast<float *>(p.job[job].out) + row * (long long)N);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int n2 = tl + m * NT;
        o[n2] = make_float2((float)pf_norm_dc(v[m].x, norm, dcv), (float)pf_norm_dc(v[m].y, norm, dcv));
      }
    } else {
      C *o = reinterpret_cast<C *>(reinterpret_cast<F *>(p.job[job].out) + row * (p.job[job].out_f32 == 2 ? (long long)N : p.out_pitch));
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int n2 = tl + m * NT;
        o[n2] = pf_norm_dc2(v[m], norm, dcv);
      }
    }
  }
}

// Persistent form of k_c2r: a workgroup walks over tiles (job, TL rows) with a stride of gridDim.x and keeps the NEXT
// tile's rows in flight in registers while it transforms the current one.  The one-shot kernel has loads outstanding
// only during its first phase (~50 KB per CU on average at 4 resident workgroups -- about what 6 TB/s x 2 us of latency
// needs, with nothing to spare); here every resident wave always has 8 KB on the way.
template <typename F, int N, int TL>
__global__ void __launch_bounds__(TL *(N / 16)) k_c2r_persistent(const PfC2RParams p, long long ntiles) {
  using C = pfc<F>;
  constexpr int M = N / 2, NT = M / 8;
  constexpr int LPL = (M + 1 > M + M / 8) ? M + 1 : M + M / 8;
  constexpr int NTHR = TL * NT;
  constexpr int NLD = (TL * (M + 1) + NTHR - 1) / NTHR;  // elements of a tile per thread (8 + the Nyquist column)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tid = threadIdx.x;
  const int l = tid / NT, tl = tid % NT;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)N);
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;

  C nxt[NLD];
  auto fetch = [&](long long t) {
    const int job = (int)(t % p.njobs);
    const long long line0 = (t / p.njobs) * TL;
    const C *__restrict__ in = reinterpret_cast<const C *>(p.job[job].in);
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int idx = tid + i * NTHR;
      const int ll = idx / (M + 1), k = idx % (M + 1);
      const long long row = line0 + ll;
      nxt[i] = (idx < TL * (M + 1) && row < p.nlines && k <= p.band_k) ? pf_ld_stream(&in[row * p.in_pitch + k]) : pf_mk<F>(0, 0);
    }
  };

  long long t = blockIdx.x;
  if (t < ntiles) fetch(t);
#pragma unroll 1
  for (; t < ntiles; t += gridDim.x) {
    int tidj = tid, tlj = tl, lj = l;
    asm volatile("" : "+v"(tidj), "+v"(tlj), "+v"(lj));  // keep the index math inside the loop (see k_strided)
    // phase A: prefetched rows -> LDS
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int idx = tidj + i * NTHR;
      if (idx < TL * (M + 1)) lds[(idx / (M + 1)) * LPL + idx % (M + 1)] = nxt[i];
    }
    __syncthreads();
    // phase B: kz factor + Hermitian fold into the half-length complex line
    const int job = (int)(t % p.njobs);
    const int mul = p.job[job].mul;
    C *L = lds + lj * LPL;
    C v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tlj + m * NT;
      v[m] = pf_zfold<F>(L[e], L[M - e], e, M, mul, kf, tw[e], m == 0);
    }
    // from here to the end of the stages a line belongs to its own NT threads: one wave for N <= 1024, whose LDS accesses
    // stay in order without workgroup barriers (four of them per tile at N = 1024)
    if (NT <= 64) pf_wave_sync(); else __syncthreads();
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);  // in flight during the stages and the stores below
    PfStages<F, M, +1, 2, 0, (NT <= 64)>::run(
        v, tlj, tw, [&](int pos, C val) { L[pf_lpad(pos)] = val; }, [&](int pos) { return L[pf_lpad(pos)]; });
    const long long row = (t / p.njobs) * TL + lj;
    if (row < p.nlines) {
      if (p.job[job].out_f32 == 1) {
        float2 *o = reinterpret_cast<float2 *>(reinterpret_cast<float *>(p.job[job].out) + row * (long long)N);
#pragma unroll
        for (int m = 0; m < 8; m++)
          pf_st_stream(reinterpret_cast<pfc<float> *>(&o[tlj + m * NT]), pf_mk<float>((float)pf_norm_dc(v[m].x, norm, dcv), (float)pf_norm_dc(v[m].y, norm, dcv)));
      } else {
        C *o = reinterpret_cast<C *>(reinterpret_cast<F *>(p.job[job].out) + row * (p.job[job].out_f32 == 2 ? (long long)N : p.out_pitch));
#pragma unroll
        for (int m = 0; m < 8; m++) pf_st_stream(&o[tlj + m * NT], pf_norm_dc2(v[m], norm, dcv));
      }
    }
    // the next phase A rewrites every line: all waves must be done with theirs
    if (NT <= 64 || pf_nstages(M) == 1) __syncthreads();
  }
}

// The z-pass of the sweep, six components of one row per workgroup: line l of the tile is row `R` of component l, and
// after the six transforms the workgroup reduces each cell's tensor to its three invariants mu1, mu2, mu3 -- all the
// collapse solve reads (pf_eigen_from_invariants) -- and stores three real rows instead of six (into the first three
// fields, in place: every input of row R is in registers or LDS before the first store).  Per radius the pass writes
// 26 GB instead of 52 and the solve reads 26 instead of 52.  The component values are formed by the same expression as in
// k_c2r (pf_norm_dc) and the invariants by the solve's own pf_invariants (no contraction): Fmax, Rmax and the variances
// come out bit for bit as with the six-field path.  fp64 fields store the invariants in place (rows of the first three
// fields); fp32 fields (F = float: the transforms in fp32 as in k_c2r, the reduction in fp64 from the fp32 components exactly
// as k_collapse<float> forms it) store them as fp64 rows of pitch inv_pitch in the separate buffers inv_out[0..2].
// MODE 1 (the Hessian of the 2LPT potential, src/LPT.c:112-137): the six rows are not stored at all; each cell's 3LPT(b)
// source is updated in place from them and the six components of the first-order Hessian (read from job[c].out):
// p.acc -= 2 phi2_ab h_ab, by the same pf_lpt3b_accumulate as k_lpt_accum.
// Round 4: fp64 rows of 512 points and more run THREE workgroups per CU instead of two.  The row of an iteration is loaded at the
// top of that iteration (PF_ZI_PREFETCH = 0: no row carried in registers across the stages: 78 instead of 114 registers) and the
// exchange area of a line is padded by p >> 4 instead of p >> 3 (544 instead of 576 complex per 512-point line: 52.2 KB per
// workgroup; the stride-8 writes of stage 0 still fall on distinct banks) -- eighteen waves per CU hide a row's load latency better than twelve with a row in flight each:
// 16.8 -> 15.7 ms per launch at 1024^3, 743 -> 722 ms per step (A/B on one box, profiles/r04_notes.md).  Carrying the row and asking
// for five waves per SIMD spills (92 bytes) and takes 26.7 ms; requesting the next row late, behind the stages, needs 114 registers
// again.  fp32 rows (28 KB of lines, 86 registers) already fit three workgroups with the row carried and keep it.
#ifndef PF_ZI_PREFETCH
#define PF_ZI_PREFETCH -1   // -1: by field type and length (above); 0 / 1: force (A/B builds)
#endif
#ifndef PF_ZI_WAVES
#define PF_ZI_WAVES 0       // (A/B: waves per SIMD the compiler is asked to make room for; 0: its own choice)
#endif
#define PF_ZI_PRAGMA_(x) _Pragma(#x)
#define PF_ZI_PRAGMA(x) PF_ZI_PRAGMA_(x)
#ifndef PF_ZI_SPEC_F32
#define PF_ZI_SPEC_F32 1    // the same for fp32 rows of 1024 points (60 registers: four workgroups per CU; 13.0 -> 10.6 ms per launch)
#endif
#ifndef PF_ZI_SPEC_512
#define PF_ZI_SPEC_512 1    // ... and for rows of 512 points (two lines per wave: three transform waves and one that reduces; 512^3: 1.86 -> 1.80 ms
                            // per launch, fp32 fields 1.48 -> 1.36)
#endif
#ifndef PF_ZI_SPEC_2048
#define PF_ZI_SPEC_2048 1   // ... and for fp32 rows of 2048 points (BASELINE config 5): twelve transform waves, four that reduce
#endif
#ifndef PF_ZI_SPEC_LPT
#define PF_ZI_SPEC_LPT 0    // (A/B) ... and for MODE 1, the contraction into the 3LPT(b) source: its reduction reads seven more fields, and two
                            // waves are too few to keep those loads in flight (23.5 -> 24.4 ms)
#endif
#ifndef PF_ZI_RUNROLL
#define PF_ZI_RUNROLL 1     // (A/B) unrolling of the per-cell reduction loop
#endif
#ifndef PF_ZI_LATE
#define PF_ZI_LATE 0
#endif
#ifndef PF_ZI_SPEC
#define PF_ZI_SPEC 1        // (0 in an A/B build: every wave transforms a line and takes part in the reduction)
#endif
#ifndef PF_ZI_UNIFORM_MUL
#define PF_ZI_UNIFORM_MUL(F) (sizeof(F) == 4)   // (fp64 rows: the kernel sits at its 80 registers, and with the factor in a scalar register the compiler's schedule spills)
#endif
#ifndef PF_ZFOLD_K0_ALL
#define PF_ZFOLD_K0_ALL(F) (sizeof(F) == 8)
#endif
#ifndef PF_ZI_ROT
#define PF_ZI_ROT 2         // wave w of a workgroup takes component (w + PF_ZI_ROT) % 6: with 2 the two components with a kz factor fall on
                            // waves 2 and 3, which have a SIMD to themselves within the workgroup (A/B: 14.84 -> 14.68 ms per launch)
#endif
#ifndef PF_ZI_OWN
#define PF_ZI_OWN 1         // (A/B) 0: phase B reads both operands of the fold from LDS
#endif
#ifndef PF_ZI_DMA
#define PF_ZI_DMA 0         // (A/B) 1: 1024-point fp64 rows are loaded straight into LDS
#endif
template <typename F, int M> struct PfZiPlan {
  static constexpr bool lean = PF_ZI_PREFETCH < 0 ? (sizeof(F) == 8 && M >= 256) : PF_ZI_PREFETCH == 0;  // no carried row, narrow pad
  static constexpr int full = (M + 1 > M + M / 8) ? M + 1 : M + M / 8, narrow = (M + 1 > M + M / 16) ? M + 1 : M + M / 16;
  static constexpr int line = (lean && M >= 256) ? narrow : full;  // complex per LDS line
  static __device__ __forceinline__ int pad(int p) { return (lean && M >= 256) ? p + (p >> 4) : pf_lpad(p); }
  // SPEC (fp64 rows of 1024 points, the invariants of the sweep): two more waves per workgroup do nothing but the per-cell reduction
  // of a row, and the six transform waves request their next row the moment they have handed this one over -- the request then
  // travels during the reduction and a row's load latency is off the workgroup's critical path (profiles/r04_notes.md, section 3f)
  static constexpr bool spec0 = PF_ZI_SPEC && !PF_ZI_DMA &&
                                (((M == 512 || (M == 256 && PF_ZI_SPEC_512)) && (lean || (PF_ZI_SPEC_F32 && sizeof(F) == 4))) ||
                                 (M == 1024 && sizeof(F) == 4 && PF_ZI_SPEC_2048));
  static constexpr bool spec(int mode) { return spec0 && (mode == 0 || PF_ZI_SPEC_LPT); }
  static constexpr int reducer_waves = M / 256;  // 2 for rows of 1024 points, 4 for 2048 (two waves per line there: sixteen waves in all)
};
template <typename F, int N, int MODE, bool SPEC>
__device__ __forceinline__ void pf_c2r_invariants_body(const PfC2RParams &p, long long nrows) {
  using C = pfc<F>;
  using F2 = typename pf_vec2<F>::type;
  constexpr bool IN_PLACE = sizeof(F) == 8;  // fp64 invariants fit the rows they replace
  constexpr int M = N / 2, NT = M / 8, TL = 6;
  using PLAN = PfZiPlan<F, M>;
  constexpr int LPL = PLAN::line;
  constexpr bool DMA = PF_ZI_DMA && PLAN::lean && NT == 64 && sizeof(C) == 16;  // rows straight into LDS (below)
  constexpr bool PREFETCH = !PLAN::lean && !SPEC;
  constexpr bool LATE = PF_ZI_LATE && PLAN::lean && !SPEC && !DMA && MODE == 0;  // (A/B) six waves, each requesting its next row in front of the reduction
  constexpr int NTHR = TL * NT;
  static_assert((size_t)LPL * sizeof(C) >= (size_t)N * sizeof(F), "a line's LDS holds its real row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tid = threadIdx.x;
  const int lw = tid / NT, tl = tid % NT;
  const bool reducer = SPEC && lw >= TL;                    // (uniform per wave) waves 6, 7: the reduction only
  const int l = reducer ? 0 : (lw + PF_ZI_ROT) % TL;        // the component this wave transforms (PF_ZI_ROT: A/B)
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)N);
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;
  // this thread's component: input rows and kz factor
  const C *in = reinterpret_cast<const C *>(p.job[0].in);
  int mul = p.job[0].mul;
#pragma unroll
  for (int j = 1; j < TL; j++)
    if (l == j) { in = reinterpret_cast<const C *>(p.job[j].in); mul = p.job[j].mul; }
  // (a line of 64 threads and more is a whole number of waves: its kz factor is the same for every lane -- scalar branches, no exec masks)
  if constexpr (NT >= 64 && PF_ZI_UNIFORM_MUL(F)) mul = __builtin_amdgcn_readfirstlane(mul);
  double *__restrict__ o1 = IN_PLACE ? reinterpret_cast<double *>(p.job[0].out) : p.inv_out[0],
         *__restrict__ o2 = IN_PLACE ? reinterpret_cast<double *>(p.job[1].out) : p.inv_out[1],
         *__restrict__ o3 = IN_PLACE ? reinterpret_cast<double *>(p.job[2].out) : p.inv_out[2];

  // a line is touched by its own threads only until the reduction: while those sit in one wave (N <= 1024) the LDS
  // queue keeps their accesses in order and no workgroup barrier is needed
  auto line_sync = [&]() { if (NT > 64) __syncthreads(); else pf_wave_sync(); };
  C nxt[9];
  auto fetch = [&](long long R) {
    const C *__restrict__ row = in + R * p.in_pitch;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = tl + m * NT;
      nxt[m] = (k <= p.band_k) ? pf_ld_stream(&row[k]) : pf_mk<F>(0, 0);
    }
    nxt[8] = (tl == 0 && M <= p.band_k) ? pf_ld_stream(&row[M]) : pf_mk<F>(0, 0);
  };

  // per cell: six components -> three invariants; two neighbouring cells per thread (16-byte LDS reads and stores)
  auto reduce_row = [&](long long R, int c0, int cstep) {
PF_ZI_PRAGMA(unroll PF_ZI_RUNROLL)
    for (int c = c0; c < N; c += cstep) {
      F2 h[6];
#pragma unroll
      for (int k = 0; k < 6; k++) h[k] = *reinterpret_cast<const F2 *>(reinterpret_cast<const F *>(lds + k * LPL) + c);
      const double da[6] = {(double)h[0].x, (double)h[1].x, (double)h[2].x, (double)h[3].x, (double)h[4].x, (double)h[5].x},
                   db[6] = {(double)h[0].y, (double)h[1].y, (double)h[2].y, (double)h[3].y, (double)h[4].y, (double)h[5].y};
      const long long a = R * p.out_pitch + c;
      if (MODE == 0) {
        double a1, a2, a3, b1, b2, b3;
        pf_invariants(da, a1, a2, a3);
        pf_invariants(db, b1, b2, b3);
        // the solve's q == 0 case takes the tensor's own diagonal (src/collapse_times.c:722-727), which is not stored: it
        // will use mu1/3 three times.  The same unless the tensor is exactly that; otherwise the sweep is repeated (pf_sweep)
        if (pf_invariants_lose_diagonal(da, a1, a2) || pf_invariants_lose_diagonal(db, b1, b2)) *p.flag = 1.0;
        const long long ao = IN_PLACE ? a : R * p.inv_pitch + c;
#if PF_NT  // (written once, read once by the solve: streaming stores, pf_fft_core.h)
        typedef double pf_d2 __attribute__((ext_vector_type(2)));
        pf_d2 w1, w2, w3;
        w1.x = a1; w1.y = b1; w2.x = a2; w2.y = b2; w3.x = a3; w3.y = b3;
        __builtin_nontemporal_store(w1, reinterpret_cast<pf_d2 *>(o1 + ao));
        __builtin_nontemporal_store(w2, reinterpret_cast<pf_d2 *>(o2 + ao));
        __builtin_nontemporal_store(w3, reinterpret_cast<pf_d2 *>(o3 + ao));
#else
        *reinterpret_cast<double2 *>(o1 + ao) = make_double2(a1, b1);
        *reinterpret_cast<double2 *>(o2 + ao) = make_double2(a2, b2);
        *reinterpret_cast<double2 *>(o3 + ao) = make_double2(a3, b3);
#endif
      } else {
        double ha[6], hb[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const F2 g = *reinterpret_cast<const F2 *>(reinterpret_cast<const F *>(p.job[k].out) + a);
          ha[k] = (double)g.x; hb[k] = (double)g.y;
        }
        F2 *acc = reinterpret_cast<F2 *>(reinterpret_cast<F *>(p.acc) + a);
        const F2 s = *acc;
        F2 r;
        r.x = (F)pf_lpt3b_accumulate((double)s.x, da, ha);
        r.y = (F)pf_lpt3b_accumulate((double)s.y, db, hb);
        *acc = r;
      }
    }
  };
  if (SPEC && reducer) {  // the reducing waves: between the two barriers of a row, its reduction (a loop of their own: nothing of the transform waves' state is alive here)
    // (lines of two waves -- 2048 points -- synchronise their halves by workgroup barriers: the reducing waves join those too)
    constexpr int LINE_BARRIERS = NT > 64 ? 3 + 2 * (pf_nstages(M) - 1) : 0;
#pragma unroll 1
    for (long long Rr = blockIdx.x; Rr < nrows; Rr += gridDim.x) {
      int tr = tid - TL * NT;
      asm volatile("" : "+v"(tr));
#pragma unroll
      for (int i = 0; i < LINE_BARRIERS; i++) __syncthreads();
      __syncthreads();
      reduce_row(Rr, 2 * tr, 2 * 64 * PLAN::reducer_waves);
      __syncthreads();
    }
    return;
  }
  long long R = blockIdx.x;
  if ((PREFETCH || SPEC || LATE) && R < nrows) fetch(R);
#pragma unroll 1
  for (; R < nrows; R += gridDim.x) {
    int tlj = tl, lj = l, tidj = tid;
    asm volatile("" : "+v"(tlj), "+v"(lj), "+v"(tidj));  // keep the index math inside the loop (see k_strided)
    C *L = lds + lj * LPL;
    if (DMA) {
      // phase A without registers: each 64-lane piece of the row that lies inside the band goes from memory straight into the
      // line (global_load_lds_dwordx4: wave-uniform LDS base + 16 bytes per lane, which is the line's own order); pieces cut by
      // the band or beyond it are written as before.  The wave waits for its own transfers (vmcnt) before it reads the line.
      const C *__restrict__ row = in + R * p.in_pitch;
      const unsigned lbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char *)(lds + (tid / NT) * LPL));
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int k = tlj + m * NT;
        if (m * NT + NT - 1 <= p.band_k)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(&row[k]),
                                           (__attribute__((address_space(3))) void *)(size_t)(lbase + m * NT * (unsigned)sizeof(C)), 16, 0, PF_NT ? 2 : 0);
        else
          L[k] = (k <= p.band_k) ? pf_ld_stream(&row[k]) : pf_mk<F>(0, 0);
      }
      if (tlj == 0) L[M] = (M <= p.band_k) ? pf_ld_stream(&row[M]) : pf_mk<F>(0, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) (gfx9 encoding: lgkmcnt and expcnt left at their maxima)
    } else {
      if (!PREFETCH && !SPEC && !LATE) fetch(R);
      // phase A: the prefetched row of this component -> its LDS line
#pragma unroll
      for (int m = 0; m < 8; m++) L[tlj + m * NT] = nxt[m];
      if (tlj == 0) L[M] = nxt[8];
    }
    line_sync();
    // phase B: kz factor + Hermitian fold into the half-length complex line
    C v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tlj + m * NT;
      // (element e itself is still in the register it was loaded into, where the row was loaded by this iteration)
      v[m] = pf_zfold<F>((PF_ZI_OWN && !PREFETCH && !DMA) ? nxt[m] : L[e], L[M - e], e, M, mul, kf, tw[e], m == 0 || PF_ZFOLD_K0_ALL(F));  // (SPEC: nxt holds this row since the end of the iteration before)
    }
    line_sync();
    if (PREFETCH && R + gridDim.x < nrows) fetch(R + gridDim.x);  // in flight during the stages and the reduction below
#ifndef PF_ZI_HOIST_TW
#define PF_ZI_HOIST_TW 0
#endif
    PfStages<F, M, +1, 2, 0, (NT <= 64)>::run(
        v, tlj, tw, [&](int pos, C val) { L[PLAN::pad(pos)] = val; }, [&](int pos) { return L[PLAN::pad(pos)]; }, PF_ZI_HOIST_TW ? tl : -1);
    line_sync();  // every thread of the line is done with the exchange area
    // the real row of this component, in order, into its line
    C *H = L;
#pragma unroll
    for (int m = 0; m < 8; m++) H[tlj + m * NT] = pf_norm_dc2(v[m], norm, dcv);
    __syncthreads();
    // the transform waves are done with this row: their next one travels while waves 6, 7 reduce.  (Past the last row: zeros, not
    // "nothing" -- a conditional request would keep the OLD row alive through the stages, in 36 registers the kernel does not have)
    if (SPEC || LATE) {
      if (R + gridDim.x < nrows) fetch(R + gridDim.x);
      else {
#pragma unroll
        for (int m = 0; m < 9; m++) nxt[m] = pf_mk<F>(0, 0);
      }
    }
    if (!SPEC) reduce_row(R, 2 * tidj, 2 * NTHR);  // (LATE: with every wave's next row in flight)
    __syncthreads();  // the lines are rewritten by the next phase A
  }
}
template <typename F, int N, int MODE = 0>
__global__ void __launch_bounds__(6 * (N / 16))
#if PF_ZI_WAVES > 0
__attribute__((amdgpu_waves_per_eu(PF_ZI_WAVES, PF_ZI_WAVES)))
#endif
k_c2r_invariants(const PfC2RParams p, long long nrows) { pf_c2r_invariants_body<F, N, MODE, false>(p, nrows); }
// eight waves: six transform, two reduce (PfZiPlan::spec).  Six waves per SIMD = three workgroups per CU is what the 52 KB of lines
// allow: the kernel must stay within 80 registers.
template <typename F, int N, int MODE = 0>
__global__ void __launch_bounds__(6 * (N / 16) + 64 * (N / 512)) __attribute__((amdgpu_waves_per_eu(N >= 2048 ? 8 : 6)))
k_c2r_invariants_spec(const PfC2RParams p, long long nrows) { pf_c2r_invariants_body<F, N, MODE, true>(p, nrows); }

// ---- the invariant z-pass of fp32 fields, TWO ROWS PER THREAD (round 6) ------------------------------------------------------------
// k_c2r_invariants_spec<float, N, 0> in the packed algebra still spends half of its vector instructions on what is not arithmetic --
// indices, band tests, LDS addresses, twiddle powers, waits -- and moves its lines through LDS in 8-byte accesses, with a barrier pair
// per exchange.  None of that depends on the row.  Here a thread owns the same eight points of TWO neighbouring rows (2 q, 2 q + 1) of
// its component: every complex is a (re, im) register pair as it lies in memory (PfCxPk), an LDS element is the pair of both rows
// (16 bytes: ds_read_b128 / ds_write_b128), and the index arithmetic, the twiddles and their powers, the barriers and the waits are
// paid once for two rows.  Waves that only reduce as before; the next pair of rows is requested when this one is handed over.
// Same operations per cell as the one-row kernel (the same PfCxPk expressions in the same order): the same bits.
#ifndef PF_ZI_PK2
#define PF_ZI_PK2 1         // (0 in an A/B build: k_c2r_invariants_spec<float, N, 0> as in round 5)
#endif
#ifndef PF_ZI_PK2_MAXN
#define PF_ZI_PK2_MAXN 1024  // measured on one box (profiles/r06_notes.md): 512 points 1.31 -> 1.10 ms per launch, 1024 points 9.66 -> 9.44, 2048 points (BASELINE
                            // config 5's slab) 121.7 -> 131.3 ms per step: two rows of 2048 points are 104 KB of lines, one workgroup per CU instead of two
#endif
struct alignas(16) PfRow2 { pf_f2 a, b; };   // one complex of row 2 q (a) and of row 2 q + 1 (b)

template <int M, int S, int NW>
__device__ __forceinline__ void pf_pk2_stage(pf_f2 (&a)[8], pf_f2 (&b)[8], const pf_f2 (&w)[NW]) {
  using A = PfCxPk;
  constexpr int R = pf_radix(M, S), NS = pf_ns(M, S), Q = 8 / R;
#pragma unroll
  for (int q = 0; q < Q; q++) {
    if constexpr (NS > 1) {
      const pf_f2 w1 = w[q];
      a[q + Q] = A::cmul<+1>(a[q + Q], w1); b[q + Q] = A::cmul<+1>(b[q + Q], w1);
      if constexpr (R >= 4) {
        const pf_f2 w2 = A::twmul(w1, w1), w3 = A::twmul(w2, w1);
        a[q + 2 * Q] = A::cmul<+1>(a[q + 2 * Q], w2); b[q + 2 * Q] = A::cmul<+1>(b[q + 2 * Q], w2);
        a[q + 3 * Q] = A::cmul<+1>(a[q + 3 * Q], w3); b[q + 3 * Q] = A::cmul<+1>(b[q + 3 * Q], w3);
        if constexpr (R == 8) {
          const pf_f2 w4 = A::twmul(w2, w2), w5 = A::twmul(w4, w1), w6 = A::twmul(w3, w3), w7 = A::twmul(w4, w3);
          a[4] = A::cmul<+1>(a[4], w4); b[4] = A::cmul<+1>(b[4], w4);
          a[5] = A::cmul<+1>(a[5], w5); b[5] = A::cmul<+1>(b[5], w5);
          a[6] = A::cmul<+1>(a[6], w6); b[6] = A::cmul<+1>(b[6], w6);
          a[7] = A::cmul<+1>(a[7], w7); b[7] = A::cmul<+1>(b[7], w7);
        }
      }
    }
    if constexpr (R == 8) { pfx_bfly8<A, +1>(a); pfx_bfly8<A, +1>(b); }
    else if constexpr (R == 4) {
      pfx_bfly4<A, +1>(a[q], a[q + Q], a[q + 2 * Q], a[q + 3 * Q]);
      pfx_bfly4<A, +1>(b[q], b[q + Q], b[q + 2 * Q], b[q + 3 * Q]);
    } else {
      const pf_f2 ta = A::sub(a[q], a[q + Q]), tb = A::sub(b[q], b[q + Q]);
      a[q] = A::add(a[q], a[q + Q]); b[q] = A::add(b[q], b[q + Q]);
      a[q + Q] = ta; b[q + Q] = tb;
    }
  }
}
// stages S .. last of an M-point line on both rows (stage S already applied on entry when S > 0)
template <int M, int S, bool WAVE_LOCAL>
struct PfPk2Stages {
  template <typename WR, typename RD>
  static __device__ __forceinline__ void run(pf_f2 (&a)[8], pf_f2 (&b)[8], int tl, const pf_f2 *__restrict__ tw, WR wr, RD rd) {
    constexpr int NT = M / 8;
    if constexpr (S == 0) {
      const pf_f2 none[1] = {PfCxPk::mk(1.f, 0.f)};
      pf_pk2_stage<M, 0>(a, b, none);
    }
    if constexpr (S + 1 < pf_nstages(M)) {
      constexpr int R1 = pf_radix(M, S + 1), NS1 = pf_ns(M, S + 1), Q1 = 8 / R1;
      constexpr int TWM1 = (M / (NS1 * R1)) * 2;   // the table is exp(+2 pi i j / N), N = 2 M
      pf_f2 w[Q1];
#pragma unroll
      for (int q = 0; q < Q1; q++) w[q] = tw[((tl + q * NT) & (NS1 - 1)) * TWM1];   // asked for in front of the exchange
#pragma unroll
      for (int m = 0; m < 8; m++) wr(pf_stage_pos<M, S>(tl, m), a[m], b[m]);
      if constexpr (WAVE_LOCAL) pf_wave_sync(); else __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; m++) rd(tl + m * NT, a[m], b[m]);
      if constexpr (WAVE_LOCAL) pf_wave_sync(); else __syncthreads();
      pf_pk2_stage<M, S + 1>(a, b, w);
      PfPk2Stages<M, S + 1, WAVE_LOCAL>::run(a, b, tl, tw, wr, rd);
    }
  }
};

template <int N, int RW>
__global__ void __launch_bounds__(6 * (N / 16) + 64 * RW) k_c2r_invariants_pk2(const PfC2RParams p, long long nrows) {
  using A = PfCxPk;
  constexpr int M = N / 2, NT = M / 8, TL = 6;
  constexpr int LPL = M + M / 16 + 1;                 // 16-byte elements of a line (exchange positions padded by p >> 4)
  constexpr bool WAVE_LOCAL = NT <= 64;               // a line's threads sit in one wave
  constexpr int NSTAGES = pf_nstages(M);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PfRow2 *lds = reinterpret_cast<PfRow2 *>(smem);
  const int tid = threadIdx.x;
  const long long npairs = (nrows + 1) >> 1;
  double *__restrict__ o1 = p.inv_out[0], *__restrict__ o2 = p.inv_out[1], *__restrict__ o3 = p.inv_out[2];
  if (tid >= TL * NT) {
    // ---- the reducing waves: between the two barriers of a pair of rows, its reduction
    constexpr int LINE_BARRIERS = WAVE_LOCAL ? 0 : 3 + 2 * (NSTAGES - 1);
#pragma unroll 1
    for (long long q = blockIdx.x; q < npairs; q += gridDim.x) {
      int tr = tid - TL * NT;
      asm volatile("" : "+v"(tr));
#pragma unroll
      for (int i = 0; i < LINE_BARRIERS; i++) __syncthreads();
      __syncthreads();
      const long long R0 = 2 * q;
      const bool two = R0 + 1 < nrows;
#pragma unroll 1
      for (int j = tr; j < M; j += 64 * RW) {
        PfRow2 e[6];
#pragma unroll
        for (int k = 0; k < 6; k++) e[k] = lds[k * LPL + j];
        typedef double pf_d2 __attribute__((ext_vector_type(2)));
        {
          const double da[6] = {(double)e[0].a.x, (double)e[1].a.x, (double)e[2].a.x, (double)e[3].a.x, (double)e[4].a.x, (double)e[5].a.x},
                       db[6] = {(double)e[0].a.y, (double)e[1].a.y, (double)e[2].a.y, (double)e[3].a.y, (double)e[4].a.y, (double)e[5].a.y};
          double a1, a2, a3, b1, b2, b3;
          pf_invariants(da, a1, a2, a3);
          pf_invariants(db, b1, b2, b3);
          if (pf_invariants_lose_diagonal(da, a1, a2) || pf_invariants_lose_diagonal(db, b1, b2)) *p.flag = 1.0;
          const long long ao = R0 * p.inv_pitch + 2 * j;
          pf_d2 w1, w2, w3;
          w1.x = a1; w1.y = b1; w2.x = a2; w2.y = b2; w3.x = a3; w3.y = b3;
          __builtin_nontemporal_store(w1, reinterpret_cast<pf_d2 *>(o1 + ao));
          __builtin_nontemporal_store(w2, reinterpret_cast<pf_d2 *>(o2 + ao));
          __builtin_nontemporal_store(w3, reinterpret_cast<pf_d2 *>(o3 + ao));
        }
        if (two) {
          const double da[6] = {(double)e[0].b.x, (double)e[1].b.x, (double)e[2].b.x, (double)e[3].b.x, (double)e[4].b.x, (double)e[5].b.x},
                       db[6] = {(double)e[0].b.y, (double)e[1].b.y, (double)e[2].b.y, (double)e[3].b.y, (double)e[4].b.y, (double)e[5].b.y};
          double a1, a2, a3, b1, b2, b3;
          pf_invariants(da, a1, a2, a3);
          pf_invariants(db, b1, b2, b3);
          if (pf_invariants_lose_diagonal(da, a1, a2) || pf_invariants_lose_diagonal(db, b1, b2)) *p.flag = 1.0;
          const long long ao = (R0 + 1) * p.inv_pitch + 2 * j;
          pf_d2 w1, w2, w3;
          w1.x = a1; w1.y = b1; w2.x = a2; w2.y = b2; w3.x = a3; w3.y = b3;
          __builtin_nontemporal_store(w1, reinterpret_cast<pf_d2 *>(o1 + ao));
          __builtin_nontemporal_store(w2, reinterpret_cast<pf_d2 *>(o2 + ao));
          __builtin_nontemporal_store(w3, reinterpret_cast<pf_d2 *>(o3 + ao));
        }
      }
      __syncthreads();
    }
    return;
  }
  // ---- the transforming waves
  const int lw = tid / NT, tl = tid % NT;
  const int l = (lw + PF_ZI_ROT) % TL;
  const pf_f2 *__restrict__ tw = reinterpret_cast<const pf_f2 *>(p.tw);
  const float kf = (float)(2.0 * 3.14159265358979323846 / (double)N);
  const float norm = (float)p.norm;
  const float dcv = p.dc ? (float)(*p.dc) : 0.f;
  const pf_f2 *in = reinterpret_cast<const pf_f2 *>(p.job[0].in);
  int mul = p.job[0].mul;
#pragma unroll
  for (int j = 1; j < TL; j++)
    if (l == j) { in = reinterpret_cast<const pf_f2 *>(p.job[j].in); mul = p.job[j].mul; }
  if constexpr (NT >= 64) mul = __builtin_amdgcn_readfirstlane(mul);
  auto line_sync = [&]() { if constexpr (WAVE_LOCAL) pf_wave_sync(); else __syncthreads(); };
  const pf_f2 zero = A::mk(0.f, 0.f);
  pf_f2 na[9], nb[9];
  auto fetch = [&](long long q) {
    const long long R0 = 2 * q;
    const bool two = R0 + 1 < nrows;
    const pf_f2 *__restrict__ r0 = in + R0 * p.in_pitch, *__restrict__ r1 = r0 + p.in_pitch;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = tl + m * NT;
      const bool inb = k <= p.band_k;
      na[m] = inb ? __builtin_nontemporal_load(r0 + k) : zero;
      nb[m] = (inb && two) ? __builtin_nontemporal_load(r1 + k) : zero;
    }
    const bool last = tl == 0 && M <= p.band_k;
    na[8] = last ? __builtin_nontemporal_load(r0 + M) : zero;
    nb[8] = (last && two) ? __builtin_nontemporal_load(r1 + M) : zero;
  };
  long long q = blockIdx.x;
  if (q < npairs) fetch(q);
#pragma unroll 1
  for (; q < npairs; q += gridDim.x) {
    int tlj = tl, lj = l;
    asm volatile("" : "+v"(tlj), "+v"(lj));  // keep the index math inside the loop (see k_strided)
    PfRow2 *L = lds + lj * LPL;
    // phase A: the rows of this component -> its LDS line
#pragma unroll
    for (int m = 0; m < 8; m++) { PfRow2 e; e.a = na[m]; e.b = nb[m]; L[tlj + m * NT] = e; }
    if (tlj == 0) { PfRow2 e; e.a = na[8]; e.b = nb[8]; L[M] = e; }
    line_sync();
    // phase B: kz factor + Hermitian fold into the half-length complex lines (pf_zfold / pf_c2r_pre in the packed algebra)
    pf_f2 a[8], b[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tlj + m * NT;
      const PfRow2 mk = L[M - e];
      pf_f2 xa = na[m], xb = nb[m], ya = mk.a, yb = mk.b;
      if (mul != 0 /* PF_MUL_ONE */) {
        float fk = kf * (float)e, fm = kf * (float)(M - e);
        if (mul == 2 /* PF_MUL_K2 */) { fk *= fk; fm *= fm; }
        xa = A::scale(xa, fk); xb = A::scale(xb, fk); ya = A::scale(ya, fm); yb = A::scale(yb, fm);
      }
      if (m == 0 && e == 0) {   // k = 0: only the real parts of X[0] and X[M]
        a[m] = A::mk(xa.x + ya.x, xa.x - ya.x);
        b[m] = A::mk(xb.x + yb.x, xb.x - yb.x);
      } else {
        const pf_f2 wk = tw[e];
        a[m] = A::addi<+1>(A::addc(xa, ya), A::cmul<+1>(A::subc(xa, ya), wk));
        b[m] = A::addi<+1>(A::addc(xb, yb), A::cmul<+1>(A::subc(xb, yb), wk));
      }
    }
    line_sync();
    PfPk2Stages<M, 0, WAVE_LOCAL>::run(
        a, b, tlj, tw, [&](int pos, pf_f2 va, pf_f2 vb) { PfRow2 e; e.a = va; e.b = vb; L[pos + (pos >> 4)] = e; },
        [&](int pos, pf_f2 &va, pf_f2 &vb) { const PfRow2 e = L[pos + (pos >> 4)]; va = e.a; vb = e.b; });
    line_sync();  // every thread of the line is done with the exchange area
    // the real rows of this component, in order, into its line (complex j = reals 2 j, 2 j + 1)
#pragma unroll
    for (int m = 0; m < 8; m++) { PfRow2 e; e.a = A::fma1(a[m], norm, dcv); e.b = A::fma1(b[m], norm, dcv); L[tlj + m * NT] = e; }
    __syncthreads();
    // handed over to the reducing waves: the next pair of rows travels meanwhile (past the last pair: zeros, not "nothing" -- a
    // conditional request would keep the old rows alive through the stages)
    if (q + gridDim.x < npairs) fetch(q + gridDim.x);
    else {
#pragma unroll
      for (int m = 0; m < 9; m++) { na[m] = zero; nb[m] = zero; }
    }
    __syncthreads();  // the lines are rewritten by the next phase A
  }
}

template <typename F, int N, int TL>
__global__ void __launch_bounds__(TL *(N / 16)) k_r2c(const PfR2CParams p) {
  using C = pfc<F>;
  constexpr int M = N / 2, NT = M / 8;
  constexpr int LPL = M + M / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tid = threadIdx.x;
  const int l = tid / NT, tl = tid % NT;
  const long long row = (long long)blockIdx.x * TL + l;
  const bool valid = row < p.nlines;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  C *L = lds + l * LPL;

  C v[8];
  {
    const C *in = reinterpret_cast<const C *>(reinterpret_cast<const F *>(p.in) + row * p.in_pitch);
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = valid ? pf_ld_stream(&in[tl + m * NT]) : pf_mk<F>(0, 0);
  }
  PfStages<F, M, -1, 2>::run(
      v, tl, tw, [&](int pos, C val) { L[pf_lpad(pos)] = val; }, [&](int pos) { return L[pf_lpad(pos)]; });
#pragma unroll
  for (int m = 0; m < 8; m++) L[pf_lpad(tl + m * NT)] = v[m];
  __syncthreads();
  if (valid) {
    C *out = reinterpret_cast<C *>(p.out) + row * p.out_pitch;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = tl + m * NT;
      const C zk = v[m], zmk = L[pf_lpad((M - k) & (M - 1))];
      pf_st_stream(&out[k], pf_r2c_post<F>(zk, zmk, tw[k]));
      if (k == 0) out[M] = pf_mk<F>(zk.x - zk.y, (F)0);
    }
  }
}

// etab[e] = exp(-k_e^2 rs^2 / 2), k_e = 2 pi s(e) / n, s(e) the signed wavenumber of index e
__global__ void k_exp_table(double *etab, int n, double rs) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const double ke = 2.0 * 3.14159265358979323846 / (double)n * (e > n / 2 ? e - n : e);
  etab[e] = exp(-0.5 * ke * ke * rs * rs);
}
int pf_launch_exp_table(double *etab, int n, double rs, hipStream_t st) {
  hipLaunchKernelGGL(k_exp_table, dim3((n + 255) / 256), dim3(256), 0, st, etab, n, rs);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------ dispatch ----

// columns per workgroup of the strided passes: 128-byte row segments where LDS allows
template <typename F, int N> struct PfTileCols {
// LDS of one tile.  128 KB: eight fp64 columns of a 1024-point line -- every row segment a tile reads or writes is a whole
// 128-byte line -- in one 1024-thread workgroup per CU (the same sixteen waves as two 64 KB workgroups).  Against 64 KB
// (four columns, 64-byte segments) at 1024^3, four interleaved runs each on one box: x-pass 6.41 -> 5.69 ms, y-pass
// 12.21 -> 11.31 ms per Hessian launch, 860 (854..863) -> 832 (831..833) ms per step, and the process-to-process spread of
// the half-line layout is gone.  (Round 1 saw no gain from it with the x-major intermediate and without the tile prefetch.)
#ifndef PF_TILE_LDS_KB
#define PF_TILE_LDS_KB 128
#endif
  static constexpr int lds_budget = PF_TILE_LDS_KB * 1024;
  static constexpr int t0 = 128 / (2 * (int)sizeof(F));  // 8 (fp64) or 16 (fp32)
  static constexpr int fit = lds_budget / (N * 2 * (int)sizeof(F));
  static constexpr int thr = 8192 / N;  // T*N/8 <= 1024
  static constexpr int a = t0 < fit ? t0 : fit;
  static constexpr int b = a < thr ? a : thr;
  static constexpr int value = b < 1 ? 1 : b;
};

template <typename F, int N, int DIR, bool FA>
static int launch_strided_v(const PfStridedParams &p, hipStream_t st, int ntiles, long long nwork) {
  constexpr int T = PfTileCols<F, N>::value;
  dim3 grid((unsigned)(((nwork + 7) >> 3) << 3), 1, 1), block(T * N / 8, 1, 1);
  const size_t shm = (size_t)N * T * sizeof(pfc<F>);
  if (shm > 64 * 1024) {
    // LDS beyond 64 KB is opt-in per kernel function and device.  The flag is set only once the runtime has accepted the
    // size (a refusal is reported: code 3 = "dynamic LDS opt-in refused", not a bare launch failure later on) and is atomic:
    // several host threads launch the same instantiation (virtual ranks, tests).
    static std::atomic<bool> raised[PF_MAX_DEVICES];  // per instantiation and device
    const int d = p.dev >= 0 && p.dev < PF_MAX_DEVICES ? p.dev : 0;
    if (!raised[d].load(std::memory_order_acquire)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_strided<F, N, T, DIR, FA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
        fprintf(stderr, "ERROR on task 0: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) refused for the %d-point strided pass on device %d\n", shm, N, d);
        return 3;
      }
      raised[d].store(true, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((k_strided<F, N, T, DIR, FA>), grid, block, shm, st, p, nwork, ntiles);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <typename F, int N, int DIR>
static int launch_strided_n(const PfStridedParams &p, hipStream_t st) {
  constexpr int T = PfTileCols<F, N>::value;
  constexpr int NL = pf_lane<F>::n;  // columns per thread
  constexpr int NT = N / 8;
  using S = typename pf_lane<F>::type;
  const int ntiles = (p.ncols + NL * T - 1) / (NL * T);
  const long long nwork = (long long)ntiles * p.nouter;
  // split addresses (pf_addr_uniform): a slab is a whole number of NT elements on both sides (up to eight ranks), and the
  // per-lane part -- at most (NT - 1) rows of the transformed axis plus a tile's columns -- fits 32 bits in bytes
  auto lane_fits = [&](const PfAddr &a) {
    return (1 << a.el_shift) >= NT && a.els > 0 && ((unsigned long long)(NT - 1) * (unsigned long long)a.els + (unsigned long long)(NL * T)) * sizeof(pfc<S>) < (1ull << 32);
  };
  if (lane_fits(p.ain) && lane_fits(p.aout)) return launch_strided_v<F, N, DIR, true>(p, st, ntiles, nwork);
  return launch_strided_v<F, N, DIR, false>(p, st, ntiles, nwork);
}
// fp32 lines of 1024 points and more: two columns per thread (the thread budget, 8 N / 8 <= 1024, would otherwise leave 64-byte
// row segments at 1024 points and 32-byte ones at 2048); PF_F32_PAIRS=0 builds the one-column kernels for them too
#ifndef PF_F32_PAIRS
#define PF_F32_PAIRS 1
#endif
template <int N, int DIR>
static int launch_strided_f32(const PfStridedParams &p, hipStream_t st) {
  if constexpr (N == 2048 || N == 1024) {  // the kernels in packed (re, im) arithmetic (pf_fft16_kernels.hip), where their addressing applies
    const int rc = pf_launch_strided16(4, N, DIR, p, st);
    if (rc >= 0) return rc;
  }
  if constexpr (PF_F32_PAIRS && N >= 1024) return launch_strided_n<pf_f32x2, N, DIR>(p, st);
  else return launch_strided_n<float, N, DIR>(p, st);
}

template <typename F, int N>
static int launch_c2r_n(const PfC2RParams &p, hipStream_t st) {
  constexpr int M = N / 2, NT = M / 8;
  constexpr int TL = (NT >= 256) ? 1 : 256 / NT;
  constexpr int LPL = (M + 1 > M + M / 8) ? M + 1 : M + M / 8;
  const long long nblk = (p.nlines + TL - 1) / TL;
  dim3 grid((unsigned)(nblk * p.njobs), 1, 1), block(TL * NT, 1, 1);
  const size_t shm = (size_t)TL * LPL * sizeof(pfc<F>);
  // Workgroups that walk over tiles with the next tile prefetched in registers: 21.5 -> 18.4 ms per six-field launch at
  // 1024^3 (4.8 -> 5.6 TB/s algorithmic).  3 fit per CU (140 VGPRs); 24 per CU = 8 rounds of ~256 tiles each evens out
  // the tail (measured 3: 19.1, 6: 18.9, 12: 18.6, 24: 18.4 ms).  PF_ZPASS_PERSIST=0 selects the one-shot kernel.
  const int persist = p.persist_per_cu;
  if (persist > 0 && N >= 64 && p.ncu > 0) {
    const long long ntiles = nblk * p.njobs;
    long long g = (long long)p.ncu * persist;
    if (g > ntiles) g = ntiles;
    hipLaunchKernelGGL((k_c2r_persistent<F, N, TL>), dim3((unsigned)g), block, shm, st, p, ntiles);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  hipLaunchKernelGGL((k_c2r<F, N, TL>), grid, block, shm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <typename F, int N>
static int launch_c2r_invariants_n(const PfC2RParams &p, hipStream_t st, int mode) {
  constexpr int M = N / 2, NT = M / 8;
  constexpr int LPL = PfZiPlan<F, M>::line;
  // (a second row of prefetch per workgroup -- 159 VGPRs, still two workgroups per CU -- made the pass slower: 222-227 ms
  // against 190-194 ms per step; more rows in flight on six fields at once cost more in DRAM locality than they hide)
  // two workgroups fit per CU; 32 per CU in the grid evens out the tail (measured 6: 254, 8: 248, 16: 241, 32: 235-237,
  // 64: 236 ms per step of eleven launches at 1024^3)
  long long g = (long long)(p.ncu > 0 ? p.ncu : 256) * (p.inv_per_cu > 0 ? p.inv_per_cu : 32);
  if (g > p.nlines) g = p.nlines;
  const size_t shm = (size_t)6 * LPL * sizeof(pfc<F>);
  if (mode == 1) {
    if constexpr (PfZiPlan<F, M>::spec(1)) hipLaunchKernelGGL((k_c2r_invariants_spec<F, N, 1>), dim3((unsigned)g), dim3(6 * NT + 64 * PfZiPlan<F, M>::reducer_waves), shm, st, p, p.nlines);
    else hipLaunchKernelGGL((k_c2r_invariants<F, N, 1>), dim3((unsigned)g), dim3(6 * NT), shm, st, p, p.nlines);
  } else if constexpr (PF_ZI_PK2 && sizeof(F) == 4 && N >= 512 && N <= PF_ZI_PK2_MAXN) {
    // fp32 rows of 512 and 1024 points: two rows per thread (k_c2r_invariants_pk2); a job with the factor i k has no place in the sweep's z-pass
    for (int j = 0; j < 6; j++) if (p.job[j].mul == PF_MUL_IK) return 2;
    constexpr int RW = N >= 2048 ? 4 : (N >= 1024 ? 2 : 1);
    constexpr int LPL2 = M + M / 16 + 1;
    const size_t shm2 = (size_t)6 * LPL2 * sizeof(PfRow2);
    const int threads = 6 * NT + 64 * RW;
    static std::atomic<bool> raised2[PF_MAX_DEVICES];
    if (shm2 > 64 * 1024 && p.dev >= 0 && p.dev < PF_MAX_DEVICES && !raised2[p.dev].load(std::memory_order_acquire)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_c2r_invariants_pk2<N, RW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) != hipSuccess) return 3;
      raised2[p.dev].store(true, std::memory_order_release);
    }
    long long g2 = g;
    const long long npairs = (p.nlines + 1) / 2;
    if (g2 > npairs) g2 = npairs;
    hipLaunchKernelGGL((k_c2r_invariants_pk2<N, RW>), dim3((unsigned)g2), dim3(threads), shm2, st, p, p.nlines);
  } else if constexpr (PfZiPlan<F, M>::spec(0)) hipLaunchKernelGGL((k_c2r_invariants_spec<F, N, 0>), dim3((unsigned)g), dim3(6 * NT + 64 * PfZiPlan<F, M>::reducer_waves), shm, st, p, p.nlines);
  else hipLaunchKernelGGL((k_c2r_invariants<F, N, 0>), dim3((unsigned)g), dim3(6 * NT), shm, st, p, p.nlines);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <typename F, int N>
static int launch_r2c_n(const PfR2CParams &p, hipStream_t st) {
  constexpr int M = N / 2, NT = M / 8;
  constexpr int TL = (NT >= 256) ? 1 : 256 / NT;
  constexpr int LPL = M + M / 8;
  const long long nblk = (p.nlines + TL - 1) / TL;
  dim3 grid((unsigned)nblk, 1, 1), block(TL * NT, 1, 1);
  const size_t shm = (size_t)TL * LPL * sizeof(pfc<F>);
  hipLaunchKernelGGL((k_r2c<F, N, TL>), grid, block, shm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

#define PF_SWITCH_N(n, CALL)                  \
  switch (n) {                                \
    case 16: return CALL(16);                 \
    case 32: return CALL(32);                 \
    case 64: return CALL(64);                 \
    case 128: return CALL(128);               \
    case 256: return CALL(256);               \
    case 512: return CALL(512);               \
    case 1024: return CALL(1024);             \
    case 2048: return CALL(2048);             \
    default: return 2;                        \
  }

int pf_launch_strided(int fb, int n, int dir, const PfStridedParams &p, hipStream_t st) {
  if (n & (n - 1)) return pf_launch_mixed_strided(fb, n, dir, p, st);  // not a power of two: run-time stage plan
  if (fb == 8) {
    if (dir > 0) {
#define CALL(NN) launch_strided_n<double, NN, +1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    } else {
#define CALL(NN) launch_strided_n<double, NN, -1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    }
  } else {
    if (dir > 0) {
#define CALL(NN) launch_strided_f32<NN, +1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    } else {
#define CALL(NN) launch_strided_f32<NN, -1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    }
  }
}

int pf_launch_c2r(int fb, int n, const PfC2RParams &p, hipStream_t st) {
  if (n & (n - 1)) return pf_launch_mixed_c2r(fb, n, p, st);
  if (fb == 8) {
#define CALL(NN) launch_c2r_n<double, NN>(p, st)
    PF_SWITCH_N(n, CALL)
#undef CALL
  } else {
#define CALL(NN) launch_c2r_n<float, NN>(p, st)
    PF_SWITCH_N(n, CALL)
#undef CALL
  }
}

bool pf_c2r_invariants_supported(int fb, int n) {
  // (the six lines of a 2048-point fp64 row would need 110 KB of LDS; fp32 fields go up to 2048)
  if (n & (n - 1)) return pf_mixed_invariants_supported(fb, n);
  return n >= 16 && n <= (fb == 8 ? 1024 : 2048);
}
// ... and the sweep takes it: always where the line length is a power of two; for the other sizes with fp64 fields only -- three fp64
// invariants are as many bytes as six fp32 components, and the six-component z-pass of those sizes is the faster kernel
// (768^3, fp32 fields: 273 ms per step with six components, 288 with invariants; profiles/r05_notes.md)
// -- and with a stage plan compiled in: with a run-time plan the six lines of a row in one workgroup are the slower way again
// (768^3, fp64, run-time plans: 23.2 ms per launch against 11.5 + 8.0 for six components and their reduction in the solve).
bool pf_c2r_invariants_preferred(int fb, int n) {
  return pf_c2r_invariants_supported(fb, n) && (!(n & (n - 1)) || (fb == 8 && pf_mixed_plan_compiled_in(n)));
}
int pf_launch_c2r_invariants(int fb, int n, const PfC2RParams &p, hipStream_t st, int mode) {
  if (p.njobs != 6 || (mode == 1 && !p.acc) || (mode == 0 && !p.flag) || !pf_c2r_invariants_supported(fb, n)) return 2;
  if (fb == 4 && mode == 0 && (!p.inv_out[0] || !p.inv_out[1] || !p.inv_out[2])) return 2;
  if (n & (n - 1)) return pf_launch_mixed_c2r_invariants(fb, n, p, st, mode);
  if (fb == 8) {
#define CALL(NN) launch_c2r_invariants_n<double, NN>(p, st, mode)
    PF_SWITCH_N(n, CALL)
#undef CALL
  } else {
    if (mode == 0 && (!p.inv_out[0] || !p.inv_out[1] || !p.inv_out[2])) return 2;
#define CALL(NN) launch_c2r_invariants_n<float, NN>(p, st, mode)
    PF_SWITCH_N(n, CALL)
#undef CALL
  }
}

int pf_launch_r2c(int fb, int n, const PfR2CParams &p, hipStream_t st) {
  if (n & (n - 1)) return pf_launch_mixed_r2c(fb, n, p, st);
  if (fb == 8) {
#define CALL(NN) launch_r2c_n<double, NN>(p, st)
    PF_SWITCH_N(n, CALL)
#undef CALL
  } else {
#define CALL(NN) launch_r2c_n<float, NN>(p, st)
    PF_SWITCH_N(n, CALL)
#undef CALL
  }
}
