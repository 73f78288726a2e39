// pf_mixed_kernels.hip -- the three 1-D passes for grid sizes that are NOT a power of two (gfx950).
//
// The reference takes any GridSize through FFTW / PFFT (plans at src/fmax-pfft.c:139-188; the example parameter file uses
// 200).  The passes of pf_fft_kernels.hip are specialised on the line length (stage plans of 8 points per thread) and cover
// N = 2^k; here a line of N = R0 * R1 * ... points with radices from {8, 5, 4, 3, 2}, R0 = 8 (strided passes; N a multiple of 8) or
// 8 / 4 (half-length lines of the z-pass), Stockham auto-sort as in pf_fft_core.h:
//   stage s, butterfly b of N / R_s:  inputs line[b + q N / R_s], q < R_s, times w^(q k), k = b mod NS_s,
//                                     w = exp(+-2 pi i / (NS_s R_s)), NS_s = R_0 ... R_{s-1};  DFT_{R_s};
//                                     outputs to (b - k) R_s + k + t NS_s, t < R_s.
// The plan is a RUN-TIME table (PfMixedPlan: any such N up to 2048) or, for the grid sizes of PF_MIXED_CT_SIZES, built into the
// kernel (PfPlanCT: every divisor, stride and trip count a constant, 79 instead of 128 registers in the strided pass, room to hold a
// tile over its jobs; 768^3: 587 -> 452 ms per step by that alone).
// A line has N / R0 threads; stage 0 is one butterfly per thread on the points it loaded from HBM, later stages deal their N / R_s
// butterflies over the same threads (ceil(R0 / R_s) each) and exchange through LDS; the last stage stores straight to HBM.  Same
// semantics, same parameter blocks and the same layouts as the power-of-two kernels -- filter on load, k multipliers, band limits,
// fp32 product rows, slabs of a multi-rank box -- so that the sweep keeps its shared passes (x 1 -> 3, y 3 -> 6, z 6 -> 3 invariants
// or 6 components) instead of one library transform per component.
// What these kernels do not have: the paired radix-16 stage, two columns per thread, waves that only reduce in the invariant z-pass.
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "pf_internal.h"
#include "pf_fft_core.h"
#include "pf_fft_stages.h"
#include "pf_collapse_core.h"  // pf_invariants, pf_lpt3b_accumulate

#ifndef PF_MIXED_TW_POWERS
#define PF_MIXED_TW_POWERS 1
#endif
#ifndef PF_MIXED_ZI_ONE_ROW
#define PF_MIXED_ZI_ONE_ROW 1
#endif
#ifndef PF_MIXED_NEXT_INPUT
#define PF_MIXED_NEXT_INPUT 1
#endif
#ifndef PF_MIXED_KEEP_RT
#define PF_MIXED_KEEP_RT 1
#endif
#ifndef PF_MIXED_KEEP
#define PF_MIXED_KEEP 1
#endif
// hardware deals consecutive workgroups round-robin over the 8 XCDs: a contiguous range of tiles per XCD (as pf_fft_kernels.hip)
__device__ __forceinline__ long long pf_xcd_swizzle_mixed(long long b, long long per_xcd) { return (b & 7) * per_xcd + (b >> 3); }

// ---- small DFTs in registers: X_k = sum_t u_t w^(k t), w = exp(DIR 2 pi i / R), natural order in and out ----
template <int DIR, typename F> __device__ __forceinline__ void pf_bfly3(pfc<F> &a, pfc<F> &b, pfc<F> &c) {
  const F s = (F)0.86602540378443864676;  // sin(2 pi / 3)
  const pfc<F> t1 = b + c, d = b - c;
  const pfc<F> t2 = pf_mk<F>(a.x - (F)0.5 * t1.x, a.y - (F)0.5 * t1.y);
  const pfc<F> t3 = pf_mul_i<DIR>(pf_scale(d, s));
  a = a + t1;
  b = t2 + t3;
  c = t2 - t3;
}
template <int DIR, typename F> __device__ __forceinline__ void pf_bfly5(pfc<F> &x0, pfc<F> &x1, pfc<F> &x2, pfc<F> &x3, pfc<F> &x4) {
  const F c1 = (F)0.30901699437494742410, c2 = (F)-0.80901699437494742410;  // cos 72, cos 144
  const F s1 = (F)0.95105651629515357212, s2 = (F)0.58778525229247312917;   // sin 72, sin 144
  const pfc<F> t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
  const pfc<F> m1 = pf_mk<F>(x0.x + c1 * t1.x + c2 * t2.x, x0.y + c1 * t1.y + c2 * t2.y);
  const pfc<F> m2 = pf_mk<F>(x0.x + c2 * t1.x + c1 * t2.x, x0.y + c2 * t1.y + c1 * t2.y);
  const pfc<F> n1 = pf_mul_i<DIR>(pf_mk<F>(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
  const pfc<F> n2 = pf_mul_i<DIR>(pf_mk<F>(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
  x0 = x0 + t1 + t2;
  x1 = m1 + n1; x4 = m1 - n1;
  x2 = m2 + n2; x3 = m2 - n2;
}
template <int R, int DIR, typename F> __device__ __forceinline__ void pf_dft_small(pfc<F> (&u)[R]) {
  if constexpr (R == 8) pf_bfly8<DIR>(u);
  else if constexpr (R == 5) pf_bfly5<DIR>(u[0], u[1], u[2], u[3], u[4]);
  else if constexpr (R == 4) pf_bfly4<DIR>(u[0], u[1], u[2], u[3]);
  else if constexpr (R == 3) pf_bfly3<DIR>(u[0], u[1], u[2]);
  else pf_bfly2<DIR>(u[0], u[1]);
}

// One stage s >= 1 of a line through LDS.  rd(pos) / wr(pos, val): the line's exchange area; sync(): every thread of the line
// is past its reads (writes).  LAST: the results go to out(pos, val) instead (HBM, or the post-processing of the caller).
// tw[j tws] = exp(+2 pi i j / n) (tws = 2: the half-length lines of the z-pass use the table of the full length).  Every thread
// of the line must call this (barriers inside).
template <int R, int R0, int DIR, bool LAST, typename F, typename RD, typename WR, typename SYNC, typename OUT>
__device__ __forceinline__ void pf_mixed_stage(int n, int ns, unsigned magic, int nt, int tl, const pfc<F> *__restrict__ tw, int tws, RD rd, WR wr, SYNC sync, OUT out) {
  constexpr int LOOPS = (R0 + R - 1) / R;  // butterflies per thread: ceil((n / R) / (n / R0))
  const int nb = n / R;                    // butterflies of the stage = distance of a butterfly's inputs (uniform: scalar unit)
  const int twm = n / (ns * R);
  pfc<F> u[LOOPS][R], w1[LOOPS];
  // one table value per butterfly, w = exp(+-2 pi i k / (NS R)), its powers by multiplication below (depth three for w^7): the lanes of
  // a wave ask for 64 different table entries per request, and R - 1 such requests per butterfly kept the L1 busier than the rows
  // themselves (without them the z-passes ran 25 % faster: profiles/r05_notes.md).  Asked for in front of the exchange: in flight
  // while the line's threads meet.  (PF_MIXED_TW_POWERS=0, A/B: every power from the table.)
#pragma unroll
  for (int i = 0; i < LOOPS; i++) {
    const int b = tl + i * nt;
    if (b < nb) {
      const int k = b - ns * (int)__umulhi((unsigned)b, magic);  // b mod ns (magic = ceil(2^32 / ns), exact for b < 2^11)
      if (PF_MIXED_TW_POWERS) w1[i] = tw[k * twm * tws];
#pragma unroll
      for (int q = 0; q < R; q++) u[i][q] = rd(b + q * nb);
    }
  }
  sync();
#pragma unroll
  for (int i = 0; i < LOOPS; i++) {
    const int b = tl + i * nt;
    if (b < nb) {
      const int k = b - ns * (int)__umulhi((unsigned)b, magic);
      if (PF_MIXED_TW_POWERS) {
        pfc<F> w[R];
        w[1] = w1[i];
        if (DIR < 0) w[1].y = -w[1].y;
        if constexpr (R > 2) w[2] = pf_cmul(w[1], w[1]);
        if constexpr (R > 3) w[3] = pf_cmul(w[2], w[1]);
        if constexpr (R > 4) w[4] = pf_cmul(w[2], w[2]);
        if constexpr (R > 5) { w[5] = pf_cmul(w[4], w[1]); w[6] = pf_cmul(w[3], w[3]); w[7] = pf_cmul(w[4], w[3]); }
#pragma unroll
        for (int q = 1; q < R; q++) u[i][q] = pf_cmul(u[i][q], w[q]);
      } else {
        const int step = k * twm * tws;  // k twm < n / R: the index q k twm stays below n, no reduction needed
        int idx = 0;
#pragma unroll
        for (int q = 1; q < R; q++) {
          idx += step;
          pfc<F> w = tw[idx];
          if (DIR < 0) w.y = -w.y;
          u[i][q] = pf_cmul(u[i][q], w);
        }
      }
      pf_dft_small<R, DIR>(u[i]);
      const int base = (b - k) * R + k;
#pragma unroll
      for (int t = 0; t < R; t++) {
        if (LAST) out(base + t * ns, u[i][t]);
        else wr(base + t * ns, u[i][t]);
      }
    }
  }
  if (!LAST) sync();
}

// stages 1 .. nstages-1 of the plan (stage 0 done by the caller, its outputs already written through wr and synced)
template <int R0, int DIR, typename F, typename RD, typename WR, typename SYNC, typename OUT>
__device__ __forceinline__ void pf_mixed_tail(const PfMixedPlan &pl, int tl, const pfc<F> *__restrict__ tw, int tws, RD rd, WR wr, SYNC sync, OUT out) {
  const int nt = pl.n / R0;
  int ns = R0;
#pragma unroll 1
  for (int s = 1; s < pl.nstages; s++) {
    const int R = pl.radix[s];
    const unsigned magic = pl.magic[s];
    const bool last = s + 1 == pl.nstages;
    // (an opaque copy of the thread index per stage: left to itself the compiler hoists the index arithmetic of every radix
    //  case out of this loop and spills what it hoisted -- the same cure as in k_strided)
    int tls = tl;
    asm volatile("" : "+v"(tls));
#define PF_MIXED_CASE(RR)                                                                                              \
  case RR:                                                                                                             \
    if (last) pf_mixed_stage<RR, R0, DIR, true>(pl.n, ns, magic, nt, tls, tw, tws, rd, wr, sync, out);                  \
    else pf_mixed_stage<RR, R0, DIR, false>(pl.n, ns, magic, nt, tls, tw, tws, rd, wr, sync, out);                      \
    break;
    switch (R) {
      PF_MIXED_CASE(8)
      PF_MIXED_CASE(5)
      PF_MIXED_CASE(4)
      PF_MIXED_CASE(3)
      default:
        if (last) pf_mixed_stage<2, R0, DIR, true>(pl.n, ns, magic, nt, tls, tw, tws, rd, wr, sync, out);
        else pf_mixed_stage<2, R0, DIR, false>(pl.n, ns, magic, nt, tls, tw, tws, rd, wr, sync, out);
        break;
    }
#undef PF_MIXED_CASE
    ns *= R;
  }
}

// A plan known at compile time: the same stages with every divisor, stride and trip count a constant (no switch over the radix, no
// multiply-high, no predicate where the butterflies divide evenly) -- the sizes the launchers know by name (PF_MIXED_CT_SIZES).
// PfPlanRT: the run-time plan of the argument block.
template <int... R> struct PfPlanCT {
  static constexpr int nstages = (int)sizeof...(R);
  static constexpr int radix[sizeof...(R)] = {R...};
  static constexpr int n = (R * ...);
  static constexpr int ns(int s) { int v = 1; for (int i = 0; i < s; i++) v *= radix[i]; return v; }
};
struct PfPlanRT { static constexpr int nstages = 0, n = 0; };
template <typename PLAN, int S, int R0, int DIR, typename F, typename RD, typename WR, typename SYNC, typename OUT>
__device__ __forceinline__ void pf_mixed_tail_ct(int tl, const pfc<F> *__restrict__ tw, int tws, RD rd, WR wr, SYNC sync, OUT out) {
  if constexpr (S < PLAN::nstages) {
    constexpr int R = PLAN::radix[S], NS = PLAN::ns(S);
    constexpr unsigned magic = (unsigned)(((1ull << 32) + NS - 1) / NS);
    pf_mixed_stage<R, R0, DIR, S + 1 == PLAN::nstages>(PLAN::n, NS, magic, PLAN::n / R0, tl, tw, tws, rd, wr, sync, out);
    pf_mixed_tail_ct<PLAN, S + 1, R0, DIR, F>(tl, tw, tws, rd, wr, sync, out);
  }
}
template <typename PLAN, int R0, int DIR, typename F, typename RD, typename WR, typename SYNC, typename OUT>
__device__ __forceinline__ void pf_mixed_tail_any(const PfMixedPlan &pl, int tl, const pfc<F> *__restrict__ tw, int tws, RD rd, WR wr, SYNC sync, OUT out) {
  if constexpr (PLAN::nstages > 0) {
    static_assert(PLAN::radix[0] == R0, "first radix of the plan");
    pf_mixed_tail_ct<PLAN, 1, R0, DIR, F>(tl, tw, tws, rd, wr, sync, out);
  } else pf_mixed_tail<R0, DIR, F>(pl, tl, tw, tws, rd, wr, sync, out);
}

// ------------------------------------------------------------------------------------------------ strided passes ----
// x- or y-pass: a workgroup owns T adjacent columns of one line of tiles and all n points along the transformed axis
// (n / 8 threads per column).  Parameters and semantics: PfStridedParams, as k_strided.
// Threads: (c, tl) = (threadIdx.x, threadIdx.y), T = blockDim.x columns.  A line's element e sits in slab e / el_len at row
// e % el_len (PfAddr: the blocks of a multi-rank layout; one rank: e * els) -- the quotient by multiply-high, any slab length.
// The tile is loaded for every job (jobs on the same input find it in L2): the eight points of a thread then live only
// through one job, which is what lets the run-time plan fit the register file.
// Plans known at compile time (KEEP): the loaded and filtered points are held while the jobs on the same input follow each other.
// A thread asks for its eight points in one go and filters them when they are there (asked for and filtered one by one the y-pass
// 3 -> 6 of a 768^3 box took 6.1 instead of 5.2 ms).  Not here, measured (profiles/r05_notes.md): workgroups that walk over tiles
// and request the next input while the jobs on the current one run -- 6.0 ms: the wait for those loads is a wait for every store
// issued after them as well (one counter), where a workgroup that ends leaves its stores behind and the next one starts loading.
// SMALL (run-time plans): the launch has at most 768 threads per workgroup -- 170 registers instead of 128, room to keep a tile over
// its jobs as the built-in plans do (PF_MIXED_KEEP_RT).
template <typename F, int DIR, typename PLAN = PfPlanRT, bool SMALL = false>
__global__ void __launch_bounds__(SMALL ? 768 : 1024) k_mixed_strided(const PfStridedParams p, const PfMixedPlan pl, const long long nwork, const int ntiles) {
  using C = pfc<F>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);  // [n][T]
  const int n = PLAN::n ? PLAN::n : pl.n, nt = n / 8, T = blockDim.x;
  const int nstages = PLAN::n ? PLAN::nstages : pl.nstages;
  const int c = threadIdx.x, tl = threadIdx.y;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  const double kf = 2.0 * 3.14159265358979323846 / (double)n;
  const int half = n / 2;
  // element e of a line: slab e / el_len, row e % el_len of it (PfAddr) -- one rank, or a layout whose slabs follow each other
  // at their own length: e * els.  The quotient by multiply-high (exact for e < 2^16 with the rounded-up reciprocal).
  const unsigned len_in = (unsigned)p.ain.el_len, len_out = (unsigned)p.aout.el_len;
  const unsigned mg_in = pl.magic_in, mg_out = pl.magic_out;
  auto off_in = [&](unsigned e) -> size_t {
    if (len_in >= (unsigned)n) return (size_t)e * (size_t)p.ain.els;
    const unsigned q = __umulhi(e, mg_in), s = e - q * len_in;
    return (size_t)q * (size_t)p.ain.ehs + (size_t)s * (size_t)p.ain.els;
  };
  auto off_out = [&](unsigned e) -> size_t {
    if (len_out >= (unsigned)n) return (size_t)e * (size_t)p.aout.els;
    const unsigned q = __umulhi(e, mg_out), s = e - q * len_out;
    return (size_t)q * (size_t)p.aout.ehs + (size_t)s * (size_t)p.aout.els;
  };
  const long long w = pf_xcd_swizzle_mixed(blockIdx.x, (nwork + 7) >> 3);  // XCDs take contiguous ranges of tiles
  if (w >= nwork) return;
  const int tile = (int)(w % ntiles), outer = (int)(w / ntiles);
  int so = outer + p.outer_offset;
  if (so > half) so -= n;
  if (p.band_outer < half && (so > p.band_outer || so < -p.band_outer)) return;  // (uniform: the whole workgroup leaves)
  const int col = tile * T + c;
  const bool valid = col < p.ncols;
  // the eight points of a thread as they lie in memory (zeros beyond the band limit of the transformed axis); tlx: the thread's index
  // along the line, an opaque copy where the caller sits in a loop (left to itself the compiler hoists the eight offsets out of it)
  auto load_raw = [&](const C *__restrict__ field, C *dst, int tlx) {
    const C *__restrict__ in = field + ((long long)outer * p.ain.os + col);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tlx + m * nt;
      const int se = e > half ? e - n : e;
      const bool inband = se <= p.band_e && se >= -p.band_e;
      dst[m] = (valid && inband) ? pf_ld_stream(in + off_in((unsigned)e)) : pf_zero<F>();
    }
  };
  auto sync = [&]() { __syncthreads(); };
  constexpr bool KEEP = (PLAN::n != 0 || SMALL) && PF_MIXED_KEEP;
  C src[KEEP ? 8 : 1];
  {
    double ko2kc2 = 0.0, woc = 1.0;
    if (p.pre) {
      const double ko = kf * so, kc = kf * col;
      ko2kc2 = ko * ko + kc * kc;
      woc = (p.rs != 0.0 ? exp(-0.5 * ko2kc2 * p.rs * p.rs) : 1.0) * p.growth;
    }
    const void *held = nullptr;
    bool ahead = false;  // src holds the untreated points of the next job's input
#pragma unroll 1
    for (int j = 0; j < p.njobs; j++) {
      const int mul = p.job[j].mul;
      int tlj = tl;
      asm volatile("" : "+v"(tlj));  // (keeps the per-element index and filter arithmetic inside the job loop: hoisted, it spills)
      C v[8];
      const bool fresh = !KEEP || p.job[j].in != held;  // (uniform)
      held = p.job[j].in;
      if (fresh) {
        C raw[8];
        if (KEEP && PF_MIXED_NEXT_INPUT && ahead) {
#pragma unroll
          for (int m = 0; m < 8; m++) raw[m] = src[m];  // (requested by the job before, below)
        } else load_raw(reinterpret_cast<const C *>(p.job[j].in), raw, tlj);
#pragma unroll
        for (int m = 0; m < 8; m++) {
          C x = raw[m];
          if (p.pre) {  // exp(-k^2 rs^2 / 2) g / k^2, zero at k = 0 (src/fmax-pfft.c:366-384)
            const int e = tlj + m * nt;
            const int se = e > half ? e - n : e;
            const double ke = kf * se;
            const double k2 = ke * ke + ko2kc2;
            const double we = p.rs != 0.0 ? p.etab[e] : 1.0;
            x = pf_scale(x, (F)((k2 != 0.0) ? we * woc / k2 : 0.0));
          }
          if (KEEP) src[m] = x; else v[m] = x;
        }
      }
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int e = tlj + m * nt;
        const int se = e > half ? e - n : e;
        C x = KEEP ? src[m] : v[m];
        const F kef = (F)(kf * se);
        if (mul == PF_MUL_K) x = pf_scale(x, kef);
        else if (mul == PF_MUL_K2) x = pf_scale(x, kef * kef);
        else if (mul == PF_MUL_IK) x = pf_mul_i<+1>(pf_scale(x, kef));
        v[m] = x;
      }
      // the last job on this input has its copy: the points of the NEXT input are requested into the registers that held it, and
      // travel during this job's stages and stores -- requested before those stores, so that waiting for them is not waiting for
      // the stores (as k_strided does it; y-pass 3 -> 6: two of three inputs arrive behind a job)
      if (KEEP && PF_MIXED_NEXT_INPUT) {
        ahead = j + 1 < p.njobs && p.job[j + 1].in != p.job[j].in;
        if (ahead) load_raw(reinterpret_cast<const C *>(p.job[j + 1].in), src, tlj);
      }
      C *__restrict__ outp = reinterpret_cast<C *>(p.job[j].out) + ((long long)outer * p.aout.os + col);
      auto store = [&](int e, C val) {
        if (!valid) return;
        if (p.out_ne > 0 && (unsigned)(e - p.out_e0) >= (unsigned)p.out_ne) return;
        pf_st_stream(outp + off_out((unsigned)e), val);
      };
      pf_bfly8<DIR>(v);  // stage 0: the thread's own eight points (NS = 1: no twiddles)
      if (nstages == 1) {
#pragma unroll
        for (int t = 0; t < 8; t++) store(tlj + t * nt, v[t]);  // n = 8: (b - k) R + k + t NS with b = k = 0
      } else {
#pragma unroll
        for (int t = 0; t < 8; t++) lds[(tlj * 8 + t) * T + c] = v[t];  // outputs of butterfly b = tl: b R + t
        __syncthreads();
        pf_mixed_tail_any<PLAN, 8, DIR, F>(
            pl, tlj, tw, 1, [&](int pos) { return lds[pos * T + c]; }, [&](int pos, C val) { lds[pos * T + c] = val; }, sync, store);
        __syncthreads();  // the next job rewrites the exchange area
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------------- z-pass ----
// A line's element p sits at LDS slot p + p / R0: the writes of stage 0 (a thread's R0 outputs are neighbours, the threads of a wave
// R0 apart) then fall on distinct banks instead of two (R0 = 8, 16-byte elements) -- PF_MIXED_PAD=0: A/B.
#ifndef PF_MIXED_PAD
#define PF_MIXED_PAD 1
#endif

#define PFP(q) (PF_MIXED_PAD ? (q) + ((q) >> (R0 == 8 ? 3 : 2)) : (q))
// c2r rows: Hermitian rows of n/2+1 -> n reals through the half-length complex transform (pf_c2r_pre), kz factor, 1/N^3 and
// the DC constant as k_c2r.  TL rows per workgroup, M / R0 threads per row (M = n / 2, R0 = 8 or 4).
template <typename F, int R0, typename PLAN = PfPlanRT>
__global__ void __launch_bounds__(256) k_mixed_c2r(const PfC2RParams p, const PfMixedPlan pl) {
  using C = pfc<F>;
  const int M = PLAN::n ? PLAN::n : pl.n, n = 2 * M, nt = M / R0, TL = blockDim.y;
  const int nstages = PLAN::n ? PLAN::nstages : pl.nstages;
  const int LPL = PFP(M) + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tl = threadIdx.x, l = threadIdx.y;  // (thread of the row, row of the tile)
  const int job = blockIdx.x % p.njobs;
  const long long row = (long long)(blockIdx.x / p.njobs) * TL + l;
  const bool rvalid = row < p.nlines;
  const C *__restrict__ in = reinterpret_cast<const C *>(p.job[job].in) + (rvalid ? row : 0) * p.in_pitch;
  const int mul = p.job[job].mul;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);   // exp(+2 pi i j / n), n entries
  C *L = lds + l * LPL;
  // the row's own threads stage it: R0 elements each, asked for in one go (M = R0 nt), and element M by the first thread; a thread's
  // own elements stay in its registers for the fold
  C own[R0], v[R0];
#pragma unroll
  for (int m = 0; m < R0; m++) {
    const int k = tl + m * nt;
    own[m] = (rvalid && k <= p.band_k) ? pf_ld_stream(in + k) : pf_mk<F>(0, 0);
  }
  if (tl == 0) L[PFP(M)] = (rvalid && M <= p.band_k) ? pf_ld_stream(in + M) : pf_mk<F>(0, 0);
#pragma unroll
  for (int m = 0; m < R0; m++) v[m] = tw[tl + m * nt];  // the fold's table values, asked for in front of the barrier
#pragma unroll
  for (int m = 0; m < R0; m++) L[PFP(tl + m * nt)] = own[m];
  __syncthreads();
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)n);
#pragma unroll
  for (int m = 0; m < R0; m++) {
    const int e = tl + m * nt;
    v[m] = pf_zfold<F>(own[m], L[PFP(M - e)], e, M, mul, kf, v[m], m == 0);
  }
  __syncthreads();
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;
  const int of32 = p.job[job].out_f32;
  float2 *__restrict__ o32 = reinterpret_cast<float2 *>(reinterpret_cast<float *>(p.job[job].out) + row * (long long)n);
  C *__restrict__ o64 = reinterpret_cast<C *>(reinterpret_cast<F *>(p.job[job].out) + row * (of32 == 2 ? (long long)n : p.out_pitch));
  auto store = [&](int pos, C val) {  // complex j of the half-length line = reals 2 j, 2 j + 1 of the row
    if (!rvalid) return;
    const F a = pf_norm_dc(val.x, norm, dcv), b = pf_norm_dc(val.y, norm, dcv);
    if (of32 == 1) o32[pos] = make_float2((float)a, (float)b);
    else pf_st_stream(o64 + pos, pf_mk<F>(a, b));
  };
  auto sync = [&]() { __syncthreads(); };
  pf_dft_small<R0, +1>(v);
  if (nstages == 1) {
#pragma unroll
    for (int t = 0; t < R0; t++) store(tl + t * nt, v[t]);
  } else {
#pragma unroll
    for (int t = 0; t < R0; t++) L[PFP(tl * R0 + t)] = v[t];
    __syncthreads();
    pf_mixed_tail_any<PLAN, R0, +1, F>(
        pl, tl, tw, 2, [&](int pos) { return L[PFP(pos)]; }, [&](int pos, C val) { L[PFP(pos)] = val; }, sync, store);
  }
}

// r2c rows (forward z-pass of the LPT sources), in place like k_r2c: real row -> n/2+1 complex
template <typename F, int R0, typename PLAN = PfPlanRT>
__global__ void __launch_bounds__(256) k_mixed_r2c(const PfR2CParams p, const PfMixedPlan pl) {
  using C = pfc<F>;
  const int M = PLAN::n ? PLAN::n : pl.n, nt = M / R0, TL = blockDim.y;
  const int nstages = PLAN::n ? PLAN::nstages : pl.nstages;
  const int LPL = PFP(M) + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tl = threadIdx.x, l = threadIdx.y;
  const long long row = (long long)blockIdx.x * TL + l;
  const bool valid = row < p.nlines;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  C *L = lds + l * LPL;
  C v[R0];
  {
    const C *in = reinterpret_cast<const C *>(reinterpret_cast<const F *>(p.in) + row * p.in_pitch);
#pragma unroll
    for (int m = 0; m < R0; m++) v[m] = valid ? pf_ld_stream(in + tl + m * nt) : pf_mk<F>(0, 0);
  }
  auto sync = [&]() { __syncthreads(); };
  auto keep = [&](int pos, C val) { L[PFP(pos)] = val; };  // the last stage leaves Z in LDS: the post-processing reads Z[k] and Z[M - k]
  __syncthreads();  // (in place: every row of the tile is in registers before anything is stored)
  pf_dft_small<R0, -1>(v);
  if (nstages == 1) {
#pragma unroll
    for (int t = 0; t < R0; t++) L[PFP(tl + t * nt)] = v[t];
  } else {
#pragma unroll
    for (int t = 0; t < R0; t++) L[PFP(tl * R0 + t)] = v[t];
    __syncthreads();
    pf_mixed_tail_any<PLAN, R0, -1, F>(
        pl, tl, tw, 2, [&](int pos) { return L[PFP(pos)]; }, keep, sync, keep);
  }
  __syncthreads();
  if (valid) {  // the row's own threads write it back: X[k] from Z[k] and Z[M - k]
    C *out = reinterpret_cast<C *>(p.out) + row * p.out_pitch;
    for (int k = tl; k <= M; k += nt) {
      if (k == M) out[M] = pf_mk<F>(L[PFP(0)].x - L[PFP(0)].y, (F)0);
      else pf_st_stream(out + k, pf_r2c_post<F>(L[PFP(k)], L[PFP(k == 0 ? 0 : M - k)], tw[k]));
    }
  }
}

// The z-pass of the sweep for these sizes, as k_c2r_invariants (pf_fft_kernels.hip): the six components of one row per workgroup
// (line l = threadIdx.y is component l), each transformed as in k_mixed_c2r with the real row left in its LDS line; then every
// thread of the workgroup reduces cells of the row -- MODE 0: the three invariants of the tensor, stored in place of the first three
// components (fp64 fields) or as fp64 rows of their own (fp32 fields: inv_out, inv_pitch); MODE 1: the 3LPT(b) source updated in
// place from the six rows and the first-order Hessian in job[].out.  The values are formed by the expressions of the six-field
// path (pf_norm_dc, pf_invariants, pf_lpt3b_accumulate): the same bits.  Workgroups walk over rows (grid-stride).
// (Run-time plans with R0 = 8: rows of at most 2048 points, 128 threads per line -- 768 threads at most.)
template <typename F, int R0, typename PLAN, int MODE>
__global__ void __launch_bounds__(PLAN::n ? 6 * (PLAN::n / R0) : (R0 == 8 ? 768 : 1024)) k_mixed_c2r_invariants(const PfC2RParams p, const PfMixedPlan pl) {
  using C = pfc<F>;
  using F2 = typename pf_vec2<F>::type;
  constexpr bool IN_PLACE = sizeof(F) == 8;
  const int M = PLAN::n ? PLAN::n : pl.n, n = 2 * M, nt = M / R0;
  const int nstages = PLAN::n ? PLAN::nstages : pl.nstages;
  const int LPL = M + 1;  // (no padding here: the six lines of a 768-point fp64 row then fit a CU four times, and a fourth workgroup is worth
                          //  more than the banks -- 11.3 against 11.5 ms per launch at 768^3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tl = threadIdx.x, l = threadIdx.y;
  const int tid = l * nt + tl, nthr = 6 * nt;
  const C *__restrict__ in0 = reinterpret_cast<const C *>(p.job[l].in);
  const int mul = p.job[l].mul;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  C *L = lds + l * LPL;
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)n);
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;
  double *__restrict__ o1 = IN_PLACE ? reinterpret_cast<double *>(p.job[0].out) : p.inv_out[0],
         *__restrict__ o2 = IN_PLACE ? reinterpret_cast<double *>(p.job[1].out) : p.inv_out[1],
         *__restrict__ o3 = IN_PLACE ? reinterpret_cast<double *>(p.job[2].out) : p.inv_out[2];
  auto sync = [&]() { __syncthreads(); };
  auto keep = [&](int pos, C val) { L[pos] = pf_norm_dc2(val, norm, dcv); };  // complex j = reals 2 j, 2 j + 1
#pragma unroll 1
  for (long long row = blockIdx.x; row < p.nlines; row += gridDim.x) {
    int tlj = tl, tidj = tid;
    asm volatile("" : "+v"(tlj), "+v"(tidj));
    const C *__restrict__ in = in0 + row * p.in_pitch;
    C own[R0], v[R0];  // (as k_mixed_c2r: the row asked for in one go, a thread's own elements kept for the fold)
#pragma unroll
    for (int m = 0; m < R0; m++) {
      const int k = tlj + m * nt;
      own[m] = (k <= p.band_k) ? pf_ld_stream(in + k) : pf_mk<F>(0, 0);
    }
    if (tlj == 0) L[M] = (M <= p.band_k) ? pf_ld_stream(in + M) : pf_mk<F>(0, 0);
#pragma unroll
    for (int m = 0; m < R0; m++) v[m] = tw[tlj + m * nt];  // the fold's table values, asked for in front of the barrier
#pragma unroll
    for (int m = 0; m < R0; m++) L[tlj + m * nt] = own[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < R0; m++) {
      const int e = tlj + m * nt;
      v[m] = pf_zfold<F>(own[m], L[M - e], e, M, mul, kf, v[m], m == 0);
    }
    __syncthreads();
    pf_dft_small<R0, +1>(v);
    if (nstages == 1) {
#pragma unroll
      for (int t = 0; t < R0; t++) keep(tlj + t * nt, v[t]);
    } else {
#pragma unroll
      for (int t = 0; t < R0; t++) L[tlj * R0 + t] = v[t];
      __syncthreads();
      pf_mixed_tail_any<PLAN, R0, +1, F>(
          pl, tlj, tw, 2, [&](int pos) { return L[pos]; }, [&](int pos, C val) { L[pos] = val; }, sync, keep);
    }
    __syncthreads();
    // per cell: six components -> three invariants (or the contraction); two neighbouring cells per thread
    for (int c = 2 * tidj; c < n; c += 2 * nthr) {
      F2 h[6];
#pragma unroll
      for (int k = 0; k < 6; k++) h[k] = *reinterpret_cast<const F2 *>(reinterpret_cast<const F *>(lds + k * LPL) + c);
      const double da[6] = {(double)h[0].x, (double)h[1].x, (double)h[2].x, (double)h[3].x, (double)h[4].x, (double)h[5].x},
                   db[6] = {(double)h[0].y, (double)h[1].y, (double)h[2].y, (double)h[3].y, (double)h[4].y, (double)h[5].y};
      const long long a = row * p.out_pitch + c;
      if (MODE == 0) {
        double a1, a2, a3, b1, b2, b3;
        pf_invariants(da, a1, a2, a3);
        pf_invariants(db, b1, b2, b3);
        if (pf_invariants_lose_diagonal(da, a1, a2) || pf_invariants_lose_diagonal(db, b1, b2)) *p.flag = 1.0;  // (as k_c2r_invariants: the sweep is repeated)
        const long long ao = IN_PLACE ? a : row * p.inv_pitch + c;
        typedef double pf_d2 __attribute__((ext_vector_type(2)));
        pf_d2 w1, w2, w3;
        w1.x = a1; w1.y = b1; w2.x = a2; w2.y = b2; w3.x = a3; w3.y = b3;
        __builtin_nontemporal_store(w1, reinterpret_cast<pf_d2 *>(o1 + ao));
        __builtin_nontemporal_store(w2, reinterpret_cast<pf_d2 *>(o2 + ao));
        __builtin_nontemporal_store(w3, reinterpret_cast<pf_d2 *>(o3 + ao));
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
    __syncthreads();  // the lines are rewritten by the next row
  }
}

// The same pass (MODE 0) with REDUCING threads, as k_c2r_invariants_spec of the power-of-two sizes (round 6): the 6 (M / R0) threads that
// transform are followed by the rest of their last wave and by `extra` more waves that only reduce.  When a row's six real lines are in LDS
// the transforming threads ask for their next row at once -- it travels while the others fold the row into its invariants, so a row's
// load latency is off the workgroup's critical path, and the reduction no longer waits behind it either.  One loop for every thread
// (a wave may hold threads of both kinds: 6 M / R0 is rarely a multiple of 64), every barrier met by all; the reducing threads pass
// through the stages with a butterfly index beyond the last one.  Same expressions per cell, same bits.
#ifndef PF_MIXED_ZI_SPEC
#define PF_MIXED_ZI_SPEC 0   // measured and not kept (profiles/r06_notes.md): 768^3 11.6 against 11.4 ms per launch, 1000^3 27.4 against 18.5 -- fewer workgroups
                              // fit a CU (six waves of 108 registers: two instead of four), and four resident workgroups hide a row's latency better than one early request
#endif
#ifndef PF_MIXED_ZI_EXTRA
#define PF_MIXED_ZI_EXTRA 1   // whole waves of reducing threads beyond the one that the transforming threads leave half empty
#endif
#if PF_MIXED_ZI_SPEC
template <typename F, int R0, typename PLAN>
__global__ void __launch_bounds__(PLAN::n ? ((6 * (PLAN::n / R0) + 63) / 64 + PF_MIXED_ZI_EXTRA) * 64 : 1024)
k_mixed_c2r_invariants_spec(const PfC2RParams p, const PfMixedPlan pl) {
  using C = pfc<F>;
  using F2 = typename pf_vec2<F>::type;
  constexpr bool IN_PLACE = sizeof(F) == 8;
  const int M = PLAN::n ? PLAN::n : pl.n, n = 2 * M, nt = M / R0;
  const int nstages = PLAN::n ? PLAN::nstages : pl.nstages;
  const int LPL = M + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);
  const int tid = threadIdx.x, ntr = 6 * nt, ntr_w = (ntr + 63) & ~63;   // the transforming threads fill whole waves (the last one in part)
  if (tid >= ntr_w) {
    // ---- the reducing waves: a loop of their own (nothing of the transforms' state is alive here), the same barriers as below
    const int nred = (int)blockDim.x - ntr_w;
    const int nbar = 2 + (nstages > 1 ? 2 * (nstages - 1) : 0);   // barriers of a row in front of its reduction
    double *__restrict__ o1 = IN_PLACE ? reinterpret_cast<double *>(p.job[0].out) : p.inv_out[0],
           *__restrict__ o2 = IN_PLACE ? reinterpret_cast<double *>(p.job[1].out) : p.inv_out[1],
           *__restrict__ o3 = IN_PLACE ? reinterpret_cast<double *>(p.job[2].out) : p.inv_out[2];
#pragma unroll 1
    for (long long row = blockIdx.x; row < p.nlines; row += gridDim.x) {
      int trj = tid - ntr_w;
      asm volatile("" : "+v"(trj));
      for (int i = 0; i < nbar; i++) __syncthreads();
      __syncthreads();
      for (int c = 2 * trj; c < n; c += 2 * nred) {
        F2 h[6];
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] = *reinterpret_cast<const F2 *>(reinterpret_cast<const F *>(lds + k * LPL) + c);
        const double da[6] = {(double)h[0].x, (double)h[1].x, (double)h[2].x, (double)h[3].x, (double)h[4].x, (double)h[5].x},
                     db[6] = {(double)h[0].y, (double)h[1].y, (double)h[2].y, (double)h[3].y, (double)h[4].y, (double)h[5].y};
        double a1, a2, a3, b1, b2, b3;
        pf_invariants(da, a1, a2, a3);
        pf_invariants(db, b1, b2, b3);
        if (pf_invariants_lose_diagonal(da, a1, a2) || pf_invariants_lose_diagonal(db, b1, b2)) *p.flag = 1.0;  // (as k_c2r_invariants: the sweep is repeated)
        const long long ao = IN_PLACE ? row * p.out_pitch + c : row * p.inv_pitch + c;
        typedef double pf_d2 __attribute__((ext_vector_type(2)));
        pf_d2 w1, w2, w3;
        w1.x = a1; w1.y = b1; w2.x = a2; w2.y = b2; w3.x = a3; w3.y = b3;
        __builtin_nontemporal_store(w1, reinterpret_cast<pf_d2 *>(o1 + ao));
        __builtin_nontemporal_store(w2, reinterpret_cast<pf_d2 *>(o2 + ao));
        __builtin_nontemporal_store(w3, reinterpret_cast<pf_d2 *>(o3 + ao));
      }
      __syncthreads();
    }
    return;
  }
  // ---- the transforming waves (threads ntr .. ntr_w - 1 of the last one only keep the barriers company)
  const bool is_tr = tid < ntr;
  const int l = is_tr ? tid / nt : 0;
  const int tl = is_tr ? tid - l * nt : (1 << 20);   // (idle threads: beyond every butterfly of every stage)
  const C *__restrict__ in0 = reinterpret_cast<const C *>(p.job[l].in);
  const int mul = p.job[l].mul;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  C *L = lds + l * LPL;
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)n);
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;
  auto sync = [&]() { __syncthreads(); };
  auto keep = [&](int pos, C val) { L[pos] = pf_norm_dc2(val, norm, dcv); };  // complex j = reals 2 j, 2 j + 1
  C own[R0 + 1];
  auto fetch = [&](long long row) {
    const C *__restrict__ in = in0 + row * p.in_pitch;
#pragma unroll
    for (int m = 0; m < R0; m++) {
      const int k = tl + m * nt;
      own[m] = (is_tr && k <= p.band_k) ? pf_ld_stream(in + k) : pf_mk<F>(0, 0);
    }
    own[R0] = (tl == 0 && M <= p.band_k) ? pf_ld_stream(in + M) : pf_mk<F>(0, 0);
  };
  if ((long long)blockIdx.x < p.nlines) fetch(blockIdx.x);
#pragma unroll 1
  for (long long row = blockIdx.x; row < p.nlines; row += gridDim.x) {
    int tlj = tl;
    asm volatile("" : "+v"(tlj));
    C v[R0];
    if (is_tr) {
#pragma unroll
      for (int m = 0; m < R0; m++) v[m] = tw[tlj + m * nt];  // the fold's table values, asked for in front of the barrier
#pragma unroll
      for (int m = 0; m < R0; m++) L[tlj + m * nt] = own[m];
      if (tlj == 0) L[M] = own[R0];
    }
    __syncthreads();
    if (is_tr) {
#pragma unroll
      for (int m = 0; m < R0; m++) {
        const int e = tlj + m * nt;
        v[m] = pf_zfold<F>(own[m], L[M - e], e, M, mul, kf, v[m], m == 0);
      }
    }
    __syncthreads();
    if (is_tr) {
      pf_dft_small<R0, +1>(v);
      if (nstages == 1) {
#pragma unroll
        for (int t = 0; t < R0; t++) keep(tlj + t * nt, v[t]);
      } else {
#pragma unroll
        for (int t = 0; t < R0; t++) L[tlj * R0 + t] = v[t];
      }
    }
    if (nstages > 1) {
      __syncthreads();
      pf_mixed_tail_any<PLAN, R0, +1, F>(
          pl, tlj, tw, 2, [&](int pos) { return L[pos]; }, [&](int pos, C val) { L[pos] = val; }, sync, keep);
    }
    __syncthreads();
    // this row is the reducing waves' now: the next one travels meanwhile.  (Past the last row: zeros, not "nothing" -- a conditional
    // request would keep the old row alive through the stages)
    if (row + gridDim.x < p.nlines) fetch(row + gridDim.x);
    else {
#pragma unroll
      for (int m = 0; m <= R0; m++) own[m] = pf_mk<F>(0, 0);
    }
    __syncthreads();  // the lines are rewritten by the next row
  }
}
#endif  // PF_MIXED_ZI_SPEC

// --------------------------------------------------------------------------------------------------------- launch ----
// radices of a line of n points: first R0 (8, or 4 where allowed), then 8s, a 4 or a 2, 5s and 3s.  ok = false: n has another
// prime factor (or no admissible first radix) -- the caller keeps the chirp-z path for such sizes.  One rule, evaluated at run time
// (pf_mixed_plan) and at compile time (PfPlanOf: the sizes of PF_MIXED_CT_SIZES get kernels with the plan built in).
struct PfRadices { bool ok; int n, ns; int r[PF_MIXED_MAX_STAGES]; };
constexpr PfRadices pf_radices(int n, bool allow4) {
  PfRadices p{};
  p.n = n;
  const int r0 = (n % 8 == 0) ? 8 : ((allow4 && n % 4 == 0) ? 4 : 0);
  if (!r0 || n < r0) return p;
  int rest = n / r0;
  p.r[p.ns++] = r0;
  while (rest % 8 == 0) { if (p.ns >= PF_MIXED_MAX_STAGES) return p; p.r[p.ns++] = 8; rest /= 8; }
  if (rest % 4 == 0) { if (p.ns >= PF_MIXED_MAX_STAGES) return p; p.r[p.ns++] = 4; rest /= 4; }
  if (rest % 2 == 0) { if (p.ns >= PF_MIXED_MAX_STAGES) return p; p.r[p.ns++] = 2; rest /= 2; }
  while (rest % 5 == 0) { if (p.ns >= PF_MIXED_MAX_STAGES) return p; p.r[p.ns++] = 5; rest /= 5; }
  while (rest % 3 == 0) { if (p.ns >= PF_MIXED_MAX_STAGES) return p; p.r[p.ns++] = 3; rest /= 3; }
  p.ok = rest == 1;
  return p;
}
#if PF_MIXED_PART == 0
bool pf_mixed_plan(int n, bool allow4, PfMixedPlan *pl) {
  memset(pl, 0, sizeof(*pl));
  pl->n = n;
  const PfRadices r = pf_radices(n, allow4);
  if (!r.ok) return false;
  pl->nstages = r.ns;
  unsigned long long nsv = 1;
  for (int s = 0; s < r.ns; s++) {  // ceil(2^32 / NS_s): b mod NS_s by one multiply-high (pf_mixed_stage); stage 0 needs none
    pl->radix[s] = r.r[s];
    pl->magic[s] = s == 0 ? 0u : (unsigned)(((1ull << 32) + nsv - 1) / nsv);
    nsv *= (unsigned long long)r.r[s];
  }
  return true;
}
bool pf_mixed_supported(int n) {
  PfMixedPlan a, b;
  return n >= 8 && n <= 2048 && n % 8 == 0 && pf_mixed_plan(n, false, &a) && pf_mixed_plan(n / 2, true, &b);
}
#endif
// Grid sizes whose plans are compiled in: the strided passes on N points (first radix 8) and the z-passes on N / 2 (8 or 4).
// (bench.py names the kernels of a run from the same list: MIXED_CT_SIZES there, held against this line by tests/test_bench_report.py.)
#ifndef PF_MIXED_CT
#define PF_MIXED_CT 1   // (0 in an A/B build: every size through the run-time plans)
#endif
// Round 6: every n = 8 m >= 96, m = 2^a 3^b 5^c, has its plan compiled in (38 sizes; the run-time plans, at 0.6-0.7 of their rate, are left
// with the sizes below 96).  The kernels of 38 sizes are too many for one translation unit to compile in reasonable time: this file is
// compiled three times (csrc/Makefile) -- PF_MIXED_PART 0: everything but the kernels of the sizes of lists 1 and 2; PF_MIXED_PART 1, 2:
// only the launchers of their own list, under the names pf_launch_mixed_*_p1 / _p2, which part 0 calls for those sizes.
#define PF_MIXED_CT_SIZES(X) X(200) X(384) X(400) X(640) X(768) X(800) X(1000) X(1280) X(1536) X(1600) X(2000)
#define PF_MIXED_CT_SIZES_1(X) X(96) X(120) X(144) X(160) X(192) X(216) X(240) X(288) X(320) X(360) X(432) X(480) X(576)
#define PF_MIXED_CT_SIZES_2(X) X(600) X(648) X(720) X(864) X(960) X(1080) X(1152) X(1200) X(1296) X(1440) X(1728) X(1800) X(1920) X(1944)
#ifndef PF_MIXED_PART
#define PF_MIXED_PART 0
#endif
#if PF_MIXED_PART == 0
#define PF_MIXED_MY_SIZES(X) PF_MIXED_CT_SIZES(X)
#elif PF_MIXED_PART == 1
#define PF_MIXED_MY_SIZES(X) PF_MIXED_CT_SIZES_1(X)
#else
#define PF_MIXED_MY_SIZES(X) PF_MIXED_CT_SIZES_2(X)
#endif
// which part holds the plan of n: 0 this list, 1 / 2 the others, -1 none (run-time plan)
static int pf_mixed_part_of(int n) {
  if (!PF_MIXED_CT) return -1;
#define PF_MIXED_CASE(NN) if (n == NN) return 0;
  PF_MIXED_CT_SIZES(PF_MIXED_CASE)
#undef PF_MIXED_CASE
#define PF_MIXED_CASE(NN) if (n == NN) return 1;
  PF_MIXED_CT_SIZES_1(PF_MIXED_CASE)
#undef PF_MIXED_CASE
#define PF_MIXED_CASE(NN) if (n == NN) return 2;
  PF_MIXED_CT_SIZES_2(PF_MIXED_CASE)
#undef PF_MIXED_CASE
  return -1;
}
int pf_launch_mixed_strided_p1(int fb, int n, int dir, const PfStridedParams &p, hipStream_t st);
int pf_launch_mixed_strided_p2(int fb, int n, int dir, const PfStridedParams &p, hipStream_t st);
int pf_launch_mixed_c2r_p1(int fb, int n, const PfC2RParams &p, hipStream_t st);
int pf_launch_mixed_c2r_p2(int fb, int n, const PfC2RParams &p, hipStream_t st);
int pf_launch_mixed_r2c_p1(int fb, int n, const PfR2CParams &p, hipStream_t st);
int pf_launch_mixed_r2c_p2(int fb, int n, const PfR2CParams &p, hipStream_t st);
int pf_launch_mixed_c2r_invariants_p1(int fb, int n, const PfC2RParams &p, hipStream_t st, int mode);
int pf_launch_mixed_c2r_invariants_p2(int fb, int n, const PfC2RParams &p, hipStream_t st, int mode);
#if PF_MIXED_PART == 1
#define pf_launch_mixed_strided pf_launch_mixed_strided_p1
#define pf_launch_mixed_c2r pf_launch_mixed_c2r_p1
#define pf_launch_mixed_r2c pf_launch_mixed_r2c_p1
#define pf_launch_mixed_c2r_invariants pf_launch_mixed_c2r_invariants_p1
#elif PF_MIXED_PART == 2
#define pf_launch_mixed_strided pf_launch_mixed_strided_p2
#define pf_launch_mixed_c2r pf_launch_mixed_c2r_p2
#define pf_launch_mixed_r2c pf_launch_mixed_r2c_p2
#define pf_launch_mixed_c2r_invariants pf_launch_mixed_c2r_invariants_p2
#endif
// part 0: a size of another part goes to that part's launcher; parts 1, 2: a size that is not theirs is nobody's business here
#if PF_MIXED_PART == 0
#define PF_MIXED_ROUTE(call1, call2) do { const int part__ = pf_mixed_part_of(n); if (part__ == 1) return call1; if (part__ == 2) return call2; } while (0)
#define PF_MIXED_RT(stmt) stmt
#else
#define PF_MIXED_ROUTE(call1, call2) do { if (pf_mixed_part_of(n) != PF_MIXED_PART) return 2; } while (0)
#define PF_MIXED_RT(stmt) return 2
#endif
template <int N, bool A4, typename = std::make_index_sequence<(size_t)pf_radices(N, A4).ns>> struct PfPlanOf;
template <int N, bool A4, size_t... I> struct PfPlanOf<N, A4, std::index_sequence<I...>> {
  static_assert(pf_radices(N, A4).ok, "a size of PF_MIXED_CT_SIZES without a plan");
  using type = PfPlanCT<pf_radices(N, A4).r[I]...>;
};
template <int N> using PfPlanS = typename PfPlanOf<N, false>::type;      // strided passes of an N grid
template <int N> using PfPlanZ = typename PfPlanOf<N / 2, true>::type;   // its z-passes: half-length lines
template <int N> constexpr bool pf_mixed_inv_fits() {  // (the six lines of a row in one workgroup: k_mixed_c2r_invariants; 1800 and 1944 do not)
  return 6 * ((N / 2) / PfPlanZ<N>::radix[0]) <= 1024;
}

#if PF_MIXED_PART == 0
bool pf_mixed_plan_compiled_in(int n) { return pf_mixed_part_of(n) >= 0; }
#endif
// (the LDS attribute and the occupancy belong to a kernel ON ONE DEVICE; asked once per (kernel, device[, shape]) under a lock, so that
//  ranks run as threads of one process on several devices neither race nor inherit each other's answers)
static std::mutex pf_mixed_mu;
template <typename F> static int pf_mixed_raise_lds(const void *fn, size_t shm) {
  if (shm <= 64 * 1024) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 3;
  struct Ent { const void *fn; int dev; size_t shm; };
  static std::vector<Ent> done;
  std::lock_guard<std::mutex> lock(pf_mixed_mu);
  for (const Ent &e : done)
    if (e.fn == fn && e.dev == dev && e.shm >= shm) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 3;
  done.push_back(Ent{fn, dev, shm});
  return 0;
}
// workgroups of a kernel that fit one CU (asked once per kernel, device and shape)
static int pf_mixed_resident(const void *fn, int threads, size_t shm) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  struct Ent { const void *fn; int dev, threads; size_t shm; int nb; };
  static std::vector<Ent> tab;
  std::lock_guard<std::mutex> lock(pf_mixed_mu);
  for (const Ent &e : tab)
    if (e.fn == fn && e.dev == dev && e.threads == threads && e.shm == shm) return e.nb;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, shm) != hipSuccess) { (void)hipGetLastError(); return 0; }
  tab.push_back(Ent{fn, dev, threads, shm, nb});
  return nb;
}
int pf_launch_mixed_strided(int fb, int n, int dir, const PfStridedParams &p, hipStream_t st) {
  PF_MIXED_ROUTE(pf_launch_mixed_strided_p1(fb, n, dir, p, st), pf_launch_mixed_strided_p2(fb, n, dir, p, st));
  PfMixedPlan pl;
  if (!pf_mixed_plan(n, false, &pl)) return 2;
  const int nt = n / 8;
  const int w = fb == 8 ? 16 : 8;  // bytes of a complex element
  int T = 128 / w;                 // whole 128-byte row segments where LDS and the thread budget allow
  while (T > 1 && ((size_t)n * T * w > 128 * 1024 || nt * T > 1024)) T >>= 1;
  if (nt * T > 1024 || (size_t)n * T * w > 160 * 1024) return 2;
  const int ntiles = (p.ncols + T - 1) / T;
  const long long nwork = (long long)ntiles * p.nouter;
  // slabs of any length (PfAddr::el_len): the split by multiply-high
  if (p.ain.el_len < 2 || p.aout.el_len < 2) return 2;  // (a slab of one plane would need the quotient e itself: 2^32 does not fit the multiplier)
  pl.magic_in = (unsigned)((0x100000000ull + (unsigned)p.ain.el_len - 1) / (unsigned)p.ain.el_len);
  pl.magic_out = (unsigned)((0x100000000ull + (unsigned)p.aout.el_len - 1) / (unsigned)p.aout.el_len);
  const dim3 grid((unsigned)(((nwork + 7) >> 3) << 3)), block(T, nt);
  const size_t shm = (size_t)n * T * w;
#if PF_MIXED_PART == 0
#define PF_MIXED_KEEP_TRY(FF, DD, PP)                                                                                \
    if (PF_MIXED_KEEP_RT && PP::n == 0 && block.x * block.y <= 768 && p.njobs > 1) {                                 \
      /* the form that keeps a tile over its jobs needs 156 registers for 124 (fp32: 106 for 94): taken where that costs no workgroup */ \
      /* per CU -- and with fp64 fields only where a tile fills the CU anyway (720^3: 395 -> 353 ms per step; 360^3, two workgroups */ \
      /* per CU either way: 49.8 -> 51.5, not taken); fp32 fields 720^3 349 -> 286, 360^3 44.0 -> 35.5 */ \
      if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_strided<FF, DD, PfPlanRT, true>), shm)) return 3; \
      if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_strided<FF, DD, PfPlanRT, false>), shm)) return 3; \
      const int threads = (int)(block.x * block.y);                                                                  \
      const int occ_keep = pf_mixed_resident(reinterpret_cast<const void *>(&k_mixed_strided<FF, DD, PfPlanRT, true>), threads, shm);  \
      const int occ_base = pf_mixed_resident(reinterpret_cast<const void *>(&k_mixed_strided<FF, DD, PfPlanRT, false>), threads, shm); \
      if (occ_keep >= occ_base && occ_keep >= 1 && (sizeof(FF) == 4 || occ_base == 1)) {                              \
        hipLaunchKernelGGL((k_mixed_strided<FF, DD, PfPlanRT, true>), grid, block, shm, st, p, pl, nwork, ntiles); \
        break;                                                                                                       \
      }                                                                                                              \
    }
#else
#define PF_MIXED_KEEP_TRY(FF, DD, PP)   // (the run-time plans live in part 0)
#endif
#define PF_MIXED_LAUNCH_P(FF, DD, PP)                                                                                \
  do {                                                                                                               \
    PF_MIXED_KEEP_TRY(FF, DD, PP)                                                                                    \
    if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_strided<FF, DD, PP>), shm)) return 3;         \
    hipLaunchKernelGGL((k_mixed_strided<FF, DD, PP>), grid, block, shm, st, p, pl, nwork, ntiles);                \
  } while (0)
#define PF_MIXED_CASE(NN) if (PF_MIXED_CT && n == NN) PF_MIXED_LAUNCH_P(FF_, DD_, PfPlanS<NN>); else
#define PF_MIXED_LAUNCH(FF, DD)                                                                                      \
  do {                                                                                                               \
    using FF_ = FF;                                                                                                  \
    constexpr int DD_ = DD;                                                                                          \
    PF_MIXED_MY_SIZES(PF_MIXED_CASE) PF_MIXED_RT(PF_MIXED_LAUNCH_P(FF_, DD_, PfPlanRT));                              \
  } while (0)
  if (fb == 8) { if (dir > 0) PF_MIXED_LAUNCH(double, +1); else PF_MIXED_LAUNCH(double, -1); }
  else { if (dir > 0) PF_MIXED_LAUNCH(float, +1); else PF_MIXED_LAUNCH(float, -1); }
#undef PF_MIXED_LAUNCH
#undef PF_MIXED_CASE
#undef PF_MIXED_LAUNCH_P
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int pf_mixed_line_slots(int M, int r0) { return (PF_MIXED_PAD ? M + (M >> (r0 == 8 ? 3 : 2)) : M) + 1; }  // (PFP(M) + 1)
static int pf_mixed_rows_per_wg(int nt) { int tl = 256 / nt; return tl < 1 ? 1 : tl; }  // (z-pass workgroups: at most 256 threads)
int pf_launch_mixed_c2r(int fb, int n, const PfC2RParams &p, hipStream_t st) {
  PF_MIXED_ROUTE(pf_launch_mixed_c2r_p1(fb, n, p, st), pf_launch_mixed_c2r_p2(fb, n, p, st));
  PfMixedPlan pl;
  const int M = n / 2;
  if (!pf_mixed_plan(M, true, &pl)) return 2;
  const int r0 = pl.radix[0], nt = M / r0, TL = pf_mixed_rows_per_wg(nt);
  if (TL * nt > 256) return 2;
  const long long nblk = (p.nlines + TL - 1) / TL;
  const dim3 grid((unsigned)(nblk * p.njobs)), block(nt, TL);
  const size_t shm = (size_t)TL * pf_mixed_line_slots(M, r0) * (fb == 8 ? 16 : 8);
#define PF_MIXED_LAUNCH_P(FF, RR, PP)                                                                                \
  do {                                                                                                               \
    if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_c2r<FF, RR, PP>), shm)) return 3;             \
    hipLaunchKernelGGL((k_mixed_c2r<FF, RR, PP>), grid, block, shm, st, p, pl);                                  \
  } while (0)
#define PF_MIXED_CASE(NN)                                                                                            \
  if (PF_MIXED_CT && n == NN) {                                                                                      \
    if (fb == 8) PF_MIXED_LAUNCH_P(double, PfPlanZ<NN>::radix[0], PfPlanZ<NN>); else PF_MIXED_LAUNCH_P(float, PfPlanZ<NN>::radix[0], PfPlanZ<NN>); \
  } else
  PF_MIXED_MY_SIZES(PF_MIXED_CASE)
#undef PF_MIXED_CASE
  PF_MIXED_RT(do {
    if (fb == 8) { if (r0 == 8) PF_MIXED_LAUNCH_P(double, 8, PfPlanRT); else PF_MIXED_LAUNCH_P(double, 4, PfPlanRT); }
    else { if (r0 == 8) PF_MIXED_LAUNCH_P(float, 8, PfPlanRT); else PF_MIXED_LAUNCH_P(float, 4, PfPlanRT); }
  } while (0));
#undef PF_MIXED_LAUNCH_P
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int pf_launch_mixed_r2c(int fb, int n, const PfR2CParams &p, hipStream_t st) {
  PF_MIXED_ROUTE(pf_launch_mixed_r2c_p1(fb, n, p, st), pf_launch_mixed_r2c_p2(fb, n, p, st));
  PfMixedPlan pl;
  const int M = n / 2;
  if (!pf_mixed_plan(M, true, &pl)) return 2;
  const int r0 = pl.radix[0], nt = M / r0, TL = pf_mixed_rows_per_wg(nt);
  if (TL * nt > 256) return 2;
  const long long nblk = (p.nlines + TL - 1) / TL;
  const dim3 grid((unsigned)nblk), block(nt, TL);
  const size_t shm = (size_t)TL * pf_mixed_line_slots(M, r0) * (fb == 8 ? 16 : 8);
#define PF_MIXED_LAUNCH_P(FF, RR, PP)                                                                                \
  do {                                                                                                               \
    if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_r2c<FF, RR, PP>), shm)) return 3;             \
    hipLaunchKernelGGL((k_mixed_r2c<FF, RR, PP>), grid, block, shm, st, p, pl);                                  \
  } while (0)
#define PF_MIXED_CASE(NN)                                                                                            \
  if (PF_MIXED_CT && n == NN) {                                                                                      \
    if (fb == 8) PF_MIXED_LAUNCH_P(double, PfPlanZ<NN>::radix[0], PfPlanZ<NN>); else PF_MIXED_LAUNCH_P(float, PfPlanZ<NN>::radix[0], PfPlanZ<NN>); \
  } else
  PF_MIXED_MY_SIZES(PF_MIXED_CASE)
#undef PF_MIXED_CASE
  PF_MIXED_RT(do {
    if (fb == 8) { if (r0 == 8) PF_MIXED_LAUNCH_P(double, 8, PfPlanRT); else PF_MIXED_LAUNCH_P(double, 4, PfPlanRT); }
    else { if (r0 == 8) PF_MIXED_LAUNCH_P(float, 8, PfPlanRT); else PF_MIXED_LAUNCH_P(float, 4, PfPlanRT); }
  } while (0));
#undef PF_MIXED_LAUNCH_P
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the invariant z-pass: six lines of n / 2 + 1 complex in LDS, six times (n / 2) / R0 threads
#if PF_MIXED_PART == 0
bool pf_mixed_invariants_supported(int fb, int n) {
  PfMixedPlan pl;
  if (!pf_mixed_supported(n) || !pf_mixed_plan(n / 2, true, &pl)) return false;
  const int nt = (n / 2) / pl.radix[0];
  return 6 * nt <= 1024 && (size_t)6 * (n / 2 + 1) * (fb == 8 ? 16 : 8) <= 128 * 1024;
}
#endif
int pf_launch_mixed_c2r_invariants(int fb, int n, const PfC2RParams &p, hipStream_t st, int mode) {
  if (!pf_mixed_invariants_supported(fb, n)) return 2;
  PF_MIXED_ROUTE(pf_launch_mixed_c2r_invariants_p1(fb, n, p, st, mode), pf_launch_mixed_c2r_invariants_p2(fb, n, p, st, mode));
  PfMixedPlan pl;
  const int M = n / 2;
  if (!pf_mixed_plan(M, true, &pl)) return 2;
  const int r0 = pl.radix[0], nt = M / r0;
  const size_t shm = (size_t)6 * (M + 1) * (fb == 8 ? 16 : 8);
  int per_cu = (int)((160 * 1024) / (shm + 512));
  const int by_threads = 2048 / (6 * nt);
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  long long g = (long long)(p.ncu > 0 ? p.ncu : 256) * per_cu;
  if (g > p.nlines) g = p.nlines;
  if (PF_MIXED_ZI_ONE_ROW) g = p.nlines;  // one row per workgroup
  const dim3 grid((unsigned)g), block(nt, 6);
  // MODE 0 with reducing threads (k_mixed_c2r_invariants_spec): the transforming threads, the rest of their last wave, PF_MIXED_ZI_EXTRA waves more
  const int spec_threads = ((6 * nt + 63) / 64 + PF_MIXED_ZI_EXTRA) * 64;   // (whole waves of either kind)
  const bool spec_ok = PF_MIXED_ZI_SPEC && spec_threads <= 1024 && pf_mixed_part_of(n) >= 0;   // (compiled-in plans only: with a run-time plan the kernel spills)
  (void)spec_ok;
#if PF_MIXED_ZI_SPEC
#define PF_MIXED_SPEC_TRY(FF, RR, PP, MM)                                                                            \
    if constexpr (MM == 0) {                                                                                         \
      if (spec_ok) {                                                                                                 \
        if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_c2r_invariants_spec<FF, RR, PP>), shm)) return 3; \
        /* workgroups that stay: as many as the chip holds at once, each walking over rows (the next row is what a workgroup asks for early) */ \
        const int occ = pf_mixed_resident(reinterpret_cast<const void *>(&k_mixed_c2r_invariants_spec<FF, RR, PP>), spec_threads, shm); \
        long long gs = (long long)(p.ncu > 0 ? p.ncu : 256) * (occ > 0 ? occ : 1);                                    \
        if (gs > p.nlines) gs = p.nlines;                                                                            \
        hipLaunchKernelGGL((k_mixed_c2r_invariants_spec<FF, RR, PP>), dim3((unsigned)gs), dim3(spec_threads), shm, st, p, pl); \
        break;                                                                                                       \
      }                                                                                                              \
    }
#else
#define PF_MIXED_SPEC_TRY(FF, RR, PP, MM)
#endif
#define PF_MIXED_LAUNCH_P(FF, RR, PP, MM)                                                                            \
  do {                                                                                                               \
    PF_MIXED_SPEC_TRY(FF, RR, PP, MM)                                                                                \
    if (pf_mixed_raise_lds<FF>(reinterpret_cast<const void *>(&k_mixed_c2r_invariants<FF, RR, PP, MM>), shm)) return 3; \
    hipLaunchKernelGGL((k_mixed_c2r_invariants<FF, RR, PP, MM>), grid, block, shm, st, p, pl);                    \
  } while (0)
#define PF_MIXED_CASE(NN) if constexpr (pf_mixed_inv_fits<NN>()) if (PF_MIXED_CT && n == NN) { PF_MIXED_LAUNCH_P(FF_, PfPlanZ<NN>::radix[0], PfPlanZ<NN>, MM_); break; }
#define PF_MIXED_LAUNCH(FF, MM)                                                                                      \
  do {                                                                                                               \
    using FF_ = FF;                                                                                                  \
    constexpr int MM_ = MM;                                                                                          \
    PF_MIXED_MY_SIZES(PF_MIXED_CASE)                                                                                 \
    PF_MIXED_RT(do { if (r0 == 8) PF_MIXED_LAUNCH_P(FF_, 8, PfPlanRT, MM_); else PF_MIXED_LAUNCH_P(FF_, 4, PfPlanRT, MM_); } while (0)); \
  } while (0)
  if (fb == 8) { if (mode == 0) PF_MIXED_LAUNCH(double, 0); else PF_MIXED_LAUNCH(double, 1); }
  else { if (mode == 0) PF_MIXED_LAUNCH(float, 0); else PF_MIXED_LAUNCH(float, 1); }
#undef PF_MIXED_LAUNCH
#undef PF_MIXED_CASE
#undef PF_MIXED_LAUNCH_P
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
