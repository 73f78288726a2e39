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
// Several (input, multiplier, output) jobs share one launch with the job index
// fastest in blockIdx.x, so workgroups that read the same input tile run
// concurrently and the re-reads are served by the Infinity Cache.
#include "pf_internal.h"
#include "pf_fft_core.h"

template <typename F> using C_t = pfc<F>;

template <typename F, int N, int DIR, int TWS, int S = 0>
struct PfStages {
  template <typename WR, typename RD>
  static __device__ __forceinline__ void run(pfc<F> (&v)[8], int tl, const pfc<F> *__restrict__ tw, WR wr, RD rd) {
    pf_stage<F, N, S, DIR, TWS>(v, tl, tw);
    if constexpr (S + 1 < pf_nstages(N)) {
      constexpr int NT = N / 8;
#pragma unroll
      for (int m = 0; m < 8; m++) wr(pf_stage_pos<N, S>(tl, m), v[m]);
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; m++) v[m] = rd(tl + m * NT);
      __syncthreads();
      PfStages<F, N, DIR, TWS, S + 1>::run(v, tl, tw, wr, rd);
    }
  }
};

__device__ __forceinline__ long long pf_addr(const PfAddr &a, int outer, int e, int col) {
  return (long long)outer * a.os + (long long)(e / a.el) * a.ehs + (long long)(e % a.el) * a.els + col;
}

template <typename F, int N, int T, int DIR>
__global__ void __launch_bounds__(T *N / 8) k_strided(const PfStridedParams p) {
  using C = pfc<F>;
  constexpr int NT = N / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);  // [N][T]
  const int tid = threadIdx.x;
  const int c = tid % T, tl = tid / T;
  const int job = blockIdx.x % p.njobs;
  const int tile = blockIdx.x / p.njobs;
  const int outer = blockIdx.y;
  const int col = tile * T + c;
  const bool valid = col < p.ncols;
  const C *__restrict__ in = reinterpret_cast<const C *>(p.job[job].in);
  C *__restrict__ out = reinterpret_cast<C *>(p.job[job].out);
  const int mul = p.job[job].mul;
  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);

  C v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int e = tl + m * NT;
    v[m] = valid ? in[pf_addr(p.ain, outer, e, col)] : pf_mk<F>(0, 0);
  }

  if (p.pre || mul != PF_MUL_ONE) {
    const double kf = 2.0 * 3.14159265358979323846 / (double)N;
    double ko2kc2 = 0.0;
    if (p.pre) {
      int so = outer + p.outer_offset;
      if (so > N / 2) so -= N;
      const double ko = kf * so, kc = kf * col;
      ko2kc2 = ko * ko + kc * kc;
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tl + m * NT;
      const int se = e > N / 2 ? e - N : e;
      const double ke = kf * se;
      double fac = 1.0;
      if (p.pre) {
        const double k2 = ke * ke + ko2kc2;
        fac = (k2 != 0.0) ? exp(-0.5 * k2 * p.rs * p.rs) * p.growth / k2 : 0.0;
      }
      if (mul == PF_MUL_K || mul == PF_MUL_IK) fac *= ke;
      if (mul == PF_MUL_K2) fac *= ke * ke;
      v[m] = pf_scale(v[m], (F)fac);
      if (mul == PF_MUL_IK) v[m] = pf_mul_i<+1>(v[m]);
    }
  }

  PfStages<F, N, DIR, 1>::run(
      v, tl, tw, [&](int pos, C val) { lds[pos * T + c] = val; }, [&](int pos) { return lds[pos * T + c]; });

  if (valid) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int e = tl + m * NT;
      out[pf_addr(p.aout, outer, e, col)] = v[m];
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
    lds[ll * LPL + k] = row < p.nlines ? in[row * p.in_pitch + k] : pf_mk<F>(0, 0);
  }
  __syncthreads();

  // phase B: kz factor + Hermitian fold into the half-length complex line
  C *L = lds + l * LPL;
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)N);
  C v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int e = tl + m * NT;
    C xk = L[e], xmk = L[M - e];
    if (mul != PF_MUL_ONE) {
      F fk = kf * (F)e, fm = kf * (F)(M - e);
      if (mul == PF_MUL_K2) { fk *= fk; fm *= fm; }
      xk = pf_scale(xk, fk);
      xmk = pf_scale(xmk, fm);
      if (mul == PF_MUL_IK) { xk = pf_mul_i<+1>(xk); xmk = pf_mul_i<+1>(xmk); }
    }
    v[m] = pf_c2r_pre<F>(xk, xmk, tw[e], e == 0);
  }
  __syncthreads();

  PfStages<F, M, +1, 2>::run(
      v, tl, tw, [&](int pos, C val) { L[pf_lpad(pos)] = val; }, [&](int pos) { return L[pf_lpad(pos)]; });

  const long long row = line0 + l;
  if (row < p.nlines) {
    const F norm = (F)p.norm;
    const F dcv = p.dc ? (F)(*p.dc) : (F)0;
    if (p.job[job].out_f32) {
      float2 *o = reinterpret_cast<float2 *>(reinterpret_cast<float *>(p.job[job].out) + row * (long long)N);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int n2 = tl + m * NT;
        o[n2] = make_float2((float)(v[m].x * norm + dcv), (float)(v[m].y * norm + dcv));
      }
    } else {
      C *o = reinterpret_cast<C *>(reinterpret_cast<F *>(p.job[job].out) + row * p.out_pitch);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int n2 = tl + m * NT;
        o[n2] = pf_mk<F>(v[m].x * norm + dcv, v[m].y * norm + dcv);
      }
    }
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
    for (int m = 0; m < 8; m++) v[m] = valid ? in[tl + m * NT] : pf_mk<F>(0, 0);
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
      out[k] = pf_r2c_post<F>(zk, zmk, tw[k]);
      if (k == 0) out[M] = pf_mk<F>(zk.x - zk.y, (F)0);
    }
  }
}

// ------------------------------------------------------------ dispatch ----

// columns per workgroup of the strided passes: 128-byte row segments where LDS allows
template <typename F, int N> struct PfTileCols {
  static constexpr int lds_budget = 64 * 1024;
  static constexpr int t0 = 128 / (2 * (int)sizeof(F));  // 8 (fp64) or 16 (fp32)
  static constexpr int fit = lds_budget / (N * 2 * (int)sizeof(F));
  static constexpr int thr = 8192 / N;  // T*N/8 <= 1024
  static constexpr int a = t0 < fit ? t0 : fit;
  static constexpr int b = a < thr ? a : thr;
  static constexpr int value = b < 1 ? 1 : b;
};

template <typename F, int N, int DIR>
static int launch_strided_n(const PfStridedParams &p, hipStream_t st) {
  constexpr int T = PfTileCols<F, N>::value;
  const int ntiles = (p.ncols + T - 1) / T;
  dim3 grid((unsigned)(ntiles * p.njobs), (unsigned)p.nouter, 1), block(T * N / 8, 1, 1);
  const size_t shm = (size_t)N * T * sizeof(pfc<F>);
  hipLaunchKernelGGL((k_strided<F, N, T, DIR>), grid, block, shm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <typename F, int N>
static int launch_c2r_n(const PfC2RParams &p, hipStream_t st) {
  constexpr int M = N / 2, NT = M / 8;
  constexpr int TL = (NT >= 256) ? 1 : 256 / NT;
  constexpr int LPL = (M + 1 > M + M / 8) ? M + 1 : M + M / 8;
  const long long nblk = (p.nlines + TL - 1) / TL;
  dim3 grid((unsigned)(nblk * p.njobs), 1, 1), block(TL * NT, 1, 1);
  const size_t shm = (size_t)TL * LPL * sizeof(pfc<F>);
  hipLaunchKernelGGL((k_c2r<F, N, TL>), grid, block, shm, st, p);
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
#define CALL(NN) launch_strided_n<float, NN, +1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    } else {
#define CALL(NN) launch_strided_n<float, NN, -1>(p, st)
      PF_SWITCH_N(n, CALL)
#undef CALL
    }
  }
}

int pf_launch_c2r(int fb, int n, const PfC2RParams &p, hipStream_t st) {
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

int pf_launch_r2c(int fb, int n, const PfR2CParams &p, hipStream_t st) {
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
