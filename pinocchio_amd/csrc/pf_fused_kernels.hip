// pf_fused_kernels.hip -- z-pass of the six second derivatives fused with the
// collapse-time solve (gfx950).
//
// k_zcollapse: a persistent workgroup takes L lines (x,y) at a time: the six
// Hermitian half-rows are folded (kz factor), inverse transformed by a
// half-length complex FFT (one wave-sized thread group per row, exchange in the
// row's own LDS slot), normalised (1/N^3, src/fmax-pfft.c:220-225) and left as
// six real rows in LDS; then every thread solves the ellipsoidal collapse for
// its cells straight from LDS and updates Fmax/Rmax (src/collapse_times.c:545-591).
// The Hessian never goes to HBM (except on request, for the R=0 radius the LPT
// sources need): per radius this removes 6 field writes + 6 field reads, and the
// HBM reads of the rows hide under the fp64-ALU-bound solve of the co-resident
// workgroups.  Same arithmetic, in the same order, as k_c2r + k_collapse.
#include "pf_internal.h"
#include "pf_fft_stages.h"   // FFT helpers are defined before the no-contraction pragma of the solver header
#include "pf_collapse_core.h"

#define PF_FUSED_THREADS 256

// The solve is fp64-ALU bound and wants ILP (~165 VGPRs when inlined); the kernel is bounded to 3 waves
// per SIMD (168 VGPRs), which matches the 3 workgroups per CU that the LDS rows allow.
__device__ __forceinline__ double pf_solve_cell(double d0, double d1, double d2, double d3, double d4, double d5,
                                                 const pf_spline_view &sv) {
  const double d[6] = {d0, d1, d2, d3, d4, d5};
  double lam[3];
  return pf_inverse_collapse_time<false>(d, sv, lam);
}

template <typename F, int N> struct PfFusedGeom {
  static constexpr int M = N / 2, NT = M / 8;               // threads per row transform
  static constexpr int G = PF_FUSED_THREADS / NT;           // row transforms in flight per round
  static constexpr int lds_rows = 49152 / (6 * (N + 2) * (int)sizeof(F));
  static constexpr int by_groups = G / 6;
  static constexpr int l0 = by_groups < lds_rows ? by_groups : lds_rows;
  static constexpr int L = l0 < 1 ? 1 : l0;                 // lines per iteration
  static constexpr int ROUNDS = (6 * L + G - 1) / G;
};

template <typename F, int N>
__global__ void __launch_bounds__(PF_FUSED_THREADS, 3) k_zcollapse(const PfFusedParams p) {
  using C = pfc<F>;
  using GEO = PfFusedGeom<F, N>;
  constexpr int M = GEO::M, NT = GEO::NT, G = GEO::G, L = GEO::L, ROUNDS = GEO::ROUNDS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *Hc = reinterpret_cast<C *>(smem);                                   // [6*L] rows of M+1 complex (half-spectrum in, N reals out)
  double *sk = reinterpret_cast<double *>(smem + (size_t)6 * L * (M + 1) * sizeof(C));  // spline x,y,c,b,d : 5*nk
  double *red = sk + 5 * p.spline.n;                                     // 2 * nwaves
  const int tid = threadIdx.x;
  const int nk = p.spline.n;
  for (int i = tid; i < nk; i += PF_FUSED_THREADS) {
    sk[i] = p.spline.x[i];
    sk[nk + i] = p.spline.y[i];
    sk[2 * nk + i] = p.spline.c[i];
    sk[3 * nk + i] = p.spline.b[i];
    sk[4 * nk + i] = p.spline.d[i];
  }
  pf_spline_view sv;
  sv.x = sk; sv.y = sk + nk; sv.c = sk + 2 * nk; sv.b = sk + 3 * nk; sv.d = sk + 4 * nk; sv.n = nk;

  const C *__restrict__ tw = reinterpret_cast<const C *>(p.tw);
  const int g = tid / NT, tl0 = tid % NT;
  const F kf = (F)(2.0 * 3.14159265358979323846 / (double)N);
  const F norm = (F)p.norm;
  const F dcv = p.dc ? (F)(*p.dc) : (F)0;
  double sum = 0.0, sum2 = 0.0;
  __syncthreads();

  // Stagger the workgroups that share a CU: they start together and every iteration takes the same
  // time, so without a skew all of them read HBM at once and then all of them solve at once.
  if (p.skew_ns > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    const unsigned long long wait = (unsigned long long)((blockIdx.x / p.ncu) % 3) * (unsigned long long)p.skew_ns / 10ull;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }

  for (long long line0 = (long long)blockIdx.x * L; line0 < p.nlines; line0 += (long long)gridDim.x * L) {
    // ---- all 6*L Hermitian half-rows -> LDS, one coalesced burst ----
    if (p.debug_skip != 2) {
#pragma unroll 1
      for (int idx = tid; idx < 6 * L * (M + 1); idx += PF_FUSED_THREADS) {
        const int cl = idx / (M + 1), k = idx - cl * (M + 1);
        const int comp = cl / L, ll = cl - comp * L;
        const long long row = line0 + ll;
        Hc[idx] = row < p.nlines ? reinterpret_cast<const C *>(p.in[comp])[row * p.in_pitch + k] : pf_mk<F>(0, 0);
      }
    }
    __syncthreads();
    // ---- six c2r row transforms per line, G rows at a time, in place in the row's LDS slot ----
#pragma unroll 1
    for (int r = 0; r < (p.debug_skip == 2 ? 0 : ROUNDS); r++) {
      // opaque copies: keep twiddle loads and index math inside the loops instead of hoisted into ~40 VGPRs
      int cl = r * G + g, tl = tl0;             // component-line: comp = cl / L, line = cl % L
      asm volatile("" : "+v"(cl), "+v"(tl));
      const bool active = cl < 6 * L;
      const int comp = active ? cl / L : 0;
      C *Lrow = Hc + (size_t)(active ? cl : 0) * (M + 1);
      const int mul = p.mul[comp];
      C v[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int e = tl + m * NT;
        v[m] = pf_zfold<F>(Lrow[e], Lrow[M - e], e, M, mul, kf, tw[e]);
      }
      __syncthreads();  // every fold read of a row precedes the exchange writes into it
      PfStages<F, M, +1, 2>::run(
          v, tl, tw, [&](int pos, C val) { if (active) Lrow[pos] = val; }, [&](int pos) { return Lrow[pos]; });
      if (active) {
#pragma unroll
        for (int m = 0; m < 8; m++)
          Lrow[tl + m * NT] = pf_mk<F>(pf_norm_dc(v[m].x, norm, dcv), pf_norm_dc(v[m].y, norm, dcv));
      }
      __syncthreads();
    }

    // ---- optional copy of the Hessian rows to HBM (R = 0: needed by the LPT sources) ----
    if (p.write_h) {
#pragma unroll 1
      for (int idx = tid; idx < 6 * L * (N / 2); idx += PF_FUSED_THREADS) {
        const int cl = idx / (N / 2), n2 = idx - cl * (N / 2);
        const int comp = cl / L, ll = cl - comp * L;
        const long long row = line0 + ll;
        if (row < p.nlines)
          reinterpret_cast<C *>(reinterpret_cast<F *>(p.out[comp]) + row * p.out_pitch)[n2] = Hc[(size_t)cl * (M + 1) + n2];
      }
    }

    // ---- collapse solve on the L*N cells of these lines (one cell at a time: the solve alone needs ~165 VGPRs) ----
#pragma unroll 1
    for (int idx = tid; idx < (p.debug_skip == 1 ? 0 : L * N); idx += PF_FUSED_THREADS) {
      const int ll = idx / N, z = idx - ll * N;
      const long long row = line0 + ll;
      if (row >= p.nlines) continue;
      double d[6];
#pragma unroll
      for (int c6 = 0; c6 < 6; c6++) d[c6] = (double)reinterpret_cast<const F *>(Hc + (size_t)(c6 * L + ll) * (M + 1))[z];
      const double delta = d[0] + d[1] + d[2];
      sum += delta;
      sum2 += delta * delta;
      const double Fnew = pf_solve_cell(d[0], d[1], d[2], d[3], d[4], d[5], sv);
      const long long i = row * N + z;
      const float fold = p.ismooth ? p.fmax[i] : -10.0f;
      if ((double)fold < Fnew) {
        p.fmax[i] = (float)Fnew;
        p.rmax[i] = p.ismooth;
      } else if (!p.ismooth) {
        p.fmax[i] = -10.0f;
        p.rmax[i] = -1;
      }
    }
    __syncthreads();  // rows are overwritten by the next iteration
  }

  // deterministic block reduction (same scheme as k_collapse)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    sum2 += __shfl_down(sum2, off, 64);
  }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = sum; red[2 * (tid >> 6) + 1] = sum2; }
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0;
    for (int i = 0; i < PF_FUSED_THREADS / 64; i++) { a += red[2 * i]; b += red[2 * i + 1]; }
    p.partials[2 * blockIdx.x] = a;
    p.partials[2 * blockIdx.x + 1] = b;
  }
}

template <typename F, int N>
static int launch_fused_n(const PfFusedParams &p, int blocks_per_cu, int ncu, hipStream_t st, int *nblocks_out) {
  using GEO = PfFusedGeom<F, N>;
  const size_t shm = (size_t)6 * GEO::L * (N / 2 + 1) * sizeof(pfc<F>) + (size_t)5 * p.spline.n * sizeof(double) + 2 * (PF_FUSED_THREADS / 64) * sizeof(double);
  if (shm > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_zcollapse<F, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 3;
  }
  long long nb = (p.nlines + GEO::L - 1) / GEO::L;
  const long long cap = (long long)blocks_per_cu * ncu;
  if (nb > cap) nb = cap;
  if (nb > p.max_blocks) nb = p.max_blocks;
  *nblocks_out = (int)nb;
  hipLaunchKernelGGL((k_zcollapse<F, N>), dim3((unsigned)nb), dim3(PF_FUSED_THREADS), shm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int pf_launch_zcollapse(int fb, int n, const PfFusedParams &p, int blocks_per_cu, int ncu, hipStream_t st, int *nblocks_out) {
#define CALLD(NN) launch_fused_n<double, NN>(p, blocks_per_cu, ncu, st, nblocks_out)
#define CALLF(NN) launch_fused_n<float, NN>(p, blocks_per_cu, ncu, st, nblocks_out)
  if (fb == 8) {
    switch (n) {
      case 16: return CALLD(16); case 32: return CALLD(32); case 64: return CALLD(64); case 128: return CALLD(128);
      case 256: return CALLD(256); case 512: return CALLD(512); case 1024: return CALLD(1024); case 2048: return CALLD(2048);
      default: return 2;
    }
  } else {
    switch (n) {
      case 16: return CALLF(16); case 32: return CALLF(32); case 64: return CALLF(64); case 128: return CALLF(128);
      case 256: return CALLF(256); case 512: return CALLF(512); case 1024: return CALLF(1024); case 2048: return CALLF(2048);
      default: return 2;
    }
  }
#undef CALLD
#undef CALLF
}
