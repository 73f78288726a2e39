// pf_fft_stages.h -- device-side driver of the register/LDS Stockham stages (pf_fft_core.h)
#pragma once
#include "pf_fft_core.h"

// Exchange through LDS between the lanes of ONE wave: the LDS queue serves a wave's accesses in order, so no instruction is
// needed -- but the compiler must neither move LDS accesses across this point nor assume other lanes' words unchanged: a
// release / acquire fence pair at wavefront scope around the scheduling barrier says exactly that and emits nothing.
__device__ __forceinline__ void pf_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WAVE_LOCAL: the N / 8 threads of a transform sit in one wave and exchange through LDS words nobody else touches: the
// LDS queue serves a wave's accesses in order, so the exchange needs no workgroup barrier.
// Twiddles: the table values of stage S+1 are requested BEFORE the LDS exchange that follows stage S, so their L1 latency
// passes during the exchange instead of after it (x-pass 6.4 -> 6.0 ms per Hessian launch at 1024^3; fetching them once
// outside the row loop of the invariant z-pass, 160 VGPRs, changed nothing there: profiles/r02_experiments.md).
template <typename F, int N, int DIR, int TWS, int S = 0, bool WAVE_LOCAL = false>
struct PfStages {
  template <typename WR, typename RD>
  static __device__ __forceinline__ void exchange(pfc<F> (&v)[8], int tl, WR wr, RD rd) {
    constexpr int NT = N / 8;
#pragma unroll
    for (int m = 0; m < 8; m++) wr(pf_stage_pos<N, S>(tl, m), v[m]);
    if constexpr (WAVE_LOCAL) pf_wave_sync(); else __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = rd(tl + m * NT);
    if constexpr (WAVE_LOCAL) pf_wave_sync(); else __syncthreads();
  }
  // entry (S = 0): stage 0 has no twiddles
  // tl_tw: the same thread index, as the twiddle side sees it.  A row loop that hides its index from the optimiser (so that
  // addresses are not hoisted into ~250 registers) may pass the plain, loop-invariant one here: the table values AND their
  // powers w^2 .. w^7 are then formed once, outside the loop (48 registers for a 512-point line).
  template <typename WR, typename RD>
  static __device__ __forceinline__ void run(pfc<F> (&v)[8], int tl, const pfc<typename pf_lane<F>::type> *__restrict__ tw, WR wr, RD rd, int tl_tw = -1) {
    static_assert(!WAVE_LOCAL || N / 8 <= 64, "a wave-local transform has at most 64 threads");
    const int tt = tl_tw < 0 ? tl : tl_tw;
    if constexpr (S == 0) pf_stage<F, N, 0, DIR, TWS>(v, tt, tw);
    if constexpr (S + 1 < pf_nstages(N)) {
      pfc<F> w[8 / pf_radix(N, S + 1)];
      pf_stage_twiddles<F, N, S + 1, DIR, TWS>(tt, tw, w);
      exchange(v, tl, wr, rd);
      pf_stage_apply<F, N, S + 1, DIR>(v, w);
      PfStages<F, N, DIR, TWS, S + 1, WAVE_LOCAL>::run(v, tl, tw, wr, rd, tl_tw);
    }
  }
};

// 1/N^3 normalisation plus the DC mode, one definition for every z-pass so that they round alike
template <typename F> __device__ __forceinline__ F pf_norm_dc(F v, F norm, F dc) { return v * norm + dc; }

// kz factor + Hermitian fold of one element pair (k, M-k) of a half-spectrum row (see pf_c2r_pre)
template <typename F>
__device__ __forceinline__ pfc<F> pf_zfold(pfc<F> xk, pfc<F> xmk, int e, int M, int mul, F kf, pfc<F> wk) {
  if (mul != 0 /* PF_MUL_ONE */) {
    F fk = kf * (F)e, fm = kf * (F)(M - e);
    if (mul == 2 /* PF_MUL_K2 */) { fk *= fk; fm *= fm; }
    xk = pf_scale(xk, fk);
    xmk = pf_scale(xmk, fm);
    if (mul == 3 /* PF_MUL_IK */) { xk = pf_mul_i<+1>(xk); xmk = pf_mul_i<+1>(xmk); }
  }
  return pf_c2r_pre<F>(xk, xmk, wk, e == 0);
}
