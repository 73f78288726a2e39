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
// The partner's value of the paired first stage: the thread T = 8 lanes away, i.e. the other half of the same row of sixteen
// lanes -- a rotation by eight within the row, one v_mov_b32 with a DPP modifier per 32-bit word (no LDS, no barrier).
__device__ __forceinline__ int pf_lane_xor8(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x128 /* row_ror:8 */, 0xf, 0xf, false); }
__device__ __forceinline__ double pf_lane_xor8(double x) {
  return __hiloint2double(pf_lane_xor8(__double2hiint(x)), pf_lane_xor8(__double2loint(x)));
}
__device__ __forceinline__ float pf_lane_xor8(float x) { return __int_as_float(pf_lane_xor8(__float_as_int(x))); }
__device__ __forceinline__ pf_f32x2 pf_lane_xor8(pf_f32x2 x) { pf_f32x2 r; r.x = pf_lane_xor8(x.x); r.y = pf_lane_xor8(x.y); return r; }
template <typename F> __device__ __forceinline__ pfc<F> pf_lane_xor8(pfc<F> a) { return pf_mk<F>(pf_lane_xor8(a.x), pf_lane_xor8(a.y)); }

// P16: the paired plan of pf_fft_core.h (callers: N = 16 * 8^k with the pair eight lanes apart, i.e. tiles of eight columns)
template <typename F, int N, int DIR, int TWS, int S = 0, bool WAVE_LOCAL = false, bool P16 = false>
struct PfStages {
  template <typename WR, typename RD>
  static __device__ __forceinline__ void exchange(pfc<F> (&v)[8], int tl, WR wr, RD rd) {
    constexpr int NT = N / 8;
#pragma unroll
    for (int m = 0; m < 8; m++) wr(pf_stage_pos<N, S, P16>(tl, m), v[m]);
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
    if constexpr (S == 0 && P16) {
      pf_pair16_local<DIR>(v, tl);
      typedef typename pf_lane<F>::type SC;
      const F sgn = (F)((tl & 1) ? (SC)-1 : (SC)1);
#pragma unroll
      for (int m = 0; m < 8; m++) v[m] = pf_pair16_combine1(v[m], pf_lane_xor8(v[m]), sgn);
    } else if constexpr (S == 0) pf_stage<F, N, 0, DIR, TWS>(v, tt, tw);
    if constexpr (S + 1 < pf_nstages(N, P16)) {
      pfc<F> w[8 / pf_radix(N, S + 1, P16)];
      pf_stage_twiddles<F, N, S + 1, DIR, TWS, 0, P16>(tt, tw, w);
      exchange(v, tl, wr, rd);
      pf_stage_apply<F, N, S + 1, DIR, 0, P16>(v, w);
      PfStages<F, N, DIR, TWS, S + 1, WAVE_LOCAL, P16>::run(v, tl, tw, wr, rd, tl_tw);
    }
  }
};

// 1/N^3 normalisation plus the DC mode, one definition for every z-pass so that they round alike
template <typename F> __device__ __forceinline__ F pf_norm_dc(F v, F norm, F dc) { return v * norm + dc; }
// ... on both reals of a complex (one fma per real either way: the same bits; fp32: ONE packed instruction)
template <typename F> __device__ __forceinline__ pfc<F> pf_norm_dc2(pfc<F> v, F norm, F dc) {
#if PF_PK_DEV
  if constexpr (pf_is_f32<F>::value) return pf_unpk<F>(PfCxPk::fma1(pf_pk(v), (float)norm, (float)dc));
#endif
  return pf_mk<F>(pf_norm_dc(v.x, norm, dc), pf_norm_dc(v.y, norm, dc));
}

// kz factor + Hermitian fold of one element pair (k, M-k) of a half-spectrum row (see pf_c2r_pre)
template <typename F>
__device__ __forceinline__ pfc<F> pf_zfold(pfc<F> xk, pfc<F> xmk, int e, int M, int mul, F kf, pfc<F> wk, bool maybe0 = true) {
  if (mul != 0 /* PF_MUL_ONE */) {
    F fk = kf * (F)e, fm = kf * (F)(M - e);
    if (mul == 2 /* PF_MUL_K2 */) { fk *= fk; fm *= fm; }
    xk = pf_scale(xk, fk);
    xmk = pf_scale(xmk, fm);
    if (mul == 3 /* PF_MUL_IK */) { xk = pf_mul_i<+1>(xk); xmk = pf_mul_i<+1>(xmk); }
  }
  // maybe0: false where the caller knows e > 0 (register m > 0 of a thread: e = tl + m NT) -- the k = 0 form then leaves no trace in the code
  // (fp64 callers keep the test on every element: their kernels sit at register limits -- 80 for the invariant z-pass of 1024 points,
  //  94 -> 108 for the mixed-radix one -- that the shorter code, scheduled differently, breaks: measured, round 6)
  return pf_c2r_pre<F>(xk, xmk, wk, (maybe0 || sizeof(F) == 8) && e == 0);
}
