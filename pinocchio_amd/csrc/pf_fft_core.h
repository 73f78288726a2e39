// pf_fft_core.h -- register-resident Stockham FFT building blocks (gfx950).
//
// One thread owns 8 complex points of a length-N line (points tl + m*N/8,
// m = 0..7, tl = thread index in the line, N/8 threads per line).  A stage does
// 8/R radix-R butterflies per thread in registers (R = 8, or 2/4 for the last
// stage), then the line is exchanged through LDS: write at pf_stage_pos(),
// barrier, read back at tl + m*N/8.  The first stage reads HBM directly and the
// last stage writes HBM directly, so a length-1024 line crosses LDS 3 times and
// HBM once each way.  Twiddles come from one table exp(+2 pi i j / NTAB).
//
// The same header is compiled by g++ in tests/test_fft_core.py (a serial loop
// over tl stands in for the wavefront) to unit-test the index algebra without a
// GPU; that is a test of this header, not a CPU path of the library.
#pragma once

#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif
#include "pf_cxpk.h"
// PF_PK_F32: on the device every operation on pfc<float> below goes through the packed algebra of pf_cxpk.h (round 6: the fp32
// z-passes, the fp32 strided passes of lines below 1024 points and the fp32 mixed-radix passes were written in scalar complex
// arithmetic that the compiler packed only in part, paying a v_mov / v_xor per swapped or negated operand).  -DPF_PK_F32=0: the
// plain expressions (A/B builds; the host always takes them).
#ifndef PF_PK_F32
#define PF_PK_F32 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PF_PK_F32
#define PF_PK_DEV 1
#else
#define PF_PK_DEV 0
#endif

template <typename F>
struct alignas(2 * sizeof(F)) pfc {
  F x, y;
};

template <typename F> PF_HD pfc<F> pf_mk(F x, F y) { pfc<F> r; r.x = x; r.y = y; return r; }
template <typename F> struct pf_is_f32 { static constexpr bool value = false; };
template <> struct pf_is_f32<float> { static constexpr bool value = true; };
#if PF_PK_DEV
// pfc<float> <-> the register pair of the packed algebra: the same eight bytes
template <typename F> __device__ __forceinline__ pf_f2 pf_pk(pfc<F> a) { pf_f2 r; r.x = (float)a.x; r.y = (float)a.y; return r; }
template <typename F> __device__ __forceinline__ pfc<F> pf_unpk(pf_f2 a) { return pf_mk<F>((F)a.x, (F)a.y); }
#define PF_IF_PK(F, expr) if constexpr (pf_is_f32<F>::value) return pf_unpk<F>(expr);
#else
#define PF_IF_PK(F, expr)
#endif
template <typename F> PF_HD pfc<F> operator+(pfc<F> a, pfc<F> b) { PF_IF_PK(F, pf_pk(a) + pf_pk(b)) return pf_mk<F>(a.x + b.x, a.y + b.y); }
template <typename F> PF_HD pfc<F> operator-(pfc<F> a, pfc<F> b) { PF_IF_PK(F, pf_pk(a) - pf_pk(b)) return pf_mk<F>(a.x - b.x, a.y - b.y); }
template <typename F> PF_HD pfc<F> pf_cmul(pfc<F> a, pfc<F> b) {
  PF_IF_PK(F, PfCxPk::cmul<+1>(pf_pk(a), pf_pk(b)))
  return pf_mk<F>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
template <typename F> PF_HD pfc<F> pf_cmulc(pfc<F> a, pfc<F> b) {
  PF_IF_PK(F, PfCxPk::cmul<-1>(pf_pk(a), pf_pk(b)))
  return pf_mk<F>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
template <typename F> PF_HD pfc<F> pf_scale(pfc<F> a, F s) { PF_IF_PK(F, pf_pk(a) * (float)s) return pf_mk<F>(a.x * s, a.y * s); }
template <typename F> PF_HD pfc<F> pf_conj(pfc<F> a) { return pf_mk<F>(a.x, -a.y); }
// multiply by (DIR * i): DIR=+1 -> (-y, x); DIR=-1 -> (y, -x)
template <int DIR, typename F> PF_HD pfc<F> pf_mul_i(pfc<F> a) {
  PF_IF_PK(F, PfCxPk::muli<DIR>(pf_pk(a)))
  return DIR > 0 ? pf_mk<F>(-a.y, a.x) : pf_mk<F>(a.y, -a.x);
}
// a + DIR i b, a - DIR i b, a + conj(b), a - conj(b): one instruction each in the packed algebra
template <int DIR, typename F> PF_HD pfc<F> pf_addi(pfc<F> a, pfc<F> b) { PF_IF_PK(F, PfCxPk::addi<DIR>(pf_pk(a), pf_pk(b))) return a + pf_mul_i<DIR>(b); }
template <int DIR, typename F> PF_HD pfc<F> pf_subi(pfc<F> a, pfc<F> b) { PF_IF_PK(F, PfCxPk::subi<DIR>(pf_pk(a), pf_pk(b))) return a - pf_mul_i<DIR>(b); }
template <typename F> PF_HD pfc<F> pf_addc(pfc<F> a, pfc<F> b) { PF_IF_PK(F, PfCxPk::addc(pf_pk(a), pf_pk(b))) return a + pf_conj(b); }
template <typename F> PF_HD pfc<F> pf_subc(pfc<F> a, pfc<F> b) { PF_IF_PK(F, PfCxPk::subc(pf_pk(a), pf_pk(b))) return a - pf_conj(b); }

// The scalar behind a field type: itself, or the lane type of a two-lane vector.  pfc<pf_f32x2> is TWO neighbouring fp32
// columns carried by one thread -- x = (re of column 0, re of column 1), y = the imaginary parts -- so that every operation
// of the transforms below is one packed instruction for both (v_pk_fma_f32 and friends: the fp32 rate of the chip is quoted
// on them), twiddles and k factors are shared, and a thread's loads, stores and LDS exchanges move 16 bytes as in the fp64
// kernels.  Lane by lane the arithmetic is that of pfc<float>.
#if defined(__HIPCC__)
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
#endif
template <typename F> struct pf_lane { typedef F type; static constexpr int n = 1; };
#if defined(__HIPCC__)
template <> struct pf_lane<pf_f32x2> { typedef float type; static constexpr int n = 2; };
#endif
template <typename F> PF_HD pfc<F> pf_bcast(pfc<typename pf_lane<F>::type> a) { return pf_mk<F>((F)a.x, (F)a.y); }
template <typename F> PF_HD pfc<F> pf_zero() { return pf_mk<F>((F)0, (F)0); }

// streaming (non-temporal) access to once-touched field data.  On since the strided passes move whole 128-byte lines (128 KB tiles):
// three interleaved runs each on one box, x-pass 5.96 -> 5.45, y-pass 11.26 -> 10.85, z-passes -0.2..-0.4 ms per launch, 827.6 (826..830) ->
// 808.2 (805..811) ms per step.  With 64-byte segments it cost the y-pass 13 % (round 1): half lines no longer merged in L2.
// -DPF_NT=0 builds the plain accesses.
#ifndef PF_NT
#define PF_NT 1
#endif
#if defined(__HIPCC__)
template <typename F> struct pf_vec2 { typedef F type __attribute__((ext_vector_type(2))); };
template <typename F> __device__ __forceinline__ pfc<F> pf_ld_stream(const pfc<F> *p) {
#if PF_NT
  const typename pf_vec2<F>::type v = __builtin_nontemporal_load(reinterpret_cast<const typename pf_vec2<F>::type *>(p));
  return pf_mk<F>(v.x, v.y);
#else
  return *p;
#endif
}
template <typename F> __device__ __forceinline__ void pf_st_stream(pfc<F> *p, pfc<F> v) {
#if PF_NT
  typename pf_vec2<F>::type t;
  t.x = v.x; t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<typename pf_vec2<F>::type *>(p));
#else
  *p = v;
#endif
}
#endif

constexpr int pf_ilog2(int n) { return n <= 1 ? 0 : 1 + pf_ilog2(n >> 1); }
// stage plan for 8 points per thread: radix 8 as long as it fits, the remainder (2 or 4) last.
// p16 -- the PAIRED plan for N = 16 * 8^k, k >= 1 (128, 1024: lengths whose plain plan ends on a radix-2 stage): the first stage
// is a radix-16 butterfly done by two neighbouring threads (tl, tl ^ 1) -- each its own radix 8, then one radix-2 step across
// the pair through the lanes of the wave (pf_pair16_local / pf_pair16_combine) -- followed by radix-8 stages only: one LDS
// exchange (eight 16-byte writes, eight reads and two barriers per thread) less than the plain plan; a 1024-point line
// crosses LDS twice instead of three times.
constexpr bool pf_pair16_length(int n) { return n >= 128 && pf_ilog2(n) % 3 == 1; }  // (16 itself would end on the paired stage, whose outputs are not in line order)
constexpr int pf_nstages(int n, bool p16 = false) { return p16 ? (pf_ilog2(n) - 4) / 3 + 1 : (pf_ilog2(n) + 2) / 3; }
constexpr int pf_radix(int n, int s, bool p16 = false) {
  return p16 ? (s == 0 ? 16 : 8) : (s < pf_ilog2(n) / 3 ? 8 : (1 << (pf_ilog2(n) - 3 * (pf_ilog2(n) / 3))));
}
constexpr int pf_ns(int n, int s, bool p16 = false) { return s == 0 ? 1 : pf_ns(n, s - 1, p16) * pf_radix(n, s - 1, p16); }
// the point of the line that register m of thread tl holds BEFORE the first stage (what a kernel loads there)
template <int N, bool P16 = false> PF_HD int pf_line_index(int tl, int m) {
  return P16 ? (tl >> 1) + (N / 16) * (tl & 1) + m * (N / 8) : tl + m * (N / 8);
}

// X_k = sum_t u_t w^{kt}, w = exp(DIR 2 pi i / R), natural order in and out
template <int DIR, typename F> PF_HD void pf_bfly2(pfc<F> &a, pfc<F> &b) {
  pfc<F> t = a - b;
  a = a + b;
  b = t;
}
template <int DIR, typename F> PF_HD void pf_bfly4(pfc<F> &v0, pfc<F> &v1, pfc<F> &v2, pfc<F> &v3) {
  pfc<F> a = v0 + v2, b = v0 - v2, c = v1 + v3, d = v1 - v3;
  v0 = a + c;
  v1 = pf_addi<DIR>(b, d);   // b + DIR i d (the same bits as adding pf_mul_i(d))
  v2 = a - c;
  v3 = pf_subi<DIR>(b, d);
}
template <int DIR, typename F> PF_HD void pf_bfly8(pfc<F> (&u)[8]) {
  pf_bfly4<DIR>(u[0], u[2], u[4], u[6]);  // E_0..3 in u[0],u[2],u[4],u[6]
  pf_bfly4<DIR>(u[1], u[3], u[5], u[7]);  // O_0..3 in u[1],u[3],u[5],u[7]
  const F h = (F)0.70710678118654752440;
  pfc<F> o1, o3;
#if PF_PK_DEV
  if constexpr (pf_is_f32<F>::value) {  // W8^1 = h (1 + DIR i), W8^3 = h (-1 + DIR i): a product with a constant each
    o1 = pf_unpk<F>(PfCxPk::cmulc<DIR>(pf_pk(u[3]), 0.70710678118654752440, 0.70710678118654752440));
    o3 = pf_unpk<F>(PfCxPk::cmulc<DIR>(pf_pk(u[7]), -0.70710678118654752440, 0.70710678118654752440));
  } else
#endif
  {
    // (DIR = +-1 written out: the same operations, and no int * vector products for the two-lane field type)
    o1 = DIR > 0 ? pf_mk<F>(h * (u[3].x - u[3].y), h * (u[3].y + u[3].x))      // W8^1 = h(1, DIR)
                 : pf_mk<F>(h * (u[3].x + u[3].y), h * (u[3].y - u[3].x));
    o3 = DIR > 0 ? pf_mk<F>(h * (-u[7].x - u[7].y), h * (-u[7].y + u[7].x))    // W8^3 = h(-1, DIR)
                 : pf_mk<F>(h * (-u[7].x + u[7].y), h * (-u[7].y - u[7].x));
  }
  pfc<F> e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o2 = u[5];   // W8^2 = DIR i: folded into the sums of e2
  u[0] = e0 + o0; u[4] = e0 - o0;
  u[1] = e1 + o1; u[5] = e1 - o1;
  u[2] = pf_addi<DIR>(e2, o2); u[6] = pf_subi<DIR>(e2, o2);
  u[3] = e3 + o3; u[7] = e3 - o3;
}

// LDS/HBM position of register m of thread tl AFTER stage S of a length-N line
template <int N, int S, bool P16 = false> PF_HD int pf_stage_pos(int tl, int m) {
  if (P16 && S == 0) return 16 * (tl >> 1) + 8 * (tl & 1) + m;  // output t = m + 8 h of the pair's butterfly u = tl >> 1
  constexpr int R = (P16 && S == 0) ? 8 : pf_radix(N, S, P16), NS = pf_ns(N, S, P16), Q = 8 / R, NT = N / 8;
  const int q = m % Q, t = m / Q;
  const int jb = tl + q * NT;
  const int k = jb & (NS - 1);
  return (jb - k) * R + k + t * NS;
}

// Twiddles of stage S: one table value w^1 per butterfly of the thread (Q = 8/R of them; none in stage 0, where NS = 1).
// They depend on the thread's place in the line only -- not on the line, the job or the row -- so a kernel may fetch them
// wherever their latency is hidden (before the LDS exchange that precedes the stage, or once outside its loop over rows).
constexpr int pf_stage_ntw(int n, int s, bool p16 = false) { return pf_ns(n, s, p16) > 1 ? 8 / pf_radix(n, s, p16) : 0; }
constexpr int pf_ntw_before(int n, int s, bool p16 = false) { return s == 0 ? 0 : pf_ntw_before(n, s - 1, p16) + pf_stage_ntw(n, s - 1, p16); }
constexpr int pf_ntw_total(int n, bool p16 = false) { return pf_ntw_before(n, pf_nstages(n, p16), p16); }

// tw[j] = exp(+2 pi i j / (N*TWS))
// (arrays by reference and compile-time offsets: the values must stay in registers -- through a pointer the compiler puts
//  them in scratch memory, and a scratch reload waits on the same in-order counter as every global load before it)
template <typename F, int N, int S, int DIR, int TWS, int OFF = 0, bool P16 = false, int NW>
PF_HD void pf_stage_twiddles(int tl, const pfc<typename pf_lane<F>::type> *__restrict__ tw, pfc<F> (&w)[NW]) {
  constexpr int R = pf_radix(N, S, P16), NS = pf_ns(N, S, P16), Q = 8 / R, NT = N / 8;
  constexpr int TWM = (N / (NS * R)) * TWS;
  if (NS > 1) {
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int jb = tl + q * NT;
      const int k = jb & (NS - 1);
      w[OFF + q] = pf_bcast<F>(tw[k * TWM]);
      if (DIR < 0) w[OFF + q].y = -w[OFF + q].y;
    }
  }
}

// stage S on the 8 registers of a thread, with its twiddles w[0..Q-1] in hand
template <typename F, int N, int S, int DIR, int OFF = 0, bool P16 = false, int NW>
PF_HD void pf_stage_apply(pfc<F> (&v)[8], const pfc<F> (&w)[NW]) {
  constexpr int R = pf_radix(N, S, P16), NS = pf_ns(N, S, P16), Q = 8 / R;
#pragma unroll
  for (int q = 0; q < Q; q++) {
    if (NS > 1) {
      // one table value per butterfly: w^2..w^(R-1) by products of depth <= 3 (error <= ~4 ulp, far inside the
      // transform's own rounding) instead of R-1 dependent L1 loads
      const pfc<F> w1 = w[OFF + q];
      v[q + 1 * Q] = pf_cmul(v[q + 1 * Q], w1);
      if (R >= 4) {
        const pfc<F> w2 = pf_cmul(w1, w1), w3 = pf_cmul(w2, w1);
        v[q + 2 * Q] = pf_cmul(v[q + 2 * Q], w2);
        v[q + 3 * Q] = pf_cmul(v[q + 3 * Q], w3);
        if (R == 8) {
          const pfc<F> w4 = pf_cmul(w2, w2), w5 = pf_cmul(w4, w1), w6 = pf_cmul(w3, w3), w7 = pf_cmul(w4, w3);
          v[q + 4 * Q] = pf_cmul(v[q + 4 * Q], w4);
          v[q + 5 * Q] = pf_cmul(v[q + 5 * Q], w5);
          v[q + 6 * Q] = pf_cmul(v[q + 6 * Q], w6);
          v[q + 7 * Q] = pf_cmul(v[q + 7 * Q], w7);
        }
      }
    }
    if (R == 8) {
      pf_bfly8<DIR>(v);
    } else if (R == 4) {
      pf_bfly4<DIR>(v[q], v[q + Q], v[q + 2 * Q], v[q + 3 * Q]);
    } else if (R == 2) {
      pf_bfly2<DIR>(v[q], v[q + Q]);
    }
  }
}

// stage S with its twiddles fetched on the spot
template <typename F, int N, int S, int DIR, int TWS>
PF_HD void pf_stage(pfc<F> (&v)[8], int tl, const pfc<typename pf_lane<F>::type> *__restrict__ tw) {
  pfc<F> w[8 / pf_radix(N, S)];
  pf_stage_twiddles<F, N, S, DIR, TWS>(tl, tw, w);
  pf_stage_apply<F, N, S, DIR>(v, w);
}

// ---- the paired first stage (plan p16).  The pair (tl, tl ^ 1) = (h = 0, h = 1) of butterfly u = tl >> 1 holds the sixteen
// inputs x_q = line[u + q N/16], q = 2 m + h (pf_line_index).  X[t] = sum_q x_q W16^(q t), t = ta + 8 tb:
//   each thread: Y_h[ta] = DFT8_m(x_{2m+h})                              (pf_bfly8, natural order)
//   the odd one: Y_1[ta] *= W16^ta                                       (pf_pair16_local)
//   across the pair: X[ta] = Y_0[ta] + Y_1[ta] stays with h = 0, X[ta + 8] = Y_0[ta] - Y_1[ta] with h = 1   (pf_pair16_combine:
//   `o` = the partner's eight values, fetched through the lanes of the wave -- the partner sits T = 8 lanes away)
template <int DIR, typename F> PF_HD void pf_pair16_local(pfc<F> (&v)[8], int tl) {
  pf_bfly8<DIR>(v);
  if (tl & 1) {
    const F c1 = (F)0.92387953251128675613, s1 = (F)0.38268343236508977173, h = (F)0.70710678118654752440;
    // W16^1 = (c1, DIR s1), ^2 = h (1, DIR), ^3 = (s1, DIR c1), ^4 = DIR i, ^5 = (-s1, DIR c1), ^6 = h (-1, DIR), ^7 = (-c1, DIR s1)
    v[1] = DIR > 0 ? pf_mk<F>(c1 * v[1].x - s1 * v[1].y, c1 * v[1].y + s1 * v[1].x) : pf_mk<F>(c1 * v[1].x + s1 * v[1].y, c1 * v[1].y - s1 * v[1].x);
    v[2] = DIR > 0 ? pf_mk<F>(h * (v[2].x - v[2].y), h * (v[2].y + v[2].x)) : pf_mk<F>(h * (v[2].x + v[2].y), h * (v[2].y - v[2].x));
    v[3] = DIR > 0 ? pf_mk<F>(s1 * v[3].x - c1 * v[3].y, s1 * v[3].y + c1 * v[3].x) : pf_mk<F>(s1 * v[3].x + c1 * v[3].y, s1 * v[3].y - c1 * v[3].x);
    v[4] = pf_mul_i<DIR>(v[4]);
    v[5] = DIR > 0 ? pf_mk<F>(-s1 * v[5].x - c1 * v[5].y, -s1 * v[5].y + c1 * v[5].x) : pf_mk<F>(-s1 * v[5].x + c1 * v[5].y, -s1 * v[5].y - c1 * v[5].x);
    v[6] = DIR > 0 ? pf_mk<F>(h * (-v[6].x - v[6].y), h * (-v[6].y + v[6].x)) : pf_mk<F>(h * (-v[6].x + v[6].y), h * (-v[6].y - v[6].x));
    v[7] = DIR > 0 ? pf_mk<F>(-c1 * v[7].x - s1 * v[7].y, -c1 * v[7].y + s1 * v[7].x) : pf_mk<F>(-c1 * v[7].x + s1 * v[7].y, -c1 * v[7].y - s1 * v[7].x);
  }
}
// one value at a time (the kernels: eight partner values held at once would not fit beside src and v): sgn = -1 for the odd
// thread, +1 for the even one; o + sgn v is v + o or o - v, rounded once either way
template <typename F> PF_HD pfc<F> pf_pair16_combine1(pfc<F> v, pfc<F> o, F sgn) { return pf_mk<F>(o.x + sgn * v.x, o.y + sgn * v.y); }
template <typename F> PF_HD void pf_pair16_combine(pfc<F> (&v)[8], const pfc<F> (&o)[8], int tl) {
  const F sgn = (tl & 1) ? (F)-1 : (F)1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int m = 0; m < 8; m++) v[m] = pf_pair16_combine1(v[m], o[m], sgn);
}

// LDS padding for contiguous-line kernels: breaks the stride-8 write pattern of stage 0
PF_HD int pf_lpad(int p) { return p + (p >> 3); }

// ---- real <-> half-complex glue (length-N real line as a length-M = N/2 complex FFT) ----
// c2r, unnormalised: x[n] = sum_j Xt[j] e^{+2 pi i j n / N}, Xt the Hermitian extension of
// X[0..M]; Im X[0], Im X[M] ignored (FFTW/pocketfft semantics).  With
//   Zp[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) e^{+2 pi i k / N},  k = 0..M-1
// the inverse M-point FFT gives x[2n] + i x[2n+1].
template <typename F> PF_HD pfc<F> pf_c2r_pre(pfc<F> xk, pfc<F> xmk, pfc<F> wk /* e^{+2 pi i k/N} */, bool k0) {
  if (k0) {  // k = 0: only the real parts of X[0] and X[M]
    return pf_mk<F>(xk.x + xmk.x, xk.x - xmk.x);
  }
  pfc<F> s = pf_addc(xk, xmk), d = pf_cmul(pf_subc(xk, xmk), wk);   // xk +- conj(xmk)
  return pf_addi<+1>(s, d);                                          // (s.x - d.y, s.y + d.x)
}
// r2c, unnormalised, from Z = FFT_M(x[2n] + i x[2n+1]) (forward sign):
//   X[k] = E + e^{-2 pi i k / N} O,  E = (Z[k] + conj Z[M-k])/2,  O = (Z[k] - conj Z[M-k])/(2i)
template <typename F> PF_HD pfc<F> pf_r2c_post(pfc<F> zk, pfc<F> zmk, pfc<F> wk /* e^{+2 pi i k/N} */) {
  pfc<F> e = pf_scale(pf_addc(zk, zmk), (F)0.5);
  pfc<F> d = pf_scale(pf_subc(zk, zmk), (F)0.5);
#if PF_PK_DEV
  if constexpr (pf_is_f32<F>::value) return pf_subi<+1>(e, pf_cmulc(d, wk));   // e + (d / i) conj(w) = e - i (d conj(w))
#endif
  pfc<F> o = pf_mk<F>(d.y, -d.x);  // d / i
  return e + pf_cmul(o, pf_conj(wk));
}
