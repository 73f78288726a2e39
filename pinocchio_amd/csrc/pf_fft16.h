// pf_fft16.h -- SIXTEEN points of a line per thread: the plan 2048 = 16 x 16 x 8 of the strided passes (round 5), and the
// complex algebra its fp32 form is written in.
//
// A 2048-point line of a tile that is eight 16-byte elements wide (128-byte row segments: two fp32 columns per thread) is
// 256 KB -- more than the LDS of a CU, and with eight points per thread more rows than a workgroup has threads for.  So a
// thread owns sixteen points (the tile lives in the registers: 64 per thread), the stages are radix 16, 16, 8 -- two
// exchanges, as for 1024 points -- and LDS only carries the exchanges: the first one in two halves of 128 KB, the second one
// inside each wave.  Same Stockham algebra as pf_fft_core.h (stage S: inputs data[j + r N/R], twiddle w^(r k), k = j mod NS,
// outputs at (j - k) R + k + r NS):
//
//   thread u = 0..127 (of one column pair), register m = 0..15 holds   line[u + 128 m]
//   stage 0 (R = 16, NS = 1): butterfly u over m; output t = 0..15 is position 16 u + t
//   exchange 1: the 128-point problems of different t are independent: wave w (sixteen of them, eight threads u0 = 0..7 of each
//               of the eight column pairs) takes t = w and reads u = u0 + 8 r, r = 0..15.  ONE column of the pairs at a time (all
//               128 KB of LDS hold that column of the whole tile): every thread writes its sixteen outputs, a barrier, and reads
//               the sixteen inputs of its next butterfly into the same registers; then the other column.  Slot of (t, u): t 128 + u,
//               the 128 slots of wave t
//   stage 1 (R = 16, NS = 16): j = 16 u0 + t, k = t = w: twiddle W_256^(w r) -- the same for every lane of a wave (scalar
//               registers); output s = 0..15 is position 256 u0 + w + 16 s
//   exchange 2: inside the wave (its own 128 slots: 64 per column, which no other wave touches between two jobs), s = s8 + 8 b in
//               two halves b = 0, 1: output s of thread u0 goes to register u0 (of half b) of thread s8
//   stage 2 (R = 8, NS = 256), two butterflies b per thread s0: j = w + 16 s0 + 128 b = k: twiddle W_2048^(j u0); output s1 is
//               element j + 256 s1 of the line, in register 8 b + s1
//
// The algebra (PfCxPk): one column's complex number is a (re, im) pair in a 64-bit register pair, exactly as it lies in memory --
// a thread's 16-byte element is two of them, no shuffling on load or store -- and every operation of the butterflies is ONE
// packed instruction: v_pk_add_f32 for sums and differences, the same with its second operand's halves swapped and one of them
// negated (op_sel / neg modifiers) for a +- i b, v_pk_mul_f32 + v_pk_fma_f32 for a product with a twiddle.  A packed fp32
// instruction costs what one fp64 or one scalar fp32 fma costs (profiles/r05_notes.md: valu_probe), so a column pair is
// transformed for the instructions of one fp64 column.  The compiler folds a swap into op_sel but not a swap with a negated
// half (it builds the operand with v_xor + v_mov): those forms are inline assembly.  PfCxStd is the same algebra on
// std::complex-like doubles (the host test of the index algebra, tests/test_fft_core.py).
#pragma once
#include "pf_fft_core.h"   // (brings PfCxPk: pf_cxpk.h)

// the same algebra on doubles, no tricks (host test of the index algebra)
struct PfCxStd {
  typedef pfc<double> C;
  typedef pfc<double> TW;
  typedef double SC;
  static PF_HD C mk(double re, double im) { return pf_mk<double>(re, im); }
  static PF_HD C add(C a, C b) { return a + b; }
  static PF_HD C sub(C a, C b) { return a - b; }
  template <int DIR> static PF_HD C addi(C a, C b) { return a + pf_mul_i<DIR>(b); }
  template <int DIR> static PF_HD C subi(C a, C b) { return a - pf_mul_i<DIR>(b); }
  template <int DIR> static PF_HD C muli(C a) { return pf_mul_i<DIR>(a); }
  template <int DIR> static PF_HD C cmul(C a, TW w) { return pf_cmul(a, DIR > 0 ? w : pf_conj(w)); }
  template <int DIR> static PF_HD C cmul_s(C a, TW w) { return cmul<DIR>(a, w); }
  template <int DIR> static PF_HD C cmulc(C a, double c, double s) { return cmul<DIR>(a, mk(c, s)); }
  static PF_HD C scale(C a, double k) { return pf_scale(a, k); }
  static PF_HD TW twmul(TW a, TW b) { return pf_cmul(a, b); }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define PF_UNROLL _Pragma("unroll")
#else
#define PF_UNROLL
#endif

// X_k = sum_t u_t w^(k t), w = exp(DIR 2 pi i / R), natural order in and out
template <typename A, int DIR> PF_HD void pfx_bfly4(typename A::C &v0, typename A::C &v1, typename A::C &v2, typename A::C &v3) {
  const typename A::C a = A::add(v0, v2), b = A::sub(v0, v2), c = A::add(v1, v3), d = A::sub(v1, v3);
  v0 = A::add(a, c);
  v2 = A::sub(a, c);
  v1 = A::template addi<DIR>(b, d);
  v3 = A::template subi<DIR>(b, d);
}
template <typename A, int DIR> PF_HD void pfx_bfly8(typename A::C (&u)[8]) {
  typedef typename A::C C;
  pfx_bfly4<A, DIR>(u[0], u[2], u[4], u[6]);  // E_0..3 in u[0], u[2], u[4], u[6]
  pfx_bfly4<A, DIR>(u[1], u[3], u[5], u[7]);  // O_0..3 in u[1], u[3], u[5], u[7]
  const double h = 0.70710678118654752440;
  const C o1 = A::template cmulc<DIR>(u[3], h, h), o3 = A::template cmulc<DIR>(u[7], -h, h);  // W8^1, W8^3 (W8^2 = DIR i below)
  const C e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o2 = u[5];
  u[0] = A::add(e0, o0); u[4] = A::sub(e0, o0);
  u[1] = A::add(e1, o1); u[5] = A::sub(e1, o1);
  u[2] = A::template addi<DIR>(e2, o2); u[6] = A::template subi<DIR>(e2, o2);
  u[3] = A::add(e3, o3); u[7] = A::sub(e3, o3);
}
template <typename A, int DIR> PF_HD void pfx_bfly16(typename A::C (&x)[16]) {
  typedef typename A::C C;
  C e[8], o[8];
  PF_UNROLL
  for (int m = 0; m < 8; m++) { e[m] = x[2 * m]; o[m] = x[2 * m + 1]; }
  pfx_bfly8<A, DIR>(e);
  pfx_bfly8<A, DIR>(o);
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  o[1] = A::template cmulc<DIR>(o[1], c1, s1);   // W16^k, k = 1..7 but 4 (= DIR i, folded into the sums below)
  o[2] = A::template cmulc<DIR>(o[2], h, h);
  o[3] = A::template cmulc<DIR>(o[3], s1, c1);
  o[5] = A::template cmulc<DIR>(o[5], -s1, c1);
  o[6] = A::template cmulc<DIR>(o[6], -h, h);
  o[7] = A::template cmulc<DIR>(o[7], -c1, s1);
  PF_UNROLL
  for (int k = 0; k < 8; k++) {
    if (k == 4) { x[4] = A::template addi<DIR>(e[4], o[4]); x[12] = A::template subi<DIR>(e[4], o[4]); }
    else { x[k] = A::add(e[k], o[k]); x[k + 8] = A::sub(e[k], o[k]); }
  }
}

// ---- index algebra of the plan (threads tl = 0..127 of one column pair; w = tl >> 3 its wave, tl & 7 its place in the wave) ----
constexpr int PF16_N = 2048;
PF_HD int pf16_line_index(int tl, int m) { return tl + m * (PF16_N / 16); }
// exchange 1 (one column): slot written by thread tl for its output t; slot read by thread tl into register r
PF_HD int pf16_x1_write(int tl, int t) { return t * 128 + tl; }
PF_HD int pf16_x1_read(int tl, int r) { return (tl >> 3) * 128 + (tl & 7) + 8 * r; }
// exchange 2 (column col = 0, 1; the same slots for both halves b): slot written for output s = 8 b + s8 of the stage-1 butterfly,
// slot read into register 8 b + r.  Slot of (s8, u0) = s8 8 + (u0 ^ s8): sixteen neighbouring lanes (two threads of one column
// pair set) then touch ONE 128-byte block both when they write (u0 = 2 k, 2 k + 1) and when they read (s0 = 2 k, 2 k + 1)
PF_HD int pf16_x2_write(int tl, int s8, int col) { return (tl >> 3) * 128 + 64 * col + s8 * 8 + ((tl & 7) ^ s8); }
PF_HD int pf16_x2_read(int tl, int r, int col) { return (tl >> 3) * 128 + 64 * col + (tl & 7) * 8 + (r ^ (tl & 7)); }
// table indices (tw[j] = exp(+2 pi i j / 2048)) of the twiddles w^1 of the two later stages; w^r is tw[r * index] (never wraps)
PF_HD int pf16_tw1(int w) { return 8 * w; }
PF_HD int pf16_tw2(int tl, int b) { return (tl >> 3) + 16 * (tl & 7) + 128 * b; }
// the element of the line register m = 8 b + s1 of thread tl holds after the last stage: lane part + 128 b + 256 s1
PF_HD int pf16_out_lane(int tl) { return (tl >> 3) + 16 * (tl & 7); }
PF_HD int pf16_out_index(int tl, int m) { return pf16_out_lane(tl) + 128 * (m >> 3) + 256 * (m & 7); }

// w^2 .. w^7 from w^1 by products of depth <= 3 (as pf_stage_apply): wp[r - 1] = w^r
template <typename A> PF_HD void pfx_powers7(typename A::TW w1, typename A::TW (&wp)[7]) {
  const typename A::TW w2 = A::twmul(w1, w1), w3 = A::twmul(w2, w1), w4 = A::twmul(w2, w2);
  wp[0] = w1; wp[1] = w2; wp[2] = w3; wp[3] = w4; wp[4] = A::twmul(w4, w1); wp[5] = A::twmul(w3, w3); wp[6] = A::twmul(w4, w3);
}
