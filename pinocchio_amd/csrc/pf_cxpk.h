// pf_cxpk.h -- PfCxPk: the packed (re, im) fp32 complex algebra (round 5: the sixteen-point strided pass, pf_fft16.h; round 6: every
// fp32 transform of the library -- pf_fft_core.h routes pfc<float> through it on the device).
//
// One column's complex number is a (re, im) pair in a 64-bit register pair, exactly as it lies in memory, and every operation of the
// butterflies is ONE packed instruction: v_pk_add_f32 for sums and differences, the same with its second operand's halves swapped
// and one of them negated (op_sel / neg modifiers) for a +- i b and a +- conj(b), v_pk_mul_f32 + v_pk_fma_f32 for a product with a
// twiddle.  A packed fp32 instruction costs what one fp64 or one scalar fp32 fma costs (profiles/r05_notes.md: valu_probe).  The
// compiler packs the plain sums by itself, but builds every swapped-and-negated operand with v_xor + v_mov (157 + 42 of the 874
// vector instructions of the fp32 invariant z-pass of 2048 points, round 5): those forms are inline assembly here.
#pragma once

#ifndef PF_HD
#if defined(__HIPCC__)
#define PF_HD __host__ __device__ __forceinline__
#else
#define PF_HD inline
#endif
#endif

#if defined(__HIPCC__)
typedef float pf_f2 __attribute__((ext_vector_type(2)));
#else
struct pf_f2 { float x, y; };
inline pf_f2 operator+(pf_f2 a, pf_f2 b) { return pf_f2{a.x + b.x, a.y + b.y}; }
inline pf_f2 operator-(pf_f2 a, pf_f2 b) { return pf_f2{a.x - b.x, a.y - b.y}; }
#endif

struct PfCxPk {
  typedef pf_f2 C;   // one column: (re, im)
  typedef pf_f2 TW;  // a twiddle: (cos, sin)
  typedef float SC;
  static PF_HD C mk(float re, float im) { C r; r.x = re; r.y = im; return r; }
  static PF_HD C add(C a, C b) { return a + b; }
  static PF_HD C sub(C a, C b) { return a - b; }
  // a + DIR i b, a - DIR i b
  template <int DIR> static PF_HD C addi(C a, C b) {
#if defined(__HIP_DEVICE_COMPILE__)
    C d;
    if (DIR > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return DIR > 0 ? mk(a.x - b.y, a.y + b.x) : mk(a.x + b.y, a.y - b.x);
#endif
  }
  template <int DIR> static PF_HD C subi(C a, C b) { return addi<-DIR>(a, b); }
  // DIR i a
  template <int DIR> static PF_HD C muli(C a) {
#if defined(__HIP_DEVICE_COMPILE__)
    C d;
    if (DIR > 0) asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(d) : "v"(a));
    else asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
#else
    return DIR > 0 ? mk(-a.y, a.x) : mk(a.y, -a.x);
#endif
  }
  // a * w (DIR > 0) or a * conj(w) (DIR < 0); w in vector or scalar registers ("vs": the compiler's choice)
  template <int DIR> static PF_HD C cmul(C a, TW w) {
#if defined(__HIP_DEVICE_COMPILE__)
    C t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    if (DIR > 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
#else
    return DIR > 0 ? mk(a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y) : mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
#endif
  }
  // the same with the twiddle in SCALAR registers (the same for every lane of the wave)
  template <int DIR> static PF_HD C cmul_s(C a, TW w) {
#if defined(__HIP_DEVICE_COMPILE__)
    C t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    if (DIR > 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
#else
    return cmul<DIR>(a, w);
#endif
  }
  // a * (c + DIR i s), c and s constants
  template <int DIR> static PF_HD C cmulc(C a, double c, double s) { return cmul<DIR>(a, mk((float)c, (float)s)); }
  static PF_HD C scale(C a, float k) {
#if defined(__HIPCC__)
    return a * k;
#else
    return mk(a.x * k, a.y * k);
#endif
  }
  static PF_HD TW twmul(TW a, TW b) { return cmul<+1>(a, b); }
  // a + conj(b), a - conj(b) (the Hermitian fold of the z-passes)
  static PF_HD C addc(C a, C b) {
#if defined(__HIP_DEVICE_COMPILE__)
    C d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return mk(a.x + b.x, a.y - b.y);
#endif
  }
  static PF_HD C subc(C a, C b) {
#if defined(__HIP_DEVICE_COMPILE__)
    C d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return mk(a.x - b.x, a.y + b.y);
#endif
  }
  // a * k + c on both halves (normalisation + DC mode of a real pair)
  static PF_HD C fma1(C a, float k, float c) {
#if defined(__HIPCC__)
    return a * k + c;   // (one v_pk_fma_f32 with the scalars broadcast by op_sel: the compiler's own)
#else
    return mk(a.x * k + c, a.y * k + c);
#endif
  }
};

