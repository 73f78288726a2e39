// valu_probe.hip -- round 5: issue cost of the vector instructions an fp32 transform pass can be written in (gfx950), measured with
// the wave budget of the strided passes: 16 waves per CU (4 per SIMD), every wave a long run of independent instructions.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe profiles/tools/valu_probe.hip && ./valu_probe
//
// One JSON line per instruction: cycles per wave-instruction on one SIMD (s_memtime counts at a constant 100 MHz, the shader clock is
// taken from the wall time of a v_fma_f32 run at its nominal 2 cycles; what matters are the RATIOS between the lines).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// eight independent accumulators, 8 x UNROLL instructions per loop iteration
template <int KIND>
__global__ void __launch_bounds__(256) k_valu(float *out, int iters) {
  float a[8];
  f2 p[8];
  double d[8];
  const float s = (float)threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = s + i; p[i] = (f2){s + i, s - i}; d[i] = (double)s + i; }
  const f2 w = (f2){1.0000001f, 1e-9f};
  const double dw = 1.0000001, dz = 1e-9;
  const float fw = 1.0000001f, fz = 1e-9f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fw), "v"(fz));
        REP8(X)
#undef X
      } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w));
        REP8(X)
#undef X
      } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w));
        REP8(X)
#undef X
      } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w));
        REP8(X)
#undef X
      } else if (KIND == 4) {  // the second instruction of a complex multiply: halves swapped, one negated
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(p[i]) : "v"(w));
        REP8(X)
#undef X
      } else if (KIND == 5) {  // a + i b: one operand with its halves swapped and one of them negated
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(p[i]) : "v"(w));
        REP8(X)
#undef X
      } else if (KIND == 6) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dw), "v"(dz));
        REP8(X)
#undef X
      } else if (KIND == 7) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dz));
        REP8(X)
#undef X
      } else if (KIND == 8) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dw));
        REP8(X)
#undef X
      } else if (KIND == 9) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fz));
        REP8(X)
#undef X
      } else if (KIND == 10) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(fz));
        REP8(X)
#undef X
      } else if (KIND == 11) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(fz));
        REP8(X)
#undef X
      } else if (KIND == 12) {  // packed multiply with a scalar-register operand broadcast to both halves
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(w));
        REP8(X)
#undef X
      } else if (KIND == 13) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(fw));
        REP8(X)
#undef X
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y + (float)d[i];
  if (r == 12345.678f) out[0] = r;
}

template <int KIND>
static double run(const char *name, float *out, int ncu, double ref_ms) {
  const int iters = 4096;  // x 32 instructions
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(ncu * 4), dim3(256), 0, 0, out, 16);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(ncu * 4), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  // four waves per SIMD, iters * 32 instructions each
  const double per = best / (4.0 * iters * 32.0);  // ms per wave-instruction on a SIMD
  printf("{\"instruction\": \"%s\", \"ms\": %.4f, \"ns_per_wave_instruction\": %.4f, \"relative_to_v_fma_f32\": %.3f}\n", name, best, per * 1e6, ref_ms > 0 ? best / ref_ms : 1.0);
  fflush(stdout);
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc(&out, 4));
  const double ref = run<0>("v_fma_f32", out, ncu, 0);
  run<9>("v_add_f32", out, ncu, ref);
  run<13>("v_mul_f32 (sgpr operand)", out, ncu, ref);
  run<10>("v_mov_b32", out, ncu, ref);
  run<11>("v_mov_b32_dpp row_ror:8", out, ncu, ref);
  run<1>("v_pk_fma_f32", out, ncu, ref);
  run<2>("v_pk_add_f32", out, ncu, ref);
  run<3>("v_pk_mul_f32", out, ncu, ref);
  run<4>("v_pk_fma_f32 op_sel swap + neg_lo", out, ncu, ref);
  run<5>("v_pk_add_f32 op_sel swap + neg_lo", out, ncu, ref);
  run<12>("v_pk_mul_f32 (sgpr operand, broadcast)", out, ncu, ref);
  run<6>("v_fma_f64", out, ncu, ref);
  run<7>("v_add_f64", out, ncu, ref);
  run<8>("v_mul_f64", out, ncu, ref);
  return 0;
}
