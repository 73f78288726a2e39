// seg_probe.hip -- round 5: what the row-segment width is worth for 2048-point fp32 lines (BASELINE config 5), before any
// transform is written for it.  Bare access shapes of the strided passes (no arithmetic but an optional spin), one rank's slab of
// the 2048^3 box on eight ranks, fp32 fields (rows of 1040 complex = 8320 bytes):
//
//   hipcc --offload-arch=gfx950 -O3 -o seg_probe profiles/tools/seg_probe.hip && ./seg_probe > seg_probe.jsonl
//
//   TW   16-byte elements per row segment of a tile: 4 = 64 bytes (the library's k_strided<f32x2, 2048, 4>), 8 = 128 bytes
//   P    points of a line per thread: N / P * TW = 1024 threads per workgroup
//   MODE 0  the input tile is kept in registers over the jobs that use it (P = 8: the library; P = 16: does not fit a kernel
//           that also transforms -- an upper bound)
//        1  the input tile is read again for every job (plain loads, the last one non-temporal): what a 128-byte form must do
// One JSON object per line; TB/s of algorithmic bytes (every element once, however often it is really read).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Addr { long long os, hs, es; int sh; };  // outer stride, stride of e >> sh, stride of e & ((1 << sh) - 1); 16-byte units
struct Shape {
  int nin, nout;
  Addr in, out;
  int rowlen, nouter;   // 16-byte elements per row that are moved, lines of tiles
  long long field;      // 16-byte elements per field
};
__device__ __forceinline__ long long addr(const Addr &a, int outer, int e) {
  return (long long)outer * a.os + (long long)(e >> a.sh) * a.hs + (long long)(e & ((1 << a.sh) - 1)) * a.es;
}

// Addresses as in the library (pf_addr_uniform / pf_addr_lane): a 64-bit part that is the same for every lane (scalar registers)
// plus ONE 32-bit byte offset per lane -- as 64-bit addresses per element the sixteen-point forms spill.
__device__ __forceinline__ long long addr_uniform(const Addr &a, int outer, int e0) {  // e0 = m NT: a multiple of NT, NT | (1 << sh)
  return (long long)outer * a.os + (long long)(e0 >> a.sh) * a.hs + (long long)(e0 & ((1 << a.sh) - 1)) * a.es;
}
template <bool NT_> __device__ __forceinline__ f4 ld_ul(const f4 *base, long long u, unsigned lane_bytes) {
  const f4 *q = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(base + u) + (size_t)lane_bytes);
  return NT_ ? __builtin_nontemporal_load(q) : *q;
}
__device__ __forceinline__ void st_ul(f4 *base, long long u, unsigned lane_bytes, f4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(reinterpret_cast<char *>(base + u) + (size_t)lane_bytes));
}

template <int N, int TW, int P, int MODE>
__global__ void __launch_bounds__(1024) k_seg(const f4 *__restrict__ in, f4 *__restrict__ out, Shape s, long long nwork, int ntiles, int spin) {
  extern __shared__ char smem[];
  constexpr int NT = N / P;
  static_assert(NT * TW == 1024, "1024 threads");
  const long long per = (nwork + 7) >> 3;
  const long long w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x, c = tid % TW, tl = tid / TW;
  const int tile = (int)(w % ntiles), outer = (int)(w / ntiles);
  const bool valid = tile * TW + c < s.rowlen;
  const int opi = s.nout / s.nin;
  const unsigned lane_in = (unsigned)(tl * (unsigned)s.in.es + c) * 16u, lane_out = (unsigned)(tl * (unsigned)s.out.es + c) * 16u;
  f4 src[P];
  auto load = [&](int j, bool last) {
    const f4 *p = in + (long long)j * s.field + tile * TW;
    if (last) {
#pragma unroll
      for (int m = 0; m < P; m++) src[m] = valid ? ld_ul<true>(p, addr_uniform(s.in, outer, m * NT), lane_in) : (f4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int m = 0; m < P; m++) src[m] = valid ? ld_ul<false>(p, addr_uniform(s.in, outer, m * NT), lane_in) : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  if (MODE == 0) load(0, true);
#pragma unroll 1
  for (int j = 0; j < s.nin; j++) {
#pragma unroll 1
    for (int o = 0; o < opi; o++) {
      f4 v[P];
      if (MODE == 1) load(j, o == opi - 1);
#pragma unroll
      for (int m = 0; m < P; m++) v[m] = src[m];
      if (MODE == 0 && o == opi - 1 && j + 1 < s.nin) load(j + 1, true);
      reinterpret_cast<f4 *>(smem)[tid] = v[0];
      __syncthreads();
      v[0] = reinterpret_cast<f4 *>(smem)[tid ^ 1];
      __syncthreads();
      if (spin > 0) {
        float a = v[0].x, b = 1.0000001f;
        for (int piece = 0; piece < 4; piece++) {
          for (int i = 0; i < spin / 4; i++) a = __builtin_fmaf(a, b, 1e-9f);
          __syncthreads();
        }
        v[0].x = a;
      }
      if (valid) {
        f4 *q = out + (long long)(j * opi + o) * s.field + tile * TW;
#pragma unroll
        for (int m = 0; m < P; m++) st_ul(q, addr_uniform(s.out, outer, m * NT), lane_out, v[m]);
      }
    }
  }
}

static double time_best(int reps, hipStream_t st, const std::function<void()> &launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipStreamSynchronize(st));
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(a, st));
    launch();
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return best;
}

static const char *g_geom = "";
template <int N, int TW, int P, int MODE>
static void run(const char *name, const Shape &s, const f4 *in, f4 *out, hipStream_t st, int spin) {
  const int ntiles = (s.rowlen + TW - 1) / TW;
  const long long nwork = (long long)ntiles * s.nouter;
  const unsigned grid = (unsigned)(((nwork + 7) >> 3) << 3);
  const size_t shm = 128 * 1024;
  static bool raised = false;
  if (!raised) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seg<N, TW, P, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); raised = true; }
  const double ms = time_best(3, st, [&]() { hipLaunchKernelGGL((k_seg<N, TW, P, MODE>), dim3(grid), dim3(1024), shm, st, in, out, s, nwork, ntiles, spin); });
  const double bytes = (double)s.nouter * s.rowlen * 16.0 * N * (s.nin + s.nout);
  printf("{\"geometry\": \"%s\", \"shape\": \"%s\", \"rows_per_tile\": %d, \"segment_bytes\": %d, \"points_per_thread\": %d, \"mode\": \"%s\", \"nin\": %d, \"nout\": %d, \"spin\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n",
         g_geom, name, N, TW * 16, P, MODE == 0 ? "tile kept in registers" : "tile re-read per job", s.nin, s.nout, spin, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

template <int N, int TW, int P, int MODE>
static void run_shapes(const Shape &x13, const Shape &y36, const Shape &x11, const f4 *in, f4 *out, hipStream_t st) {
  for (int spin : {0, 400}) {
    run<N, TW, P, MODE>("xpass_1to3", x13, in, out, st, spin);
    run<N, TW, P, MODE>("ypass_3to6", y36, in, out, st, spin);
    run<N, TW, P, MODE>("xpass_1to1", x11, in, out, st, spin);
  }
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  fprintf(stderr, "%s, %d CUs\n", prop.name, prop.multiProcessorCount);
  {  // rank 0 of 8 of the 2048^3 box, fp32 fields: nl = 256 planes / ky rows per rank, rows of 520 16-byte elements
    const long long n = 2048, nl = 256, zp = 520, field = n * nl * zp;
    f4 *in, *out;
    CHECK(hipMalloc(&in, (size_t)3 * field * 16)); CHECK(hipMalloc(&out, (size_t)6 * field * 16));
    CHECK(hipMemset(in, 0, (size_t)3 * field * 16)); CHECK(hipMemset(out, 0, (size_t)6 * field * 16));
    // x-pass: reads KY [kx][ky_l][kz] along kx, writes YB [q][ky_l][x_l][kz] (x = q nl + x_l)
    Shape x13 = {1, 3, {zp, 0, nl * zp, 30}, {nl * zp, nl * nl * zp, zp, 8}, (int)zp, (int)nl, field};
    // y-pass: reads YB [q][ky_l][x_l][kz] along ky = q nl + ky_l, writes XS [x_l][ky][kz]
    Shape y36 = {3, 6, {zp, nl * nl * zp, nl * zp, 8}, {n * zp, 0, zp, 30}, (int)zp, (int)nl, field};
    Shape x11 = x13; x11.nout = 1;
    g_geom = "slab 1 of 8 of 2048^3, fp32 fields";
    run_shapes<2048, 4, 8, 0>(x13, y36, x11, in, out, st);
    run_shapes<2048, 8, 16, 1>(x13, y36, x11, in, out, st);
    run_shapes<2048, 4, 8, 1>(x13, y36, x11, in, out, st);
    // half lines on the same geometry (rows 0..1023 of every line only: no transform could do that, but it separates the number of
    // rows of a tile from the width of its segments): 1024 rows x 128 bytes, 1024 rows x 64 bytes (P = 4: two workgroups per CU would fit)
    g_geom = "slab 1 of 8 of 2048^3, fp32 fields, rows 0..1023 of every line only";
    run_shapes<1024, 8, 8, 0>(x13, y36, x11, in, out, st);
    run_shapes<1024, 4, 4, 0>(x13, y36, x11, in, out, st);
    CHECK(hipFree(in)); CHECK(hipFree(out));
  }
  {  // calibration: the 1024^3 fp64 box on one rank (the shapes of hbm_probe.hip)
    const long long n = 1024, zp = 520, field = n * n * zp;
    f4 *in, *out;
    CHECK(hipMalloc(&in, (size_t)3 * field * 16)); CHECK(hipMalloc(&out, (size_t)6 * field * 16));
    CHECK(hipMemset(in, 0, (size_t)3 * field * 16)); CHECK(hipMemset(out, 0, (size_t)6 * field * 16));
    Shape x13 = {1, 3, {zp, 0, n * zp, 30}, {n * zp, 0, zp, 30}, (int)zp, (int)n, field};
    Shape y36 = {3, 6, {zp, 0, n * zp, 30}, {n * zp, 0, zp, 30}, (int)zp, (int)n, field};
    Shape x11 = x13; x11.nout = 1;
    g_geom = "1024^3 fp64 on one rank";
    run_shapes<1024, 8, 8, 0>(x13, y36, x11, in, out, st);
    run_shapes<1024, 8, 8, 1>(x13, y36, x11, in, out, st);
    CHECK(hipFree(in)); CHECK(hipFree(out));
  }
  return 0;
}
