// layout_probe.hip -- would tile-contiguous intermediates pay?  (round 4, for round 5)
// The strided passes write their 1024-row x 128-byte tiles as 1024 separate 128-byte pieces, rows apart (layout [outer][e][zp]).
// Variant B writes a tile as ONE contiguous 128 KB block (layout [outer][tile][e][8 columns]); whoever reads it next along another
// axis gathers 128-byte pieces either way.  Shapes (bare access patterns, no arithmetic, one LDS round trip per job):
//   x13 : 1 tile in (pieces),  3 tiles out  -- as now (pieces) / contiguous blocks
//   y36 : 3 tiles in (pieces), 6 tiles out  -- as now (pieces) / contiguous blocks
//   zrow: the z-pass side of it: six fields read row by row (8 KB contiguous rows) / as 65 pieces of 128 bytes 128 KB apart, three rows written
//   hipcc --offload-arch=gfx950 -O3 -o layout_probe profiles/tools/layout_probe.hip && ./layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr long long ZP = 520, N = 1024, NTILE = 65;
// pieces: element (outer, e, col) at outer*os + e*es + col;  blocks: at ((outer*NTILE + tile)*N + e)*8 + c
__device__ __forceinline__ long long addr(bool blocks, long long os, long long es, int outer, int tile, int e, int c) {
  return blocks ? (((long long)outer * NTILE + tile) * N + e) * 8 + c : (long long)outer * os + (long long)e * es + tile * 8 + c;
}
template <int NIN, int NOUT_PER_IN>
__global__ void __launch_bounds__(1024) k_tile(const f4 *const *in, f4 *const *out, long long in_os, long long in_es, bool in_blocks,
                                               long long out_os, long long out_es, bool out_blocks, long long nwork) {
  extern __shared__ char smem[];
  const long long per = (nwork + 7) >> 3;
  const long long w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x, c = tid & 7, tl = tid >> 3;
  const int tile = (int)(w % NTILE), outer = (int)(w / NTILE);
  f4 src[8];
  auto load = [&](int j) {
#pragma unroll
    for (int m = 0; m < 8; m++) src[m] = __builtin_nontemporal_load(in[j] + addr(in_blocks, in_os, in_es, outer, tile, tl + m * 128, c));
  };
  load(0);
  for (int j = 0; j < NIN; j++) {
    f4 v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = src[m];
    if (j + 1 < NIN) load(j + 1);
    for (int o = 0; o < NOUT_PER_IN; o++) {
      reinterpret_cast<f4 *>(smem)[tid] = v[0];
      __syncthreads();
      v[0] = reinterpret_cast<f4 *>(smem)[tid ^ 1];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; m++) __builtin_nontemporal_store(v[m], out[j * NOUT_PER_IN + o] + addr(out_blocks, out_os, out_es, outer, tile, tl + m * 128, c));
    }
  }
}
// z side: a workgroup of 384 threads takes row (x, y) of six fields (520 elements each) and writes three rows; six waves, one field each
__global__ void __launch_bounds__(384) k_zrow(const f4 *const *in, f4 *const *out, bool in_blocks, long long nrows) {
  const int l = threadIdx.x >> 6, tl = threadIdx.x & 63;
  for (long long r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int x = (int)(r / N), y = (int)(r % N);
    f4 v[9];
#pragma unroll
    for (int m = 0; m < 9; m++) {
      const int k = tl + 64 * m;
      // rows: [x][y][zp]; blocks: [x][tile][y][8]
      const long long a = in_blocks ? (((long long)x * NTILE + (k >> 3)) * N + y) * 8 + (k & 7) : ((long long)x * N + y) * ZP + k;
      v[m] = (k < 513) ? __builtin_nontemporal_load(in[l] + a) : f4{0, 0, 0, 0};
    }
    if (l < 3) {
#pragma unroll
      for (int m = 0; m < 8; m++) __builtin_nontemporal_store(v[m] + v[8], out[l] + ((long long)x * N + y) * ZP + tl + 64 * m);
    } else if (v[0].x == 12345.f) out[0][0] = v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + v[8];
  }
}
int main() {
  const long long field = N * N * ZP;  // >= N * NTILE * N * 8 = N*N*520
  const size_t fbytes = (size_t)field * 16;
  f4 *f[9];
  for (int i = 0; i < 9; i++) { CHECK(hipMalloc(&f[i], fbytes)); CHECK(hipMemset(f[i], 0, fbytes)); }
  f4 **d_in, **d_out;
  CHECK(hipMalloc(&d_in, 6 * sizeof(f4 *))); CHECK(hipMalloc(&d_out, 6 * sizeof(f4 *)));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const long long nwork = NTILE * N;
  const unsigned grid = (unsigned)(((nwork + 7) >> 3) << 3);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  auto time = [&](const char *name, double gb, auto launch) {
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (r && ms < best) best = ms;
    }
    printf("{\"shape\": \"%s\", \"ms\": %.3f, \"TBps\": %.3f}\n", name, best, gb / best); fflush(stdout);
  };
  // x-pass 1 -> 3: in KY layout (outer = y: os = zp, e = x: es = n * zp), out [y][x][zp] pieces (os = n*zp, es = zp) or blocks
  CHECK(hipMemcpy(d_in, f, 1 * sizeof(f4 *), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_out, f + 1, 3 * sizeof(f4 *), hipMemcpyHostToDevice));
  const double gb_tile = nwork * 131072.0 * 1e-9;
  for (int blocks = 0; blocks < 2; blocks++)
    time(blocks ? "x13 out blocks" : "x13 out pieces", 4 * gb_tile, [&]() {
      hipLaunchKernelGGL((k_tile<1, 3>), dim3(grid), dim3(1024), 128 * 1024, 0, (const f4 *const *)d_in, (f4 *const *)d_out, ZP, N * ZP, false, N * ZP, ZP, (bool)blocks, nwork); });
  // y-pass 3 -> 6: in [y][x][zp] (outer = x: os = zp, e = y: es = n*zp) pieces, or in blocks written by the x-pass ([y][tile][x][8]: outer = x, e = y
  //   -> a piece per y, 128 KB * 65 apart: emulated by swapping the roles in addr(): pieces with os = 8, es = NTILE*N*8 -- the tile term differs, a pattern equivalent)
  CHECK(hipMemcpy(d_in, f, 3 * sizeof(f4 *), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_out, f + 3, 6 * sizeof(f4 *), hipMemcpyHostToDevice));
  for (int blocks = 0; blocks < 2; blocks++)
    time(blocks ? "y36 out blocks" : "y36 out pieces", 9 * gb_tile, [&]() {
      hipLaunchKernelGGL((k_tile<3, 2>), dim3(grid), dim3(1024), 128 * 1024, 0, (const f4 *const *)d_in, (f4 *const *)d_out, ZP, N * ZP, false, N * ZP, ZP, (bool)blocks, nwork); });
  // z side: six fields in (rows / pieces), three rows out
  CHECK(hipMemcpy(d_in, f + 3, 6 * sizeof(f4 *), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_out, f, 3 * sizeof(f4 *), hipMemcpyHostToDevice));
  const double gb_z = (6.0 * 513 + 3.0 * 512) * 16.0 * N * N * 1e-9;
  for (int blocks = 0; blocks < 2; blocks++)
    for (int g = 32; g <= 128; g *= 2)
      time(blocks ? (g == 32 ? "zrow in pieces g32" : g == 64 ? "zrow in pieces g64" : "zrow in pieces g128") : (g == 32 ? "zrow in rows g32" : g == 64 ? "zrow in rows g64" : "zrow in rows g128"), gb_z, [&]() {
        hipLaunchKernelGGL(k_zrow, dim3(256 * g), dim3(384), 0, 0, (const f4 *const *)d_in, (f4 *const *)d_out, (bool)blocks, N * N); });
  return 0;
}
