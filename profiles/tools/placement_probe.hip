// placement_probe.hip -- does WHERE the allocator puts the fields change the rate of the y-pass access shape?  (round 4)
// The bare 3 -> 6 tile kernel of hbm_probe.hip on freshly allocated fields, again and again in one process, with allocations of other
// sizes in between so that the nine fields land elsewhere; prints the device pointers and the time of every trial.
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe profiles/tools/placement_probe.hip && ./placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Shape { long long in_es, in_os, out_es, out_os; int ntiles, nouter; };
__global__ void __launch_bounds__(1024) k_tile(const f4 *const *in, f4 *const *out, Shape s, long long nwork) {
  extern __shared__ char smem[];
  const long long per = (nwork + 7) >> 3;
  const long long w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x, c = tid & 7, tl = tid >> 3;
  const int tile = (int)(w % s.ntiles), outer = (int)(w / s.ntiles);
  const long long col = tile * 8 + c;
  f4 src[8];
  auto load = [&](int j) {
    const f4 *p = in[j] + outer * s.in_os + col;
#pragma unroll
    for (int m = 0; m < 8; m++) src[m] = __builtin_nontemporal_load(p + (long long)(tl + m * 128) * s.in_es);
  };
  load(0);
  for (int j = 0; j < 3; j++) {
    f4 v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = src[m];
    if (j + 1 < 3) load(j + 1);
    reinterpret_cast<f4 *>(smem)[tid] = v[0];
    __syncthreads();
    v[0] = reinterpret_cast<f4 *>(smem)[tid ^ 1];
    __syncthreads();
    for (int o = 0; o < 2; o++) {
      f4 *q = out[j * 2 + o] + outer * s.out_os + col;
#pragma unroll
      for (int m = 0; m < 8; m++) __builtin_nontemporal_store(v[m], q + (long long)(tl + m * 128) * s.out_es);
    }
  }
}
int main() {
  const long long zp = 520, n = 1024, plane = n * zp, field = n * plane;
  const size_t fbytes = (size_t)field * 16;
  Shape y36 = {n * zp, zp, zp, n * zp, 65, (int)n};
  const long long nwork = 65LL * n;
  const unsigned grid = (unsigned)(((nwork + 7) >> 3) << 3);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f4 **d_in, **d_out;
  CHECK(hipMalloc(&d_in, 3 * sizeof(f4 *))); CHECK(hipMalloc(&d_out, 6 * sizeof(f4 *)));
  std::vector<void *> spacers;
  for (int trial = 0; trial < 10; trial++) {
    f4 *f[9];
    // mode: even trials nine separate allocations (as the library), odd trials one arena of nine fields
    f4 *arena = nullptr;
    if (trial & 1) { CHECK(hipMalloc(&arena, 9 * fbytes)); for (int i = 0; i < 9; i++) f[i] = arena + (size_t)i * field; }
    else for (int i = 0; i < 9; i++) CHECK(hipMalloc(&f[i], fbytes));
    for (int i = 0; i < 9; i++) CHECK(hipMemsetAsync(f[i], 0, fbytes, st));
    CHECK(hipMemcpyAsync(d_in, f, 3 * sizeof(f4 *), hipMemcpyHostToDevice, st));
    CHECK(hipMemcpyAsync(d_out, f + 3, 6 * sizeof(f4 *), hipMemcpyHostToDevice, st));
    CHECK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
      CHECK(hipEventRecord(a, st));
      hipLaunchKernelGGL(k_tile, dim3(grid), dim3(1024), 128 * 1024, st, (const f4 *const *)d_in, (f4 *const *)d_out, y36, nwork);
      CHECK(hipEventRecord(b, st)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (r && ms < best) best = ms;
    }
    printf("{\"trial\": %d, \"mode\": \"%s\", \"ms\": %.3f, \"TBps\": %.3f, \"ptrs\": [", trial, (trial & 1) ? "arena" : "separate", best, 9.0 * nwork * 131072.0 / best * 1e-9);
    for (int i = 0; i < 9; i++) printf("\"%p\"%s", (void *)f[i], i < 8 ? ", " : "");
    printf("]}\n"); fflush(stdout);
    if (arena) CHECK(hipFree(arena)); else for (int i = 0; i < 9; i++) CHECK(hipFree(f[i]));
    // perturb the allocator: keep an odd-sized block alive from now on
    void *sp; CHECK(hipMalloc(&sp, (size_t)(37 + 61 * trial) << 20)); spacers.push_back(sp);
  }
  return 0;
}
