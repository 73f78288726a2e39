// mall_rewrite.hip -- does the 256 MB Infinity Cache absorb REPEATED writes of one buffer (write-back), or does every store reach HBM?
// One streaming store kernel over a buffer of S MB, launched R times back to back; GB/s against S.  (And the same for a copy
// inside one buffer pair, and for read-after-write of the same buffer.)
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/mall_rewrite.hip -o mall_rewrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ void __launch_bounds__(256) k_write(f4 *p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const f4 x = {v, v + 1, v + 2, v + 3};
    if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
  }
}
template <bool NT> __global__ void __launch_bounds__(256) k_read(const f4 *p, size_t n, float *out) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const f4 x = NT ? __builtin_nontemporal_load(p + i) : p[i];
    acc += x;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
int main() {
  const size_t maxb = (size_t)8 << 30;
  char *buf; float *out;
  hipMalloc(&buf, maxb); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# MB   rewrite GB/s  rewrite(nt)   write-then-read: write GB/s  read GB/s   (one buffer reused; 8 GB = never resident)\n");
  const size_t sizes[] = {16, 32, 64, 96, 128, 192, 256, 512, 2048, 8192};
  for (size_t s : sizes) {
    const size_t bytes = s << 20, n = bytes / 16;
    const int reps = (int)(((size_t)64 << 30) / bytes > 2000 ? 2000 : ((size_t)64 << 30) / bytes);
    float ms; double g[4];
    for (int var = 0; var < 2; var++) {
      hipLaunchKernelGGL(k_write<false>, dim3(2048), dim3(256), 0, 0, (f4 *)buf, n, 1.f); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; r++) {
        if (var) hipLaunchKernelGGL(k_write<true>, dim3(2048), dim3(256), 0, 0, (f4 *)buf, n, (float)r);
        else hipLaunchKernelGGL(k_write<false>, dim3(2048), dim3(256), 0, 0, (f4 *)buf, n, (float)r);
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      g[var] = (double)bytes * reps / (ms * 1e-3) / 1e9;
    }
    // write then read the same buffer, alternating: time each kind separately with events around every launch
    double tw = 0, tr = 0;
    hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
    const int reps2 = reps > 200 ? 200 : reps;
    for (int r = 0; r < reps2; r++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_write<false>, dim3(2048), dim3(256), 0, 0, (f4 *)buf, n, (float)r);
      hipEventRecord(b);
      hipLaunchKernelGGL(k_read<false>, dim3(2048), dim3(256), 0, 0, (const f4 *)buf, n, out);
      hipEventRecord(c); hipEventSynchronize(c);
      hipEventElapsedTime(&ms, a, b); tw += ms; hipEventElapsedTime(&ms, b, c); tr += ms;
    }
    g[2] = (double)bytes * reps2 / (tw * 1e-3) / 1e9; g[3] = (double)bytes * reps2 / (tr * 1e-3) / 1e9;
    printf("%5zu   %9.0f   %9.0f      %9.0f   %9.0f\n", s, g[0], g[1], g[2], g[3]);
  }
  return 0;
}
