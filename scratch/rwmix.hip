// scratch/rwmix.hip -- practical HBM ceilings for streams with a given share of writes (sizing reference for the
// strided passes: z-pass 50 % writes, y-pass 67 %, x-pass 75 %); not part of the library
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every workgroup streams contiguous 4 KiB pieces; NR input arrays are read, NW output arrays written
template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_mix(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n, size_t pitch) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 s = make_double2(0.0, (double)i);
#pragma unroll
    for (int r = 0; r < NR; r++) { double2 x = NT ? make_double2(__builtin_nontemporal_load(&in[r * pitch + i].x), __builtin_nontemporal_load(&in[r * pitch + i].y)) : in[r * pitch + i]; s.x += x.x; s.y += x.y; }
#pragma unroll
    for (int w = 0; w < NW; w++) {
      double2 v = make_double2(s.x + w, s.y);
      if (NT) { __builtin_nontemporal_store(v.x, &out[w * pitch + i].x); __builtin_nontemporal_store(v.y, &out[w * pitch + i].y); }
      else out[w * pitch + i] = v;
    }
  }
}

template <int NR, int NW, bool NT> static void run(const char *name, double2 *in, double2 *out, size_t n, size_t pitch, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_mix<NR, NW, NT><<<grid, 256>>>(in, out, n, pitch); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int it = 0; it < 5; it++) k_mix<NR, NW, NT><<<grid, 256>>>(in, out, n, pitch);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s grid %5d  %6.2f TB/s\n", name, grid, 5.0 * (NR + NW) * n * 16 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  const size_t pitch = ((size_t)2 << 30) / 16, n = pitch;  // 2 GiB per array
  double2 *in, *out;
  CK(hipMalloc(&in, 3 * pitch * 16)); CK(hipMalloc(&out, 6 * pitch * 16));
  CK(hipMemset(in, 0, 3 * pitch * 16));
  for (int grid : {2048, 8192}) {
    run<1, 0, false>("read only", in, out, n, pitch, grid);
    run<0, 1, false>("write only", in, out, n, pitch, grid);
    run<0, 1, true>("write only, nontemporal", in, out, n, pitch, grid);
    run<1, 1, false>("copy 1->1 (z-pass mix)", in, out, n, pitch, grid);
    run<1, 1, true>("copy 1->1, nontemporal", in, out, n, pitch, grid);
    run<3, 6, false>("3->6 (y-pass mix)", in, out, n, pitch, grid);
    run<1, 3, false>("1->3 (x-pass mix)", in, out, n, pitch, grid);
    run<1, 3, true>("1->3, nontemporal", in, out, n, pitch, grid);
  }
  return 0;
}
