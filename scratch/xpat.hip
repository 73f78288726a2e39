// scratch/xpat.hip -- the x-pass access pattern without its arithmetic: every workgroup reads one tile
// (1024 x-planes x 64 bytes) and writes three; does the plane pitch (a multiple of 2^17 bytes in the library)
// matter?  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NW>
__global__ void __launch_bounds__(512) k_xpat(const double2 *__restrict__ in, double2 *__restrict__ out, size_t plane, size_t field, int ntiles, int rowp, size_t iplane) {
  const int tid = threadIdx.x, c = tid & 3, tl = tid >> 2;
  const size_t b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous tiles
  const int tile = (int)(b % ntiles), ky = (int)(b / ntiles);
  const size_t base = (size_t)ky * rowp + tile * 4 + c;
  double2 v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = in[(size_t)(tl + m * 128) * iplane + base];
#pragma unroll
  for (int w = 0; w < NW; w++)
#pragma unroll
    for (int m = 0; m < 8; m++) out[w * field + (size_t)(tl + m * 128) * plane + base] = make_double2(v[m].x + w, v[m].y);
}

int main() {
  const int n = 1024, rowp = 520, ntiles = rowp / 4;
  const size_t maxplane = (size_t)(n + 8) * rowp, field = (size_t)n * maxplane;
  double2 *in, *out;
  CK(hipMalloc(&in, field * 16)); CK(hipMalloc(&out, 3 * field * 16));
  CK(hipMemset(in, 0, field * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t planes[] = {(size_t)n * rowp, (size_t)(n + 1) * rowp, (size_t)n * rowp + 8, (size_t)n * rowp + 16, (size_t)n * rowp + 64, (size_t)(n + 3) * rowp, (size_t)n * rowp};
  for (int ip = 0; ip < 2; ip++)
  for (size_t pi = 0; pi < sizeof(planes) / sizeof(planes[0]); pi++) {
    const size_t plane = planes[pi];
    const int grid = n * ntiles;
    k_xpat<3><<<grid, 512>>>(in, out, plane, field, ntiles, rowp, ip ? plane : (size_t)n * rowp); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; it++) k_xpat<3><<<grid, 512>>>(in, out, plane, field, ntiles, rowp, ip ? plane : (size_t)n * rowp);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 4.0 * n * (double)n * rowp * 16;
    printf("in %s  plane pitch %9zu complex (%10zu B = 2^%d x odd)  %6.2f ms  %5.2f TB/s\n", ip ? "padded too" : "2^17 pitch ", plane, plane * 16, __builtin_ctzll(plane * 16), ms / 5, 5 * bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
  }
  return 0;
}
