// scratch/mall.hip -- does a buffer written by one kernel get served from the Infinity Cache to the next kernel?
// (sizing experiment for a y-pass -> z-pass handoff through the 256 MiB L3; not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_write(double2 *p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_double2(v, (double)i);
}
__global__ void __launch_bounds__(256) k_read(const double2 *p, size_t n, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 x = p[i]; s += x.x + x.y; }
  if (s == 1.2345e-300) *sink = s;
}

int main() {
  const size_t total = (size_t)8 << 30;  // 8 GiB arena
  double2 *buf; double *sink;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  printf("# chunk_MB  write_TBps  read_after_write_TBps  pair_TBps(chunked W,R over the arena)  read_only_TBps(arena sweep)\n");
  const size_t sizes[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
  for (size_t si = 0; si < sizeof(sizes) / sizeof(sizes[0]); si++) {
    const size_t cb = sizes[si] << 20, cn = cb / 16, nchunk = total / cb;
    float ms;
    // (1) W all chunks back to back
    k_write<<<grid, 256>>>(buf, total / 16, 1.0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (size_t c = 0; c < nchunk; c++) k_write<<<grid, 256>>>(buf + c * cn, cn, 2.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double w = total / (ms * 1e-3) / 1e12;
    // (2) R all chunks back to back (arena sweep: from HBM)
    CK(hipEventRecord(e0));
    for (size_t c = 0; c < nchunk; c++) k_read<<<grid, 256>>>(buf + c * cn, cn, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double r = total / (ms * 1e-3) / 1e12;
    // (3) W(c), R(c) alternating: the pair moves 2x the arena
    CK(hipEventRecord(e0));
    for (size_t c = 0; c < nchunk; c++) { k_write<<<grid, 256>>>(buf + c * cn, cn, 3.0); k_read<<<grid, 256>>>(buf + c * cn, cn, sink); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double pair = 2.0 * total / (ms * 1e-3) / 1e12;
    // read rate inside the pair, if the writes cost what (1) says
    const double t_r = ms * 1e-3 - total / (w * 1e12);
    printf("%8zu  %6.2f  %6.2f  %6.2f  %6.2f\n", sizes[si], w, total / t_r / 1e12, pair, r);
    fflush(stdout);
  }
  // (4) same with an unrelated stream of X MB between the write and the read of a 64 MB chunk
  printf("# 64 MB chunk, X MB of other traffic (half read, half written) between its write and its read: read_TBps\n");
  const size_t cb = (size_t)64 << 20, cn = cb / 16;
  double2 *other = buf + ((size_t)4 << 30) / 16;
  for (size_t x = 0; x <= 512; x = x ? x * 2 : 32) {
    const size_t on = (x << 20) / 16 / 2;
    float acc = 0;
    for (int it = 0; it < 20; it++) {
      k_write<<<grid, 256>>>(buf, cn, 4.0);
      if (on) { k_read<<<grid, 256>>>(other, on, sink); k_write<<<grid, 256>>>(other + on, on, 5.0); }
      CK(hipEventRecord(e0));
      k_read<<<grid, 256>>>(buf, cn, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); acc += ms;
    }
    printf("%6zu  %6.2f\n", x, 20.0 * cb / (acc * 1e-3) / 1e12);
  }
  return 0;
}
