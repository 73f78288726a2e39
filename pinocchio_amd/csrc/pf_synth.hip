// pf_synth.hip -- synthetic delta(k) generated in HBM: the
// stand-in for the reference's GenIC feeder (src/GenIC.c:73-460) on benchmark
// and large property-test inputs.  Counter-based (Philox-4x32-10, counter =
// global pair-of-cells index), so every rank generates its own x-slab and the
// field does not depend on the decomposition.  numpy mirror: synth.philox_density.
#include "pf_internal.h"

#define PF_SYN_BLOCK 256

struct u4 { unsigned int a, b, c, d; };

__device__ __forceinline__ u4 pf_philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                               unsigned int k0, unsigned int k1) {
  const unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned int hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const unsigned int hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const unsigned int n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  u4 o; o.a = c0; o.b = c1; o.c = c2; o.d = c3;
  return o;
}

template <typename F>
__global__ void __launch_bounds__(PF_SYN_BLOCK)
    k_white(F *real, long long nrows, long long row0, int n, long long pitch, unsigned int k0, unsigned int k1) {
  const int half = n / 2;
  const long long npairs = nrows * half;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / half;
    const int j = (int)(i - row * half);
    const unsigned long long g = (unsigned long long)(row0 + row) * (unsigned long long)half + (unsigned long long)j;
    const u4 r = pf_philox4x32_10((unsigned int)(g & 0xFFFFFFFFull), (unsigned int)(g >> 32), 0u, 0u, k0, k1);
    const double u1 = ((double)(((unsigned long long)r.a << 21) ^ ((unsigned long long)r.b >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(((unsigned long long)r.c << 21) ^ ((unsigned long long)r.d >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 2.0 * 3.14159265358979323846 * u2;
    F *o = real + row * pitch + 2 * j;
    o[0] = (F)(rad * cos(ang));
    o[1] = (F)(rad * sin(ang));
  }
}

// spectrum rows are (x, y_local) with nzp complex each
template <typename F>
__global__ void __launch_bounds__(PF_SYN_BLOCK) k_shape(const PfShapeParams p) {
  __shared__ double red[PF_SYN_BLOCK / 64];
  F *spec = (F *)p.spec;
  const int n = p.n, nzh = n / 2 + 1, h = n / 2;
  const long long total = (long long)n * p.nyl * nzh;
  const double kf = 2.0 * 3.14159265358979323846 / (double)n;
  const double pi2 = 3.14159265358979323846 * 3.14159265358979323846;
  double acc = 0.0;
  const double ds = (p.mode == 1) ? *p.dscale : 1.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int kz = (int)(i % nzh);
    const long long r = i / nzh;
    const int yl = (int)(r % p.nyl);
    const int x = (int)(r / p.nyl);
    const int y = yl + p.y0;
    F *e = spec + 2 * (r * p.nzp + kz);
    if (p.mode == 1) {
      e[0] = (F)((double)e[0] * ds);
      e[1] = (F)((double)e[1] * ds);
      continue;
    }
    const int sx = x > h ? x - n : x, sy = y > h ? y - n : y;
    const double kx = kf * sx, ky = kf * sy, kzz = kf * kz;
    const double k2 = kx * kx + ky * ky + kzz * kzz;
    double amp = 0.0;
    if (k2 > 0.0 && k2 < pi2 && x != h && y != h && kz != h) amp = pow(k2, 0.25 * p.slope);
    const double re = (double)e[0] * amp, im = (double)e[1] * amp;
    e[0] = (F)re;
    e[1] = (F)im;
    const double w = (kz == 0 || kz == h) ? 1.0 : 2.0;
    acc += w * (re * re + im * im);
  }
  if (p.mode == 0) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0;
      for (int i = 0; i < PF_SYN_BLOCK / 64; i++) s += red[i];
      p.partials[blockIdx.x] = s;
    }
  }
}

__global__ void k_sigma_scale(const double *power_sum, double sigma0, double n3, double *dscale) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double var = power_sum[0] / (n3 * n3);
    dscale[0] = var > 0 ? sigma0 / sqrt(var) : 0.0;
  }
}

int pf_launch_white(int fb, void *real, long long nrows, long long row0, int n, long long pitch, uint64_t seed, hipStream_t st) {
  long long npairs = nrows * (n / 2);
  long long b = (npairs + PF_SYN_BLOCK - 1) / PF_SYN_BLOCK;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  const unsigned int k0 = (unsigned int)(seed & 0xFFFFFFFFull), k1 = (unsigned int)(seed >> 32);
  if (fb == 8) hipLaunchKernelGGL(k_white<double>, dim3((unsigned)b), dim3(PF_SYN_BLOCK), 0, st, (double *)real, nrows, row0, n, pitch, k0, k1);
  else hipLaunchKernelGGL(k_white<float>, dim3((unsigned)b), dim3(PF_SYN_BLOCK), 0, st, (float *)real, nrows, row0, n, pitch, k0, k1);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int pf_launch_shape(int fb, const PfShapeParams &p, hipStream_t st) {
  if (fb == 8) hipLaunchKernelGGL(k_shape<double>, dim3(p.nblocks), dim3(PF_SYN_BLOCK), 0, st, p);
  else hipLaunchKernelGGL(k_shape<float>, dim3(p.nblocks), dim3(PF_SYN_BLOCK), 0, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int pf_launch_sigma_scale(const double *power_sum, double sigma0, double n3, double *dscale, hipStream_t st) {
  hipLaunchKernelGGL(k_sigma_scale, dim3(1), dim3(64), 0, st, power_sum, sigma0, n3, dscale);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- streaming yardsticks (pf_debug_stream_rate): what this memory system gives a kernel that does nothing but read, write or
// copy -- the rates the transform passes are measured against in bench.py beside the 8 TB/s of the specification.  16 bytes per
// lane, grid-stride over the buffer, in the best form each kind showed in the sweep of profiles/tools/hbm_probe.hip
// (profiles/r04_hbm_ceilings.json: 144 / 144 / 216 variants of unroll, workgroup size, workgroups per CU, plain / non-temporal,
// grid-stride / one range per XCD): reads want many waves (non-temporal loads, 8 workgroups of 256 per CU: 7.2-7.3 TB/s against
// 6.6 for round 3's plain form), writes want FEW (one workgroup of 256 per CU: 6.4-6.5 TB/s; 8 per CU, round 3's yardstick: 5.0),
// a copy sits between (one workgroup per CU, two to four loads in flight per lane: 5.6-5.7 against 4.7).
typedef float pf_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream_read(const pf_f4 *p, size_t n, float *sink) {
  pf_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(p + i);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;  // never true for field data; keeps the loads
}
__global__ void __launch_bounds__(256) k_stream_write(pf_f4 *p, size_t n, float v) {
  const pf_f4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(x, p + i);
}
__global__ void __launch_bounds__(256) k_stream_copy(const pf_f4 *a, pf_f4 *b, size_t n) {
  const size_t nthr = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * nthr < n; i += 4 * nthr) {
    const pf_f4 v0 = a[i], v1 = a[i + nthr], v2 = a[i + 2 * nthr], v3 = a[i + 3 * nthr];
    b[i] = v0; b[i + nthr] = v1; b[i + 2 * nthr] = v2; b[i + 3 * nthr] = v3;
  }
  for (; i < n; i += nthr) b[i] = a[i];
}
int pf_launch_stream(int kind, const void *src, void *dst, size_t bytes, float *sink, hipStream_t st) {
  const size_t n = bytes / 16;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v; }
  const dim3 block(256);
  if (kind == 0) hipLaunchKernelGGL(k_stream_read, dim3(8 * ncu), block, 0, st, (const pf_f4 *)src, n, sink);
  else if (kind == 1) hipLaunchKernelGGL(k_stream_write, dim3(ncu), block, 0, st, (pf_f4 *)dst, n, 0.f);
  else hipLaunchKernelGGL(k_stream_copy, dim3(ncu), block, 0, st, (const pf_f4 *)src, (pf_f4 *)dst, n);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
