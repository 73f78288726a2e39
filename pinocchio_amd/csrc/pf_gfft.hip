// pf_gfft.hip -- hand-written 3-D c2r / r2c transforms for ANY even grid size (the "general path" of pf_api.hip): the sizes the
// stage plans of pf_fft_kernels.hip (N = 2^k) and pf_mixed_kernels.hip (N = 8 m, m = 2^a 3^b 5^c) do not cover -- 20, 36, 100,
// anything with a prime factor 7, 11, 13 ... -- which the reference plans through FFTW / PFFT like any other GridSize
// (src/fmax-pfft.c:139-188).  Round 5: this replaces the hipFFT binding of rounds 1-4; the library no longer calls a transform
// it did not write.
//
// Every 1-D transform of length n is a chirp-z (Bluestein) transform on the power-of-two Stockham stages of pf_fft_core.h:
//   X_k = sum_j x_j e^(s 2 pi i j k / n),  j k = (j^2 + k^2 - (k - j)^2) / 2,  b_m = e^(s pi i m^2 / n):
//   X_k = b_k sum_j (x_j b_j) conj(b_(k-j))  --  a circular convolution of length M = 2^p >= 2 n - 1:
//   a = x b padded with zeros -> FFT_M -> times H = FFT_M(conj(b) wrapped) -> inverse FFT_M -> times b_k / M.
// One kernel, k_blue<M>: a workgroup owns T adjacent lines (T 16-byte columns of a tile in LDS, [M][T], 128 KB) and M / 8 threads
// per line, eight points each; the two M-point transforms run back to back in registers and LDS (PfStages), the chirp and the
// spectrum H come from tables made on the host in long double (m^2 reduced modulo 2 n before the angle is formed).  The 3-D
// transform is three such passes on the natural layouts (spectrum [n][n][n/2+1], real [n][n][n]): x and y in place on the
// spectrum, z between the spectrum rows and the real rows (Hermitian rows extended to n points; FFTW semantics: the imaginary
// parts of the DC and Nyquist modes are ignored).  ~6 M log M operations per line where a direct plan needs n log n: this is the
// path of completeness, at a fraction of the hand-planned sizes' speed (profiles/r05_notes.md).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pf_internal.h"
#include "pf_fft_core.h"
#include "pf_fft_stages.h"

struct PfBluePlan {
  int n, M;
  pfc<double> *b[2];   // chirp b_m = e^(s pi i m^2 / n), m < n: [0] s = +1 (inverse), [1] s = -1 (forward)
  pfc<double> *H[2];   // FFT_M of conj(b) wrapped, in the ORDER THE STAGES LEAVE IT IN REGISTERS' POSITIONS (natural order of k)
  pfc<double> *tw;     // exp(+2 pi i j / M), M entries
  hipStream_t st;
};

// mode 0: complex lines -> complex lines (in place allowed); 1: Hermitian rows (n/2+1) -> real rows (n); 2: real rows -> Hermitian rows
struct PfBlueParams {
  const void *in;
  void *out;
  long long os_in, ls_in, es_in, os_out, ls_out, es_out;  // element strides (complex elements; real elements for the real side of modes 1, 2): outer, line, element
  int ninner;          // lines per outer index (adjacent lines of a tile are adjacent inner lines)
  long long nlines;    // all lines
  int n, mode, s;      // s: 0 inverse (+), 1 forward (-)
  const pfc<double> *b, *H, *tw;
  double scale;        // 1 / M
};

// (MODE is a template parameter: with the three load / store forms behind run-time branches on p.mode, ROCm 7.2's compiler merged
//  their tails and lost the index of the chirp table on the real-input path -- a wild address; one instantiation per mode has no
//  such branches)
template <int M, int MODE>
__global__ void __launch_bounds__(1024) k_blue(const PfBlueParams p) {
  using C = pfc<double>;
  constexpr int NT = M / 8;              // threads per line
  constexpr int T = 1024 / NT < 1 ? 1 : 1024 / NT;  // lines per workgroup (M = 8192 would need 1024 threads for one line)
  constexpr int LAST = pf_nstages(M) - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C *lds = reinterpret_cast<C *>(smem);  // [M][T]
  const int tid = threadIdx.x, c = tid % T, tl = tid / T;
  const long long line = (long long)blockIdx.x * T + c;
  const bool valid = line < p.nlines;
  const long long lo = valid ? line / p.ninner : 0, li = valid ? line - lo * p.ninner : 0;
  const int n = p.n, h = n / 2;
  C v[8];
  // ---- a_j = x_j b_j, zeros beyond n
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int j = tl + m * NT;
    C x = pf_mk<double>(0.0, 0.0);
    if (valid && j < n) {
      if (MODE == 0) {
        x = reinterpret_cast<const C *>(p.in)[lo * p.os_in + li * p.ls_in + (long long)j * p.es_in];
      } else if (MODE == 1) {  // Hermitian extension of the row: X[n - j] = conj X[j]; Im X[0], Im X[n/2] ignored
        const C *row = reinterpret_cast<const C *>(p.in) + lo * p.os_in + li * p.ls_in;
        if (j <= h) { x = row[(long long)j * p.es_in]; if (j == 0 || j == h) x.y = 0.0; }
        else x = pf_conj(row[(long long)(n - j) * p.es_in]);
      } else {
        x = pf_mk<double>(reinterpret_cast<const double *>(p.in)[lo * p.os_in + li * p.ls_in + (long long)j * p.es_in], 0.0);
      }
      x = pf_cmul(x, p.b[j]);
    }
    v[m] = x;
  }
  auto wr = [&](int pos, C val) { lds[pos * T + c] = val; };
  auto rd = [&](int pos) { return lds[pos * T + c]; };
  // ---- A = FFT_M(a) (forward), times H, back (inverse): register m of thread tl holds position pf_stage_pos<M, LAST>(tl, m) after a transform
  PfStages<double, M, -1, 1>::run(v, tl, p.tw, wr, rd);
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int pos = pf_stage_pos<M, LAST>(tl, m);
    wr(pos, pf_cmul(v[m], p.H[pos]));
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = rd(tl + m * NT);
  __syncthreads();
  PfStages<double, M, +1, 1>::run(v, tl, p.tw, wr, rd);
  // ---- X_k = b_k y_k / M for k < n
  if (valid) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = pf_stage_pos<M, LAST>(tl, m);
      if (k >= n) continue;
      const C y = pf_scale(pf_cmul(v[m], p.b[k]), p.scale);
      if (MODE == 0) reinterpret_cast<C *>(p.out)[lo * p.os_out + li * p.ls_out + (long long)k * p.es_out] = y;
      else if (MODE == 1) reinterpret_cast<double *>(p.out)[lo * p.os_out + li * p.ls_out + (long long)k * p.es_out] = y.x;
      else if (k <= h) reinterpret_cast<C *>(p.out)[lo * p.os_out + li * p.ls_out + (long long)k * p.es_out] = (k == 0 || k == h) ? pf_mk<double>(y.x, 0.0) : y;
    }
  }
}

template <int M, int MODE> static int blue_launch_m(const PfBlueParams &p, hipStream_t st) {
  constexpr int NT = M / 8, T = 1024 / NT < 1 ? 1 : 1024 / NT;
  const size_t shm = (size_t)M * T * sizeof(pfc<double>);
  // (the attribute belongs to the kernel ON ONE DEVICE: a flag per device, as pf_launch_strided16 keeps them)
  static std::atomic<bool> raised[PF_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PF_MAX_DEVICES) return 3;
  if (shm > 64 * 1024 && !raised[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_blue<M, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 3;
    raised[dev].store(true, std::memory_order_release);
  }
  const long long nblk = (p.nlines + T - 1) / T;
  hipLaunchKernelGGL((k_blue<M, MODE>), dim3((unsigned)nblk), dim3(NT * T), shm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int M> static int blue_launch(const PfBlueParams &p, hipStream_t st) {
  return p.mode == 0 ? blue_launch_m<M, 0>(p, st) : p.mode == 1 ? blue_launch_m<M, 1>(p, st) : blue_launch_m<M, 2>(p, st);
}
static int blue_dispatch(int M, const PfBlueParams &p, hipStream_t st) {
  switch (M) {
    case 16: return blue_launch<16>(p, st);
    case 32: return blue_launch<32>(p, st);
    case 64: return blue_launch<64>(p, st);
    case 128: return blue_launch<128>(p, st);
    case 256: return blue_launch<256>(p, st);
    case 512: return blue_launch<512>(p, st);
    case 1024: return blue_launch<1024>(p, st);
    case 2048: return blue_launch<2048>(p, st);
    case 4096: return blue_launch<4096>(p, st);
    default: return 2;
  }
}

// host: in-place iterative radix-2 FFT (forward sign) of M = 2^p long double complex numbers, for the table H
static void host_fft(std::vector<long double> &re, std::vector<long double> &im) {
  const size_t M = re.size();
  for (size_t i = 1, j = 0; i < M; i++) {
    size_t bit = M >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  const long double PI = 3.14159265358979323846264338327950288L;
  for (size_t len = 2; len <= M; len <<= 1) {
    for (size_t k = 0; k < len / 2; k++) {
      const long double a = -2.0L * PI * (long double)k / (long double)len, wr = cosl(a), wi = sinl(a);
      for (size_t i = k; i < M; i += len) {
        const size_t j = i + len / 2;
        const long double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
        re[j] = re[i] - xr; im[j] = im[i] - xi;
        re[i] += xr; im[i] += xi;
      }
    }
  }
}

int pf_gfft_create(int n, hipStream_t st, void **c2r, void **r2c) {
  if (n < 4 || n > 2048 || (n & 1)) { printf("ERROR on task 0: the general transform path takes even grid sizes in [4, 2048], not %d\n", n); return 1; }
  int M = 16;
  while (M < 2 * n - 1) M <<= 1;
  PfBluePlan *pl = new PfBluePlan();
  pl->n = n; pl->M = M; pl->st = st;
  pl->b[0] = pl->b[1] = pl->H[0] = pl->H[1] = pl->tw = nullptr;
  const long double PI = 3.14159265358979323846264338327950288L;
  std::vector<double> tmp;
  auto upload = [&](pfc<double> **dst, size_t count) -> int {
    if (hipMalloc((void **)dst, count * sizeof(pfc<double>)) != hipSuccess) return 1;
    return hipMemcpy(*dst, tmp.data(), count * sizeof(pfc<double>), hipMemcpyHostToDevice) != hipSuccess;
  };
  int rc = 0;
  for (int s = 0; s < 2 && !rc; s++) {
    const long double sg = s == 0 ? 1.0L : -1.0L;
    std::vector<long double> br(n), bi(n), hr(M, 0.0L), hi(M, 0.0L);
    for (int m = 0; m < n; m++) {
      const long long m2 = ((long long)m * m) % (2LL * n);  // the angle pi m^2 / n only matters modulo 2 pi
      const long double a = sg * PI * (long double)m2 / (long double)n;
      br[m] = cosl(a); bi[m] = sinl(a);
    }
    tmp.assign((size_t)2 * n, 0.0);
    for (int m = 0; m < n; m++) { tmp[2 * m] = (double)br[m]; tmp[2 * m + 1] = (double)bi[m]; }
    rc = upload(&pl->b[s], n);
    // h_m = conj(b_m) at indices m and M - m (m = 1 .. n-1), h_0 at 0
    for (int m = 0; m < n; m++) {
      hr[m] = br[m]; hi[m] = -bi[m];
      if (m) { hr[M - m] = br[m]; hi[M - m] = -bi[m]; }
    }
    host_fft(hr, hi);
    tmp.assign((size_t)2 * M, 0.0);
    for (int k = 0; k < M; k++) { tmp[2 * k] = (double)hr[k]; tmp[2 * k + 1] = (double)hi[k]; }
    if (!rc) rc = upload(&pl->H[s], M);
  }
  tmp.assign((size_t)2 * M, 0.0);
  for (int j = 0; j < M; j++) {
    const long double a = 2.0L * PI * (long double)j / (long double)M;
    double re = (double)cosl(a), im = (double)sinl(a);
    if (j == 0) { re = 1; im = 0; } else if (4 * j == M) { re = 0; im = 1; } else if (2 * j == M) { re = -1; im = 0; } else if (4 * j == 3 * M) { re = 0; im = -1; }
    tmp[2 * j] = re; tmp[2 * j + 1] = im;
  }
  if (!rc) rc = upload(&pl->tw, M);
  if (rc) { printf("ERROR on task 0: the tables of the general transform path (n = %d, M = %d) could not be placed on the device\n", n, M); pf_gfft_destroy(pl); return 2; }
  *c2r = pl;
  *r2c = pl;  // one plan serves both directions: the caller releases it once
  return 0;
}

static void blue_common(const PfBluePlan *pl, PfBlueParams &p, int s) {
  p.n = pl->n; p.s = s; p.b = pl->b[s]; p.H = pl->H[s]; p.tw = pl->tw; p.scale = 1.0 / (double)pl->M;
}

// spectrum [n][n][n/2+1] -> real [n][n][n], unnormalised; the spectrum is overwritten by its x- and y-transforms
int pf_gfft_c2r(void *plan, void *spec, void *real) {
  const PfBluePlan *pl = (const PfBluePlan *)plan;
  const long long n = pl->n, nzh = n / 2 + 1;
  PfBlueParams p;
  blue_common(pl, p, 0);
  // x: lines (y, kz) flattened (adjacent in memory), elements n nzh apart
  p.in = spec; p.out = spec; p.mode = 0;
  p.os_in = p.os_out = 0; p.ls_in = p.ls_out = 1; p.es_in = p.es_out = n * nzh; p.ninner = (int)(n * nzh); p.nlines = n * nzh;
  if (int rc = blue_dispatch(pl->M, p, pl->st)) return rc;
  // y: lines (x; kz), elements nzh apart
  p.os_in = p.os_out = n * nzh; p.ls_in = p.ls_out = 1; p.es_in = p.es_out = nzh; p.ninner = (int)nzh; p.nlines = n * nzh;
  if (int rc = blue_dispatch(pl->M, p, pl->st)) return rc;
  // z: Hermitian rows -> real rows
  p.out = real; p.mode = 1;
  p.os_in = 0; p.ls_in = nzh; p.es_in = 1; p.os_out = 0; p.ls_out = n; p.es_out = 1; p.ninner = (int)(n * n); p.nlines = n * n;
  return blue_dispatch(pl->M, p, pl->st);
}
// real [n][n][n] -> spectrum [n][n][n/2+1], unnormalised
int pf_gfft_r2c(void *plan, void *real, void *spec) {
  const PfBluePlan *pl = (const PfBluePlan *)plan;
  const long long n = pl->n, nzh = n / 2 + 1;
  PfBlueParams p;
  blue_common(pl, p, 1);
  p.in = real; p.out = spec; p.mode = 2;
  p.os_in = 0; p.ls_in = n; p.es_in = 1; p.os_out = 0; p.ls_out = nzh; p.es_out = 1; p.ninner = (int)(n * n); p.nlines = n * n;
  if (int rc = blue_dispatch(pl->M, p, pl->st)) return rc;
  p.in = spec; p.mode = 0;
  p.os_in = p.os_out = n * nzh; p.ls_in = p.ls_out = 1; p.es_in = p.es_out = nzh; p.ninner = (int)nzh; p.nlines = n * nzh;
  if (int rc = blue_dispatch(pl->M, p, pl->st)) return rc;
  p.os_in = p.os_out = 0; p.ls_in = p.ls_out = 1; p.es_in = p.es_out = n * nzh; p.ninner = (int)(n * nzh); p.nlines = n * nzh;
  return blue_dispatch(pl->M, p, pl->st);
}
void pf_gfft_destroy(void *plan) {
  if (!plan) return;
  PfBluePlan *pl = (PfBluePlan *)plan;
  for (int s = 0; s < 2; s++) { (void)hipFree(pl->b[s]); (void)hipFree(pl->H[s]); }
  (void)hipFree(pl->tw);
  delete pl;
}

// test tap without a context: the 3-D transforms above on host arrays (natural layouts), n even: dir > 0: spectrum [n][n][n/2+1]
// complex -> real [n][n][n]; dir < 0: real -> spectrum.  tests/test_gpu_lines.py holds them against numpy's pocketfft.
extern "C" int pf_debug_gfft(int n, int dir, const double *in, double *out) {
  void *a = nullptr, *b = nullptr;
  if (pf_gfft_create(n, nullptr, &a, &b)) return 1;
  const size_t nspec = (size_t)n * n * (n / 2 + 1) * 2, nreal = (size_t)n * n * n;
  double *dspec = nullptr, *dreal = nullptr;
  int rc = 1;
  if (hipMalloc((void **)&dspec, nspec * sizeof(double)) == hipSuccess && hipMalloc((void **)&dreal, nreal * sizeof(double)) == hipSuccess) {
    if (dir > 0) {
      if (hipMemcpy(dspec, in, nspec * sizeof(double), hipMemcpyHostToDevice) == hipSuccess && pf_gfft_c2r(a, dspec, dreal) == 0 &&
          hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dreal, nreal * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    } else {
      if (hipMemcpy(dreal, in, nreal * sizeof(double), hipMemcpyHostToDevice) == hipSuccess && pf_gfft_r2c(a, dreal, dspec) == 0 &&
          hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dspec, nspec * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
  }
  (void)hipFree(dspec); (void)hipFree(dreal);
  pf_gfft_destroy(a);
  return rc;
}

// test tap without a context: ONE pass of k_blue on a few lines of n points (any even n in 4..2048: M = 16 .. 4096 -- the sizes a
// whole n^3 box of which no test can afford), host in / host out.  mode 0: complex lines stored as the x- and y-passes of the 3-D
// transforms meet them, [n][nlines] (adjacent lines adjacent in memory, elements nlines apart), dir > 0 inverse, < 0 forward;
// mode 1: Hermitian rows [nlines][n/2+1] -> real rows [nlines][n] (inverse); mode 2: real rows -> Hermitian rows (forward).  Unnormalised.
extern "C" int pf_debug_gfft_lines(int n, int mode, int dir, int nlines, const double *in, double *out) {
  if (!in || !out || nlines < 1 || mode < 0 || mode > 2) return 1;
  void *a = nullptr, *b = nullptr;
  if (pf_gfft_create(n, nullptr, &a, &b)) return 1;
  const PfBluePlan *pl = (const PfBluePlan *)a;
  const long long nzh = n / 2 + 1;
  const size_t n_in = mode == 0 ? (size_t)2 * n * nlines : mode == 1 ? (size_t)2 * nzh * nlines : (size_t)n * nlines;
  const size_t n_out = mode == 0 ? (size_t)2 * n * nlines : mode == 1 ? (size_t)n * nlines : (size_t)2 * nzh * nlines;
  double *din = nullptr, *dout = nullptr;
  int rc = 1;
  if (hipMalloc((void **)&din, n_in * sizeof(double)) == hipSuccess && hipMalloc((void **)&dout, n_out * sizeof(double)) == hipSuccess &&
      hipMemcpy(din, in, n_in * sizeof(double), hipMemcpyHostToDevice) == hipSuccess && hipMemset(dout, 0, n_out * sizeof(double)) == hipSuccess) {
    PfBlueParams p;
    blue_common(pl, p, mode == 0 ? (dir > 0 ? 0 : 1) : mode == 1 ? 0 : 1);
    p.in = din; p.out = dout; p.mode = mode; p.nlines = nlines; p.ninner = nlines;
    if (mode == 0) { p.os_in = p.os_out = 0; p.ls_in = p.ls_out = 1; p.es_in = p.es_out = nlines; }
    else if (mode == 1) { p.os_in = 0; p.ls_in = nzh; p.es_in = 1; p.os_out = 0; p.ls_out = n; p.es_out = 1; }
    else { p.os_in = 0; p.ls_in = n; p.es_in = 1; p.os_out = 0; p.ls_out = nzh; p.es_out = 1; }
    if (blue_dispatch(pl->M, p, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(out, dout, n_out * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  (void)hipFree(din); (void)hipFree(dout);
  pf_gfft_destroy(a);
  return rc;
}
