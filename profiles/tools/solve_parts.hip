// solve_parts.hip -- where the per-cell collapse solve spends its time: the pipeline of pf_collapse_core.h cut off after each
// part, timed on Gaussian random tensors (unit variance per component scale, like a field smoothed at ~1 cell).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pinocchio_amd/csrc profiles/tools/solve_parts.hip -o gpurun_out/solve_parts
//   gpurun_out/solve_parts [cells] [sigma]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "pf_collapse_core.h"

// timing probes of the inverse-growth evaluation (not the library's code): how the spline is looked up
template <int VAR> __device__ __forceinline__ double probe_invgrow(const pf_spline_view &s, const unsigned short *lut, double inv_w, const double *rec, double D) {
  const double v = pf_log10_pos(D);
  double sv;
  if (VAR == 0) sv = v * 1.01;                       // no spline at all
  else {
    const double *xa = s.x;
    const int last = s.n - 1;
    int bin = (int)((v - xa[0]) * inv_w) - 1;
    bin = bin < 0 ? 0 : (bin > PF_SPLINE_LUT_BINS - 1 ? PF_SPLINE_LUT_BINS - 1 : bin);
    int ilo = lut[bin];
    if (VAR == 2 || VAR == 4) { ilo += (ilo + 1 < last && xa[ilo + 1] <= v) ? 1 : 0; ilo += (ilo + 1 < last && xa[ilo + 1] <= v) ? 1 : 0; }   // two branch-free steps
    if (VAR == 3) { while (ilo + 1 < last && xa[ilo + 1] <= v) ilo++; }
    if (VAR == 4) {                                  // packed records [x, y, b, c, d]
      const double *r = rec + 5 * ilo;
      const double delx = v - r[0];
      sv = r[1] + delx * (r[2] + delx * (r[3] + delx * r[4]));
    } else {
      const double delx = v - xa[ilo];
      sv = s.y[ilo] + delx * (s.b[ilo] + delx * (s.c[ilo] + delx * s.d[ilo]));
    }
  }
  return pf_exp10_series(-sv) - 1.;
}

template <int STOP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_part(const double *mu, size_t n, pf_spline_view s, double *out) {
  __shared__ double sk[5 * 512];
  __shared__ double rec[5 * 512];
  __shared__ unsigned short slut[PF_SPLINE_LUT_BINS];
  for (int i = threadIdx.x; i < s.n; i += blockDim.x) { rec[5 * i] = s.x[i]; rec[5 * i + 1] = s.y[i]; rec[5 * i + 2] = s.b[i]; rec[5 * i + 3] = s.c[i]; rec[5 * i + 4] = s.d[i]; }
  for (int i = threadIdx.x; i < s.n; i += blockDim.x) { sk[i] = s.x[i]; sk[512 + i] = s.y[i]; sk[1024 + i] = s.c[i]; sk[1536 + i] = s.b[i]; sk[2048 + i] = s.d[i]; }
  __syncthreads();
  pf_spline_view sv; sv.x = sk; sv.y = sk + 512; sv.c = sk + 1024; sv.b = sk + 1536; sv.d = sk + 2048; sv.n = s.n;
  // STOP 7: the library's table (direct form for this knot set); the probes use the walk geometry
  double lx0, inv_w;
  pf_spline_lut_geometry(sk, s.n, STOP == 7, lx0, inv_w);
  for (int b = threadIdx.x; b < PF_SPLINE_LUT_BINS; b += blockDim.x) slut[b] = pf_spline_lut_entry(sk, s.n, b, lx0, inv_w, STOP == 7);
  __syncthreads();
  if (STOP == 7) { sv.lut = slut; sv.lut_inv_w = inv_w; sv.lut_x0 = lx0; sv.lut_direct = 1; sv.x_first = sk[0]; sv.x_last = sk[s.n - 1]; }
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double m1 = mu[i], m2 = mu[n + i], m3 = mu[2 * n + i];
    if (STOP == 0) { acc += m1 + m2 + m3; continue; }
    const double third = m1 * (1.0 / 3.0), diag[3] = {third, third, third};
    double lam[3];
    const bool ok = pf_eigen_from_invariants<true>(m1, m2, m3, diag, lam);
    if (STOP == 1) { acc += lam[0] + lam[1] * 0.5 + lam[2] * 0.25 + (ok ? 1 : 0); continue; }
    double F = -10.0;
    if (ok) {
      double ell = 0.0;
      pf_cubic c;
      const int kind = pf_ell_setup<true>(lam[0], lam[1], lam[2], ell, c);
      if (STOP == 2) { acc += ell + c.q + c.r + c.disc + c.a1 + kind; continue; }
      if (STOP == 3) { if (kind == 1) ell = pf_ell_one_root<true>(c); acc += ell; continue; }      // one-root branch only
      if (STOP == 4) { if (kind == 2) ell = pf_ell_three_roots<true>(c); acc += ell; continue; }   // three-root branch only
      if (kind == 1) ell = pf_ell_one_root<true>(c);
      else if (kind == 2) ell = pf_ell_three_roots<true>(c);
      if (STOP == 5) { acc += ell; continue; }
      ell = pf_ell_finish<true>(ell, lam[0], lam[1], lam[2]);
      if (STOP == 6) { acc += ell; continue; }
      if (STOP <= 8) F = ell > 0.0 ? 1. + pf_inverse_growing_mode<true>(sv, ell) : 0.0;       // 7: with the start table; 8: bisection
      else F = ell > 0.0 ? 1. + probe_invgrow<STOP - 9>(sv, slut, inv_w, rec, ell) : 0.0;
    }
    acc += F;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// probe: the two branches of the cubic executed on cells regrouped inside the workgroup (one-root cells to the low thread indices,
// three-root cells to the high ones, through LDS), so that most waves run ONE branch; three barriers per iteration
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_regroup(const double *mu, size_t n, pf_spline_view s, double *out) {
  __shared__ double sk[5 * 512];
  __shared__ unsigned short slut[PF_SPLINE_LUT_BINS];
  __shared__ double P[256 * 4];
  __shared__ double E[256];
  __shared__ unsigned short org[256];
  __shared__ int cnt[8];
  for (int i = threadIdx.x; i < s.n; i += blockDim.x) { sk[i] = s.x[i]; sk[512 + i] = s.y[i]; sk[1024 + i] = s.c[i]; sk[1536 + i] = s.b[i]; sk[2048 + i] = s.d[i]; }
  __syncthreads();
  pf_spline_view sv; sv.x = sk; sv.y = sk + 512; sv.c = sk + 1024; sv.b = sk + 1536; sv.d = sk + 2048; sv.n = s.n;
  double lx0, inv_w;
  pf_spline_lut_geometry(sk, s.n, true, lx0, inv_w);
  for (int b = threadIdx.x; b < PF_SPLINE_LUT_BINS; b += blockDim.x) slut[b] = pf_spline_lut_entry(sk, s.n, b, lx0, inv_w, true);
  __syncthreads();
  sv.lut = slut; sv.lut_inv_w = inv_w; sv.lut_x0 = lx0; sv.lut_direct = 1; sv.x_first = sk[0]; sv.x_last = sk[s.n - 1];
  const int tid = threadIdx.x, w = tid >> 6;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < n; i += (size_t)gridDim.x * blockDim.x) {  // n is a multiple of the grid: uniform trip count
    const double m1 = mu[i], m2 = mu[n + i], m3 = mu[2 * n + i];
    const double third = m1 * (1.0 / 3.0), diag[3] = {third, third, third};
    double lam[3];
    const bool ok = pf_eigen_from_invariants<true>(m1, m2, m3, diag, lam);
    double ell = 0.0;
    pf_cubic c;
    c.a1 = c.q = c.r = c.disc = 0.0;
    const int kind = ok ? pf_ell_setup<true>(lam[0], lam[1], lam[2], ell, c) : 0;
    const unsigned long long b1 = __ballot(kind == 1), b2 = __ballot(kind == 2);
    const int p1 = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0));
    const int p2 = __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
    if ((tid & 63) == 0) { cnt[w] = __popcll(b1); cnt[4 + w] = __popcll(b2); }
    __syncthreads();
    int base1 = 0, base2 = 0, tot1 = 0, tot2 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int a = cnt[k], b = cnt[4 + k]; if (k < w) { base1 += a; base2 += b; } tot1 += a; tot2 += b; }
    const int slot = kind == 1 ? base1 + p1 : 255 - (base2 + p2);
    if (kind > 0) { P[4 * slot] = c.a1; P[4 * slot + 1] = c.q; P[4 * slot + 2] = c.r; P[4 * slot + 3] = c.disc; org[slot] = (unsigned short)tid; }
    __syncthreads();
    if (tid < tot1 || tid >= 256 - tot2) {
      pf_cubic d; d.a1 = P[4 * tid]; d.q = P[4 * tid + 1]; d.r = P[4 * tid + 2]; d.disc = P[4 * tid + 3];
      const double e = tid < tot1 ? pf_ell_one_root<true>(d) : pf_ell_three_roots<true>(d);
      E[org[tid]] = e;
    }
    __syncthreads();
    if (kind > 0) ell = E[tid];
    double F = -10.0;
    if (ok) {
      ell = pf_ell_finish<true>(ell, lam[0], lam[1], lam[2]);
      F = ell > 0.0 ? 1. + pf_inverse_growing_mode<true>(sv, ell) : 0.0;
    }
    acc += F;
  }
  out[(size_t)blockIdx.x * blockDim.x + tid] = acc;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)1 << 27;
  const double sigma = argc > 2 ? atof(argv[2]) : 1.0;
  std::vector<double> h(3 * n);
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g(0.0, 1.0);
  // Hessian of a Gaussian field: d_ab with <d_aa^2> = 3 s, <d_aa d_bb> = s, <d_ab^2> = s (s = sigma^2 / 15) -- Doroshkevich
  const double s15 = sigma / sqrt(15.0);
  size_t one = 0, three = 0;
  for (size_t i = 0; i < n; i++) {
    const double u = g(rng), v = g(rng), w = g(rng);
    double d[6];
    const double tr = sigma * u;  // trace
    // traceless diagonal part with variance 2 s * (2/3) per ... simple construction: independent Gaussians, good enough for a timing mix
    const double a = s15 * sqrt(2.0) * v, b = s15 * sqrt(2.0) * w;
    d[0] = tr / 3 + a + b / sqrt(3.0); d[1] = tr / 3 - a + b / sqrt(3.0); d[2] = tr / 3 - 2 * b / sqrt(3.0);
    d[3] = s15 * g(rng); d[4] = s15 * g(rng); d[5] = s15 * g(rng);
    double m1, m2, m3;
    pf_invariants(d, m1, m2, m3);
    h[i] = m1; h[n + i] = m2; h[2 * n + i] = m3;
  }
  // a smooth monotone inverse-growth table: log10 a against log10 D (EdS-like: a = D), 300 knots
  const int nk = 300;
  std::vector<double> x(nk), y(nk), c(nk), b(nk), dd(nk);
  for (int i = 0; i < nk; i++) { x[i] = -3.0 + 4.0 * i / (nk - 1); y[i] = x[i] * (1.0 + 0.02 * sin(x[i])); }
  pf_spline_coeffs(x.data(), y.data(), nk, c.data());
  pf_spline_bd(x.data(), y.data(), c.data(), nk, b.data(), dd.data());
  double *dmu, *dout, *dx, *dy, *dc, *db, *ddd;
  hipMalloc(&dmu, 3 * n * 8); hipMemcpy(dmu, h.data(), 3 * n * 8, hipMemcpyHostToDevice);
  hipMalloc(&dx, nk * 8); hipMalloc(&dy, nk * 8); hipMalloc(&dc, nk * 8); hipMalloc(&db, nk * 8); hipMalloc(&ddd, nk * 8);
  hipMemcpy(dx, x.data(), nk * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), nk * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), nk * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), nk * 8, hipMemcpyHostToDevice);
  hipMemcpy(ddd, dd.data(), nk * 8, hipMemcpyHostToDevice);
  pf_spline_view sv; sv.x = dx; sv.y = dy; sv.c = dc; sv.b = db; sv.d = ddd; sv.n = nk;
  const int grid = 256 * 8;
  hipMalloc(&dout, (size_t)grid * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[14] = {"loads only", "+ eigenvalues", "+ ell setup", "+ one-root branch alone", "+ three-root branch alone", "+ both branches", "+ finish (exp)", "+ inverse growth (log10, spline, 10^)",
                           "  ... bisection instead of the start table", "  probe: log10 and 10^ only, no spline", "  probe: start table, no walk", "  probe: start table + two branch-free steps",
                           "  probe: start table + walk loop", "  probe: two steps, packed records"};
  double prev = 0;
#define RUN(S)                                                                                         \
  {                                                                                                    \
    hipLaunchKernelGGL(k_part<S>, dim3(grid), dim3(256), 0, 0, dmu, n, sv, dout);                      \
    hipDeviceSynchronize();                                                                            \
    hipEventRecord(e0);                                                                                \
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_part<S>, dim3(grid), dim3(256), 0, 0, dmu, n, sv, dout); \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                       \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                        \
    const double ps = ms / 5 * 1e9 / (double)n;                                                        \
    printf("%-40s %8.3f ms per 2^30 cells   (%+.3f)\n", names[S], ps * 1.073741824e9 / 1e9, (ps - prev) * 1.073741824);   \
    prev = ps;                                                                                         \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
  {
    hipLaunchKernelGGL(k_regroup, dim3(grid), dim3(256), 0, 0, dmu, n, sv, dout);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_regroup, dim3(grid), dim3(256), 0, 0, dmu, n, sv, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms per 2^30 cells\n", "probe: branches regrouped in the workgroup", ms / 5 * 1e9 / (double)n * 1.073741824e9 / 1e9);
  }
  return 0;
}
