// handoff_probe.hip -- what the host side of the boundary can do on this box (round 6): rates of the ways
// products can leave HBM for the caller's (pageable) product_data array, so that pf_get_products / pf_update_products are
// built on measured numbers.  hipcc -O2 -o bin/handoff_probe handoff_probe.hip -lpthread ; prints one JSON line per probe.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at %s:%d\"}\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <class F> static void par(int nt, size_t n, F f) {
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back([=]() { const size_t a = n * t / nt, b = n * (t + 1) / nt; f(a, b); });
  for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
  const size_t GB = 1ull << 30;
  const size_t big = (argc > 1 ? atol(argv[1]) : 8) * GB;   // the caller's array
  const size_t chunk = 256ull << 20;
  const int hw = (int)std::thread::hardware_concurrency();
  printf("{\"probe\": \"host\", \"hardware_concurrency\": %d}\n", hw);
  char *dev; CHK(hipMalloc((void **)&dev, big)); CHK(hipMemset(dev, 1, big));
  hipStream_t st[2]; CHK(hipStreamCreate(&st[0])); CHK(hipStreamCreate(&st[1]));
  char *pin[2]; CHK(hipHostMalloc((void **)&pin[0], chunk)); CHK(hipHostMalloc((void **)&pin[1], chunk));
  // 1. pageable target, one hipMemcpy: first touch, then again
  char *host = (char *)aligned_alloc(4096, big);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); CHK(hipMemcpy(host, dev, big, hipMemcpyDeviceToHost)); double t1 = now();
    printf("{\"probe\": \"d2h_pageable_one_call\", \"rep\": %d, \"GB\": %.1f, \"GBps\": %.2f}\n", rep, big / 1e9, big / 1e9 / (t1 - t0));
  }
  // 2. pageable target in chunks of 256 MB as pf_get_products issues them
  {
    double t0 = now();
    for (size_t o = 0; o < big; o += chunk) { CHK(hipMemcpyAsync(host + o, dev + o, chunk, hipMemcpyDeviceToHost, st[0])); CHK(hipStreamSynchronize(st[0])); }
    double t1 = now();
    printf("{\"probe\": \"d2h_pageable_chunks_sync\", \"GBps\": %.2f}\n", big / 1e9 / (t1 - t0));
  }
  // 3. pinned staging + host threads copying into the pageable target, double-buffered
  for (int nt : {4, 8, 16, 32}) {
    if (nt > 2 * hw) continue;
    double t0 = now();
    size_t k = 0;
    std::thread worker;
    for (size_t o = 0; o < big; o += chunk, k++) {
      const int b = k & 1;
      CHK(hipMemcpyAsync(pin[b], dev + o, chunk, hipMemcpyDeviceToHost, st[b]));
      if (worker.joinable()) worker.join();       // the copy-out of chunk k-1 (other buffer) -- must end before chunk k+1 lands there
      CHK(hipStreamSynchronize(st[b]));
      char *dst = host + o; const char *src = pin[b];
      worker = std::thread([=]() { par(nt, chunk, [=](size_t a, size_t e) { memcpy(dst + a, src + a, e - a); }); });
    }
    if (worker.joinable()) worker.join();
    double t1 = now();
    printf("{\"probe\": \"d2h_pinned_staging_threads\", \"threads\": %d, \"GBps\": %.2f}\n", nt, big / 1e9 / (t1 - t0));
  }
  // 4. host threads alone: pinned -> pageable copy rate, and the 48-of-104-byte scatter of pf_update_products
  for (int nt : {4, 8, 16, 32}) {
    if (nt > 2 * hw) continue;
    double t0 = now();
    for (int r = 0; r < 8; r++) par(nt, chunk, [=](size_t a, size_t e) { memcpy(host + (size_t)r * chunk + a, pin[0] + a, e - a); });
    double t1 = now();
    const size_t ncell = chunk / 48;
    double t2 = now();
    for (int r = 0; r < 8; r++) {
      char *rec = host + (size_t)r * ncell * 104;
      if ((size_t)(r + 1) * ncell * 104 > big) break;
      par(nt, ncell, [=](size_t a, size_t e) { for (size_t i = a; i < e; i++) memcpy(rec + i * 104 + 8, pin[0] + i * 48, 48); });
    }
    double t3 = now();
    printf("{\"probe\": \"host_threads\", \"threads\": %d, \"memcpy_GBps\": %.2f, \"scatter48of104_payload_GBps\": %.2f}\n", nt, 8.0 * chunk / 1e9 / (t1 - t0),
           8.0 * ncell * 48 / 1e9 / (t3 - t2));
  }
  // 5. hipHostRegister of the caller's array: cost, then direct DMA into it
  {
    double t0 = now();
    hipError_t e = hipHostRegister(host, big, hipHostRegisterDefault);
    double t1 = now();
    if (e != hipSuccess) { (void)hipGetLastError(); printf("{\"probe\": \"host_register\", \"failed\": \"%s\"}\n", hipGetErrorString(e)); }
    else {
      printf("{\"probe\": \"host_register\", \"GB\": %.1f, \"seconds\": %.3f, \"GBps\": %.2f}\n", big / 1e9, t1 - t0, big / 1e9 / (t1 - t0));
      for (int rep = 0; rep < 2; rep++) {
        double a = now(); CHK(hipMemcpyAsync(host, dev, big, hipMemcpyDeviceToHost, st[0])); CHK(hipStreamSynchronize(st[0])); double b = now();
        printf("{\"probe\": \"d2h_registered\", \"rep\": %d, \"GBps\": %.2f}\n", rep, big / 1e9 / (b - a));
      }
      {  // 2-D copy: 48 bytes per 104-byte record straight into the registered records
        const size_t ncell = big / 104;
        double a = now(); CHK(hipMemcpy2DAsync(host + 8, 104, dev, 48, 48, ncell > (64u << 20) ? (64u << 20) : ncell, hipMemcpyDeviceToHost, st[0])); CHK(hipStreamSynchronize(st[0])); double b = now();
        const size_t rows = ncell > (64u << 20) ? (64u << 20) : ncell;
        printf("{\"probe\": \"d2h_2d_48_of_104_registered\", \"payload_GBps\": %.2f}\n", rows * 48 / 1e9 / (b - a));
      }
      double c0 = now(); CHK(hipHostUnregister(host)); double c1 = now();
      printf("{\"probe\": \"host_unregister\", \"seconds\": %.3f}\n", c1 - c0);
    }
  }
  // 6. H2D from pageable (pf_set_density) and from pinned staging
  {
    double t0 = now(); CHK(hipMemcpy(dev, host, big, hipMemcpyHostToDevice)); double t1 = now();
    printf("{\"probe\": \"h2d_pageable_one_call\", \"GBps\": %.2f}\n", big / 1e9 / (t1 - t0));
    double t2 = now();
    for (int r = 0; r < 16; r++) { CHK(hipMemcpyAsync(dev + (size_t)r * chunk, pin[r & 1], chunk, hipMemcpyHostToDevice, st[r & 1])); }
    CHK(hipStreamSynchronize(st[0])); CHK(hipStreamSynchronize(st[1]));
    double t3 = now();
    printf("{\"probe\": \"h2d_pinned\", \"GBps\": %.2f}\n", 16.0 * chunk / 1e9 / (t3 - t2));
    double t4 = now();
    for (int r = 0; r < 16; r++) { CHK(hipMemcpyAsync(pin[r & 1], dev + (size_t)r * chunk, chunk, hipMemcpyDeviceToHost, st[r & 1])); }
    CHK(hipStreamSynchronize(st[0])); CHK(hipStreamSynchronize(st[1]));
    double t5 = now();
    printf("{\"probe\": \"d2h_pinned\", \"GBps\": %.2f}\n", 16.0 * chunk / 1e9 / (t5 - t4));
  }
  free(host);
  return 0;
}
