// pf_gfft.cpp -- library 3-D transforms for grid sizes that are not a power of two (the "general path" of
// pf_api.hip).  The hand-written Stockham passes cover N = 2^k, which is every GPU-targeted configuration; the
// reference itself takes any GridSize through FFTW/PFFT (e.g. the 200^3 of INSTALLATION:101), so those sizes go
// through hipFFT's double-precision c2r / r2c here (the north-star's "rocFFT (or hand-rolled Stockham)").
// hipFFT is bound at run time so that libpinfmax_hip.so has no load-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>

#include "pf_internal.h"

// types, transform codes and prototypes from the hipFFT header of the build (never copied by hand); the library bound at run
// time must report the same major version
#include <hipfft/hipfft.h>
#include <hipfft/hipfft-version.h>

static struct {
  void *h;
  decltype(&hipfftGetVersion) GetVersion;
  decltype(&hipfftPlan3d) Plan3d;
  decltype(&hipfftSetStream) SetStream;
  decltype(&hipfftExecZ2D) ExecZ2D;
  decltype(&hipfftExecD2Z) ExecD2Z;
  decltype(&hipfftDestroy) Destroy;
} g_fft = {};

static int load_hipfft() {
  if (g_fft.h) return 0;
  const char *names[] = {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) { printf("ERROR on task 0: cannot load libhipfft (%s)\n", dlerror()); return 1; }
#define SYM(field, name) *(void **)(&g_fft.field) = dlsym(h, name); if (!g_fft.field) { printf("ERROR on task 0: missing %s in libhipfft\n", name); return 1; }
  SYM(GetVersion, "hipfftGetVersion")
  SYM(Plan3d, "hipfftPlan3d") SYM(SetStream, "hipfftSetStream") SYM(ExecZ2D, "hipfftExecZ2D") SYM(ExecD2Z, "hipfftExecD2Z") SYM(Destroy, "hipfftDestroy")
#undef SYM
  int v = 0;
  if (g_fft.GetVersion(&v) != HIPFFT_SUCCESS || (v / 10000 != hipfftVersionMajor && v / 1000 != hipfftVersionMajor)) {  // (major * 10000 + minor * 100 + patch)
    printf("ERROR on task 0: libhipfft at run time reports version %d, libpinfmax_hip was built against major %d: refusing to bind it\n", v, hipfftVersionMajor);
    g_fft = {};
    return 1;
  }
  g_fft.h = h;
  return 0;
}

int pf_gfft_create(int n, hipStream_t st, void **c2r, void **r2c) {
  if (load_hipfft()) return 1;
  hipfftHandle a = nullptr, b = nullptr;
  if (g_fft.Plan3d(&a, n, n, n, HIPFFT_Z2D) != HIPFFT_SUCCESS || g_fft.Plan3d(&b, n, n, n, HIPFFT_D2Z) != HIPFFT_SUCCESS) return 2;
  if (g_fft.SetStream(a, st) != HIPFFT_SUCCESS || g_fft.SetStream(b, st) != HIPFFT_SUCCESS) return 3;
  *c2r = a; *r2c = b;
  return 0;
}
int pf_gfft_c2r(void *plan, void *spec, void *real) { return g_fft.ExecZ2D((hipfftHandle)plan, (hipfftDoubleComplex *)spec, (double *)real) != HIPFFT_SUCCESS; }
int pf_gfft_r2c(void *plan, void *real, void *spec) { return g_fft.ExecD2Z((hipfftHandle)plan, (double *)real, (hipfftDoubleComplex *)spec) != HIPFFT_SUCCESS; }
void pf_gfft_destroy(void *plan) { if (plan && g_fft.Destroy) g_fft.Destroy((hipfftHandle)plan); }
