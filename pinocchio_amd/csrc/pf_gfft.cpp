// pf_gfft.cpp -- library 3-D transforms for grid sizes that are not a power of two (the "general path" of
// pf_api.hip).  The hand-written Stockham passes cover N = 2^k, which is every GPU-targeted configuration; the
// reference itself takes any GridSize through FFTW/PFFT (e.g. the 200^3 of INSTALLATION:101), so those sizes go
// through hipFFT's double-precision c2r / r2c here (the north-star's "rocFFT (or hand-rolled Stockham)").
// hipFFT is bound at run time so that libpinfmax_hip.so has no load-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>

#include "pf_internal.h"

typedef struct hipfftHandle_t *hipfftHandle;
enum { HIPFFT_D2Z_ = 0x6a, HIPFFT_Z2D_ = 0x6c };

static struct {
  void *h;
  int (*Plan3d)(hipfftHandle *, int, int, int, int);
  int (*SetStream)(hipfftHandle, hipStream_t);
  int (*ExecZ2D)(hipfftHandle, void *, double *);
  int (*ExecD2Z)(hipfftHandle, double *, void *);
  int (*Destroy)(hipfftHandle);
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
  SYM(Plan3d, "hipfftPlan3d") SYM(SetStream, "hipfftSetStream") SYM(ExecZ2D, "hipfftExecZ2D") SYM(ExecD2Z, "hipfftExecD2Z") SYM(Destroy, "hipfftDestroy")
#undef SYM
  g_fft.h = h;
  return 0;
}

int pf_gfft_create(int n, hipStream_t st, void **c2r, void **r2c) {
  if (load_hipfft()) return 1;
  hipfftHandle a = nullptr, b = nullptr;
  if (g_fft.Plan3d(&a, n, n, n, HIPFFT_Z2D_) || g_fft.Plan3d(&b, n, n, n, HIPFFT_D2Z_)) return 2;
  if (g_fft.SetStream(a, st) || g_fft.SetStream(b, st)) return 3;
  *c2r = a; *r2c = b;
  return 0;
}
int pf_gfft_c2r(void *plan, void *spec, void *real) { return g_fft.ExecZ2D((hipfftHandle)plan, spec, (double *)real); }
int pf_gfft_r2c(void *plan, void *real, void *spec) { return g_fft.ExecD2Z((hipfftHandle)plan, (double *)real, spec); }
void pf_gfft_destroy(void *plan) { if (plan && g_fft.Destroy) g_fft.Destroy((hipfftHandle)plan); }
