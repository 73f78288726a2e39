// hbm_probe.hip -- what this memory system gives kernels that only move bytes (round 4).
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe profiles/tools/hbm_probe.hip && ./hbm_probe > r04_hbm_ceilings.jsonl
//
// Two families, one JSON object per line:
//   "lin"  : read / write / copy of a flat array, 16 bytes per lane; swept over loads in flight per lane (unroll 1..8),
//            workgroup size, workgroups per CU, plain vs non-temporal accesses, and the walk (grid-stride over the whole
//            array, or one contiguous range per XCD: hardware deals consecutive workgroups round-robin over the 8 XCDs).
//   "tile" : the access shape of the strided transform passes without their arithmetic (k_strided, pf_fft_kernels.hip):
//            a 1024-thread workgroup holding 128 KB of LDS (one per CU) moves tiles of 1024 rows x 128 bytes; every wave
//            instruction touches 8 rows x 128 B; NIN tiles in, NOUT tiles out, at the library's strides for the x-pass
//            (1 -> 3), the y-pass (3 -> 6) and the displacement passes (1 -> 2, 2 -> 3), with the library's order of
//            stores and with alternatives.
// Bytes are algorithmic bytes (every element once); rates in TB/s = 1e12 B/s; best of REPS timed launches by HIP events.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// KIND 0 read, 1 write, 2 copy.  WALK 0: grid-stride (element i of iteration k: base + k * threads * U + u * threads);
// WALK 1: XCD-contiguous (workgroup b works in the eighth of the array that belongs to XCD b % 8, grid-stride inside it).
template <int KIND, int U, bool NTL, bool NTS, int WALK>
__global__ void k_lin(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n, float *sink) {
  size_t lo = 0, hi = n, nthr = (size_t)gridDim.x * blockDim.x, me = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (WALK == 1) {
    const size_t per = n >> 3;
    const int x = blockIdx.x & 7;
    lo = per * x; hi = lo + per;
    nthr = (size_t)(gridDim.x >> 3) * blockDim.x;
    me = (size_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const f4 one = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = lo + me; i + (U - 1) * nthr < hi; i += nthr * U) {
    f4 v[U];
    if (KIND != 1) {
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = ld<NTL>(a + i + u * nthr);
    }
    if (KIND == 0) {
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) st<NTS>(b + i + u * nthr, KIND == 1 ? one : v[u]);
    }
  }
  if (KIND == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}

struct TileShape {
  int nin, nout;
  long long in_es, in_os, out_es, out_os;  // strides in 16-byte elements: along the transformed axis (e) and the outer axis
  int ntiles, nouter;                      // tiles of 8 columns per line of tiles, lines of tiles
  long long field;                         // elements per field (inputs and outputs are separate fields)
};
// ORDER 0: the library's order -- for each output, m = 0..7 (rows tl + 128 m);
// ORDER 1: m outermost, outputs inside (a row's segments of all outputs together);
// ORDER 2: as 0, rows remapped so that a thread's 8 stores go to 8 CONSECUTIVE rows (e = 8 tl + m): a wave then covers 64
//          consecutive rows of 128 B over its 8 instructions instead of 8 rows 128 apart per instruction.
// spin: dependent fp64 fma per thread and job between the loads and the stores, in six pieces with a workgroup barrier behind
// each (the arithmetic and the LDS exchanges of a 1024-point transform: ~500 fp64 instructions per job); fence: 1 = an
// s_waitcnt vmcnt(0) behind the first piece of every job (what a global twiddle load in the job loop amounts to)
template <bool NTL, bool NTS, int ORDER>
__global__ void __launch_bounds__(1024) k_tile(const f4 *__restrict__ in, f4 *__restrict__ out, TileShape s, long long nwork, int spin, int fence) {
  extern __shared__ char smem[];  // 128 KB: one workgroup per CU, as the passes
  const long long per = (nwork + 7) >> 3;
  const long long w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= nwork) return;
  const int tid = threadIdx.x, c = tid & 7, tl = tid >> 3;
  const int tile = (int)(w % s.ntiles), outer = (int)(w / s.ntiles);
  const long long col = tile * 8 + c;
  f4 src[8];
  auto load = [&](int j) {
    const f4 *p = in + (long long)j * s.field + outer * s.in_os + col;
#pragma unroll
    for (int m = 0; m < 8; m++) src[m] = ld<NTL>(p + (long long)(tl + m * 128) * s.in_es);
  };
  load(0);
  const int opi = s.nout / s.nin;  // outputs per input
  for (int j = 0; j < s.nin; j++) {
    f4 v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = src[m];
    if (j + 1 < s.nin) load(j + 1);
    // stand-in for the LDS exchanges: the stores depend on every load of the tile
    reinterpret_cast<f4 *>(smem)[tid] = v[0];
    __syncthreads();
    v[0] = reinterpret_cast<f4 *>(smem)[tid ^ 1];
    __syncthreads();
    if (spin > 0) {
      double a = (double)v[0].x, b = 1.0000001;
      for (int piece = 0; piece < 6; piece++) {
        for (int i = 0; i < spin / 6; i++) a = __builtin_fma(a, b, 1e-9);
        if (piece == 0 && fence) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) (gfx9 encoding: vmcnt low bits 3:0 and 15:14 = 0, others max)
        __syncthreads();
      }
      v[0].x = (float)a;
    }
    if (ORDER == 1) {
#pragma unroll
      for (int m = 0; m < 8; m++)
        for (int o = 0; o < opi; o++) {
          f4 *q = out + (long long)(j * opi + o) * s.field + outer * s.out_os + col;
          st<NTS>(q + (long long)(tl + m * 128) * s.out_es, v[m]);
        }
    } else {
      for (int o = 0; o < opi; o++) {
        f4 *q = out + (long long)(j * opi + o) * s.field + outer * s.out_os + col;
#pragma unroll
        for (int m = 0; m < 8; m++) st<NTS>(q + (long long)(ORDER == 2 ? 8 * tl + m : tl + m * 128) * s.out_es, v[m]);
      }
    }
  }
}

static double time_best(int reps, hipStream_t st, const std::function<void()> &launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipStreamSynchronize(st));
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(a, st));
    launch();
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return best;
}

template <int KIND, int U, bool NTL, bool NTS, int WALK>
static void run_lin(const f4 *a, f4 *b, size_t n, float *sink, int ncu, hipStream_t st) {
  const char *kn[3] = {"read", "write", "copy"};
  for (int block : {256, 1024})
    for (int per_cu : {1, 2, 4, 8, 16}) {
      if (block == 1024 && per_cu > 8) continue;
      const int grid = ncu * per_cu;
      const double ms = time_best(3, st, [&]() { hipLaunchKernelGGL((k_lin<KIND, U, NTL, NTS, WALK>), dim3(grid), dim3(block), 0, st, a, b, n, sink); });
      const double bytes = (double)n * 16.0 * (KIND == 2 ? 2.0 : 1.0);
      printf("{\"family\": \"lin\", \"kind\": \"%s\", \"unroll\": %d, \"nt_load\": %d, \"nt_store\": %d, \"walk\": \"%s\", \"block\": %d, \"wg_per_cu\": %d, "
             "\"ms\": %.4f, \"TBps\": %.3f}\n",
             kn[KIND], U, (int)NTL, (int)NTS, WALK ? "xcd" : "grid", block, per_cu, ms, bytes / ms * 1e-9);
      fflush(stdout);
    }
}
template <int KIND, int U>
static void run_lin_nt(const f4 *a, f4 *b, size_t n, float *sink, int ncu, hipStream_t st) {
  run_lin<KIND, U, false, false, 0>(a, b, n, sink, ncu, st);
  run_lin<KIND, U, false, false, 1>(a, b, n, sink, ncu, st);
  run_lin<KIND, U, true, true, 0>(a, b, n, sink, ncu, st);
  run_lin<KIND, U, true, true, 1>(a, b, n, sink, ncu, st);
  if (KIND == 2) {  // mixed: streaming loads with cached stores and the reverse
    run_lin<KIND, U, true, false, 1>(a, b, n, sink, ncu, st);
    run_lin<KIND, U, false, true, 1>(a, b, n, sink, ncu, st);
  }
}
template <int KIND>
static void run_lin_all(const f4 *a, f4 *b, size_t n, float *sink, int ncu, hipStream_t st) {
  run_lin_nt<KIND, 1>(a, b, n, sink, ncu, st);
  run_lin_nt<KIND, 2>(a, b, n, sink, ncu, st);
  run_lin_nt<KIND, 4>(a, b, n, sink, ncu, st);
  run_lin_nt<KIND, 8>(a, b, n, sink, ncu, st);
}

template <bool NTL, bool NTS, int ORDER>
static void run_tile(const char *name, const TileShape &s, const f4 *in, f4 *out, hipStream_t st, int spin = 0, int fence = 0) {
  const long long nwork = (long long)s.ntiles * s.nouter;
  const unsigned grid = (unsigned)(((nwork + 7) >> 3) << 3);
  const size_t shm = 128 * 1024;
  static bool raised = false;
  if (!raised) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<NTL, NTS, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); raised = true; }
  const double ms = time_best(3, st, [&]() { hipLaunchKernelGGL((k_tile<NTL, NTS, ORDER>), dim3(grid), dim3(1024), shm, st, in, out, s, nwork, spin, fence); });
  const double bytes = (double)nwork * 1024.0 * 128.0 * (s.nin + s.nout);
  printf("{\"family\": \"tile\", \"shape\": \"%s\", \"nin\": %d, \"nout\": %d, \"nt_load\": %d, \"nt_store\": %d, \"order\": %d, \"spin\": %d, \"fence\": %d, \"ms\": %.4f, \"TBps\": %.3f, "
         "\"read_GB\": %.2f, \"write_GB\": %.2f}\n",
         name, s.nin, s.nout, (int)NTL, (int)NTS, ORDER, spin, fence, ms, bytes / ms * 1e-9, (double)nwork * 131072.0 * s.nin * 1e-9, (double)nwork * 131072.0 * s.nout * 1e-9);
  fflush(stdout);
}
template <int ORDER>
static void run_tile_nt(const char *name, const TileShape &s, const f4 *in, f4 *out, hipStream_t st) {
  run_tile<true, true, ORDER>(name, s, in, out, st);
  run_tile<false, false, ORDER>(name, s, in, out, st);
  run_tile<true, false, ORDER>(name, s, in, out, st);
  run_tile<false, true, ORDER>(name, s, in, out, st);
}

int main(int argc, char **argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const bool tiles_only = argc > 1 && !strcmp(argv[1], "tiles");
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  fprintf(stderr, "%s, %d CUs\n", prop.name, ncu);
  // ---- flat arrays: 4 GiB each (16 x the Infinity Cache)
  if (!tiles_only) {
    const size_t bytes = quick ? (size_t)1 << 30 : (size_t)4 << 30, n = bytes / 16;
    f4 *a, *b; float *sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
    run_lin_all<0>(a, b, n, sink, ncu, st);
    run_lin_all<1>(a, b, n, sink, ncu, st);
    run_lin_all<2>(a, b, n, sink, ncu, st);
    CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(sink));
  }
  // ---- the passes' shapes at 1024^3 fp64: rows of nzp = 520 complex (8320 B), 65 tiles of 8 columns per row,
  // planes of 1024 rows (8,519,680 B), fields of 1024 planes
  {
    const long long zp = 520, n = quick ? 256 : 1024, plane = n * zp, field = n * plane;
    f4 *in, *out;
    const size_t slack = (size_t)8 << 21;  // room for the field pads below
    CHECK(hipMalloc(&in, (size_t)3 * field * 16 + slack)); CHECK(hipMalloc(&out, (size_t)6 * field * 16 + slack));
    CHECK(hipMemset(in, 0, (size_t)3 * field * 16 + slack)); CHECK(hipMemset(out, 0, (size_t)6 * field * 16 + slack));
    fprintf(stderr, "in %p out %p\n", (void *)in, (void *)out);
    TileShape x13 = {1, 3, plane, zp, zp, n * zp, 65, (int)n, field};   // x-pass: reads [kx][ky][kz] along kx, writes [ky][x][kz]
    TileShape y36 = {3, 6, n * zp, zp, zp, n * zp, 65, (int)n, field};  // y-pass: reads [ky][x][kz] along ky (plane stride), writes [x][y][kz]
    TileShape x12 = x13; x12.nout = 2;
    TileShape y23 = y36; y23.nin = 3; y23.nout = 3;  // (2 -> 3 in the library; 3 -> 3 here: one output per input)
    TileShape y11 = y36; y11.nin = 1; y11.nout = 1;
    // the same with the unit-stride side only (reads or writes at the row stride on both sides): what the plane stride costs
    TileShape y36r = y36; y36r.in_es = zp; y36r.in_os = n * zp;
    if (n != 1024) { fprintf(stderr, "quick: tile kernels need 1024 rows; skipped\n"); }
    else {
      run_tile_nt<0>("xpass_1to3", x13, in, out, st);
      run_tile_nt<0>("ypass_3to6", y36, in, out, st);
      run_tile_nt<1>("xpass_1to3", x13, in, out, st);
      run_tile_nt<1>("ypass_3to6", y36, in, out, st);
      run_tile_nt<2>("xpass_1to3", x13, in, out, st);
      run_tile_nt<2>("ypass_3to6", y36, in, out, st);
      run_tile<true, true, 0>("xpass_1to2", x12, in, out, st);
      run_tile<true, true, 0>("ypass_3to3", y23, in, out, st);
      run_tile<true, true, 0>("ypass_1to1", y11, in, out, st);
      run_tile<true, true, 0>("ypass_3to6_rowstride_in", y36r, in, out, st);
      run_tile<true, true, 2>("ypass_3to6_rowstride_in", y36r, in, out, st);
      // the same shapes with arithmetic between the loads and the stores, with and without a fence in every job
      for (int spin : {240, 480, 720})
        for (int fence : {0, 1}) {
          run_tile<true, true, 0>("xpass_1to3", x13, in, out, st, spin, fence);
          run_tile<true, true, 0>("ypass_3to6", y36, in, out, st, spin, fence);
        }
      // fields whose bases are not congruent modulo a large power of two (the library's fields are 65 x 2^27 bytes each)
      for (long long pad_bytes : {0LL, 4096LL, 69632LL, 1052672LL, 2101248LL}) {
        TileShape yp = y36, xp = x13;
        yp.field = field + pad_bytes / 16; xp.field = field + pad_bytes / 16;
        char nm[64];
        snprintf(nm, sizeof(nm), "ypass_3to6_fieldpad_%lld", pad_bytes);
        run_tile<true, true, 0>(nm, yp, in, out, st);
        snprintf(nm, sizeof(nm), "xpass_1to3_fieldpad_%lld", pad_bytes);
        run_tile<true, true, 0>(nm, xp, in, out, st);
      }
    }
    CHECK(hipFree(in)); CHECK(hipFree(out));
  }
  return 0;
}
