// pf_select_sort.hip -- hand-off of the products to the consumers of the path.
//
//  * pf_select_sorted: the first stage of fragmentation on the device -- keep the cells with Fmax >= Flast
//    (update_distmap, src/distribute.c:695) and order them by descending Fmax (sort_and_organize,
//    src/fragment.c:484-503 with index_compare_F :118-126).  qsort leaves the order of equal keys unspecified; here
//    ties go by ascending cell index, which is one of the orders qsort may produce.
//    One 64-bit key per selected cell = (descending-orderable Fmax bits) << 32 | cell index, appended with one atomic
//    per wavefront, then an LSD radix sort of the keys (rocPRIM, the sort primitive shipped with ROCm).
//  * pf_get_block: the per-particle payloads of the Gadget-2 style "timeless snapshot" (write_timeless_snapshot,
//    src/write_snapshot.c:207-342; initialize_ID/FMAX/RMAX/ZEL/2LPT/3LPT_1/3LPT_2 :620-855) straight from the SoA
//    columns in HBM, without the AoS product_data detour.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "pf_internal.h"

#define PF_SEL_BLOCK 256

__device__ __forceinline__ unsigned int pf_desc_key(float f) {
  unsigned int u = __float_as_uint(f);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending-orderable
  return ~u;                                   // descending
}
__device__ __forceinline__ float pf_key_to_float(unsigned int k) {
  unsigned int u = ~k;
  u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(u);
}

__global__ void __launch_bounds__(PF_SEL_BLOCK) k_count_selected(const float *__restrict__ fmax, size_t ncell, float flast, unsigned long long *count) {
  unsigned long long mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) mine += fmax[i] >= flast;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

__global__ void __launch_bounds__(PF_SEL_BLOCK)
    k_select_keys(const float *__restrict__ fmax, size_t ncell, float flast, unsigned long long *__restrict__ keys, unsigned long long *cursor) {
  const int lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t rounds = (ncell + stride - 1) / stride;  // every lane of a wave runs the same number of rounds (ballot)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t r = 0; r < rounds; r++, i += stride) {
    const bool in = i < ncell;
    const float f = in ? fmax[i] : 0.0f;
    const bool take = in && f >= flast;
    const unsigned long long m = __ballot(take);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
    base = __shfl(base, 0, 64);
    if (take) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)pf_desc_key(f) << 32) | (unsigned int)i;
  }
}

__global__ void __launch_bounds__(PF_SEL_BLOCK)
    k_unpack_keys(const unsigned long long *__restrict__ keys, size_t count, unsigned int *__restrict__ idx, float *__restrict__ f) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    idx[i] = (unsigned int)k;
    f[i] = pf_key_to_float((unsigned int)(k >> 32));
  }
}

static int sel_grid(size_t n) {
  size_t b = (n + PF_SEL_BLOCK - 1) / PF_SEL_BLOCK;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

#define SELCHK(x) do { if ((x) != hipSuccess) { rc = 1; goto done; } } while (0)

// device arrays out: *d_idx / *d_f (hipMalloc'ed here, the caller frees), *count
int pf_select_sort_device(const float *fmax, size_t ncell, float flast, unsigned int **d_idx, float **d_f, size_t *count, hipStream_t st) {
  int rc = 0;
  unsigned long long *d_count = nullptr, *keys_in = nullptr, *keys_out = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  unsigned long long h = 0;
  *d_idx = nullptr; *d_f = nullptr; *count = 0;
  SELCHK(hipMalloc(&d_count, 2 * sizeof(unsigned long long)));
  SELCHK(hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_count_selected, dim3(sel_grid(ncell)), dim3(PF_SEL_BLOCK), 0, st, fmax, ncell, flast, d_count);
  SELCHK(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, st));
  SELCHK(hipStreamSynchronize(st));
  *count = (size_t)h;
  if (h) {
    SELCHK(hipMalloc(&keys_in, h * sizeof(unsigned long long)));
    SELCHK(hipMalloc(&keys_out, h * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_select_keys, dim3(sel_grid(ncell)), dim3(PF_SEL_BLOCK), 0, st, fmax, ncell, flast, keys_in, d_count + 1);
    SELCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_in, keys_out, (size_t)h, 0, 64, st));
    SELCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    SELCHK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys_in, keys_out, (size_t)h, 0, 64, st));
    // keys_in is dead: the two output arrays fit in it (8 bytes per entry)
    *d_idx = (unsigned int *)keys_in;
    *d_f = (float *)((unsigned int *)keys_in + h);
    hipLaunchKernelGGL(k_unpack_keys, dim3(sel_grid(h)), dim3(PF_SEL_BLOCK), 0, st, keys_out, (size_t)h, *d_idx, *d_f);
    SELCHK(hipGetLastError());
    SELCHK(hipStreamSynchronize(st));
    keys_in = nullptr;  // now owned by the caller through *d_idx
  }
done:
  hipFree(d_count); hipFree(keys_in); hipFree(keys_out); hipFree(tmp);
  if (rc) { hipFree(*d_idx); *d_idx = nullptr; *d_f = nullptr; }
  return rc;
}

// ------------------------------------------------------------------------------------------- snapshot blocks ----
// vector blocks: AuxStruct {float axis[3]} per particle (src/write_snapshot.c:745-760)
__global__ void __launch_bounds__(PF_SEL_BLOCK)
    k_block_vec3(const float *__restrict__ vel12, size_t ncell, int o, size_t first, size_t count, float *__restrict__ out) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < 3 * count; j += (size_t)gridDim.x * blockDim.x) {
    const size_t p = j / 3;
    const int k = (int)(j - 3 * p);
    out[j] = vel12[(size_t)(3 * o + k) * ncell + first + p];
  }
}
// ID = 1 + global cell index, coherent with INDEX_TO_COORD (src/write_snapshot.c:648-664)
template <typename ID>
__global__ void __launch_bounds__(PF_SEL_BLOCK) k_block_id(unsigned long long global_first, size_t count, ID *__restrict__ out) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) out[j] = (ID)(1ull + global_first + j);
}
int pf_launch_block_vec3(const float *vel12, size_t ncell, int o, size_t first, size_t count, float *out, hipStream_t st) {
  hipLaunchKernelGGL(k_block_vec3, dim3(sel_grid(3 * count)), dim3(PF_SEL_BLOCK), 0, st, vel12, ncell, o, first, count, out);
  return hipGetLastError() != hipSuccess;
}
int pf_launch_block_id(int id_bytes, unsigned long long global_first, size_t count, void *out, hipStream_t st) {
  if (id_bytes == 8) hipLaunchKernelGGL(k_block_id<unsigned long long>, dim3(sel_grid(count)), dim3(PF_SEL_BLOCK), 0, st, global_first, count, (unsigned long long *)out);
  else hipLaunchKernelGGL(k_block_id<unsigned int>, dim3(sel_grid(count)), dim3(PF_SEL_BLOCK), 0, st, global_first, count, (unsigned int *)out);
  return hipGetLastError() != hipSuccess;
}
