// pf_strided_addr.h -- element addresses and work-item order of the strided passes (shared by their translation units)
#pragma once
#include "pf_internal.h"
#include "pf_fft_core.h"

__device__ __forceinline__ long long pf_addr(const PfAddr &a, int outer, int e, int col) {
  return (long long)outer * a.os + (long long)(e >> a.el_shift) * a.ehs + (long long)(e & ((1 << a.el_shift) - 1)) * a.els + col;
}

// The same address split into a part that is the same for every lane of a workgroup and a 32-bit part per lane, for the
// element e = tl + m * NT of a thread (m = 0..7 a constant after unrolling).  When a slab holds whole multiples of NT elements
// (el = 1 << el_shift >= NT: up to eight ranks), e >> el_shift and e & (el - 1) follow from m alone:
//   m = q r + s, r = el / NT:  e >> el_shift = q,  e & (el - 1) = tl + s NT
// so that  address = [outer os + q ehs + s NT els + tile's first column]  +  [tl els + column in the tile].
// The first bracket lives in scalar registers (scalar unit, 64-bit), the second is one 32-bit register per thread and job, and a
// load or store takes them as they are (global_load ... v_offset, s[base]) -- as one 64-bit address per element the eight
// addresses of a job cost ~110 vector instructions, among them 24 quarter-rate 64-bit multiply-adds, a fifth of the job's
// vector work (which is what bounds the pass: profiles/r04_notes.md).
template <int NT> __device__ __forceinline__ long long pf_addr_uniform(const PfAddr &a, int outer, int m, int col0) {
  const int rs = a.el_shift - pf_ilog2(NT);
  const int q = m >> rs, s = m & ((1 << rs) - 1);
  return (long long)outer * a.os + (long long)q * a.ehs + (long long)(s * NT) * a.els + col0;
}
__device__ __forceinline__ unsigned pf_addr_lane(const PfAddr &a, int tl, int c) { return (unsigned)tl * (unsigned)a.els + (unsigned)c; }

// XCD-aware work-item id: hardware deals consecutive workgroups round-robin over the 8 XCDs
// (speed only, never correctness); give each XCD a contiguous range of tiles so that
// neighbouring tiles, which share 128-byte lines when T*sizeof(complex) < 128, meet in one L2.
__device__ __forceinline__ long long pf_xcd_swizzle(long long b, long long per_xcd) { return (b & 7) * per_xcd + (b >> 3); }
