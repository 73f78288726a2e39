// pf_fabric.hip -- in-process exchange between P contexts that share ONE GPU.
//
// A bring-up / test transport (what MPI calls a "self" BTL): P host threads,
// each driving its own pf_ctx (rank r of P) on the same device, meet in the
// all-to-all and the all-reduce and move the blocks with device-to-device copies.
// It exercises exactly the slab code path of an 8-GPU run (KY/XS layouts, block
// addressing, wavenumber offsets, pack-free send blocks, the pipelined exchange on
// its communication stream) on a 1-GPU box; production runs use pf_init_rccl.
//
// The all-to-all is ASYNCHRONOUS on the device, like RCCL's: the host threads meet only to
// publish their send pointers and events (no stream is ever synchronised); every rank makes
// its stream wait for the senders' "blocks complete" events, pulls its blocks with copies
// on that stream, records a "pulled" event, and makes its stream wait for the pullers of
// its own send buffer.  What the caller sees is the RCCL contract: once the operation has
// completed in stream order the receive buffer is filled and the send buffer reusable.
// So an exchange enqueued on the communication stream really runs beside the kernels of the
// compute stream, and a missing event wait in the pipeline (pf_api.hip: pipelined_band)
// shows up as wrong results -- pf_fabric_set_delay widens the window for the tests.
#include <hip/hip_runtime.h>

#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/pinfmax.h"

struct pf_fabric {
  int P;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  std::vector<const void *> send;
  std::vector<hipEvent_t> ready, pulled;  // per rank: my blocks are complete / I have pulled my blocks
  std::vector<void *> buf;
  int error = 0;
  int delay_us = 0;                        // test knob: the communication stream idles this long before it pulls
  unsigned long long exchanges = 0;
};

struct FabricLink { pf_fabric *f; int rank; };

static void fabric_barrier(pf_fabric *f) {
  std::unique_lock<std::mutex> lk(f->mu);
  const unsigned long long gen = f->generation;
  if (++f->arrived == f->P) {
    f->arrived = 0;
    f->generation++;
    f->cv.notify_all();
  } else {
    f->cv.wait(lk, [&] { return f->generation != gen; });
  }
}

// one thread that watches the constant-rate counter (100 MHz on gfx950) for `us` microseconds
__global__ void k_fabric_idle(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// common part: rendezvous, pull my row range of every peer's block for me, hand the send buffers back
static int fabric_exchange(FabricLink *l, const void *send, void *recv, size_t block_bytes, const size_t *recv_off, const size_t *recv_bytes,
                           hipStream_t st) {
  pf_fabric *f = l->f;
  const int r = l->rank;
  if (hipEventRecord(f->ready[r], st) != hipSuccess) f->error = 1;  // stream order: after whatever filled my send blocks
  f->send[r] = send;
  fabric_barrier(f);  // host only: pointers and events of this exchange are published
  if (f->delay_us > 0) hipLaunchKernelGGL(k_fabric_idle, dim3(1), dim3(1), 0, st, (long long)f->delay_us * 100);
  for (int p = 0; p < f->P; p++) {
    const size_t off = recv_off ? recv_off[p] : 0, len = recv_bytes ? recv_bytes[p] : block_bytes;
    if (!len) continue;
    if (p != r && hipStreamWaitEvent(st, f->ready[p], 0) != hipSuccess) f->error = 1;
    if (hipMemcpyAsync((char *)recv + (size_t)p * block_bytes + off, (const char *)f->send[p] + (size_t)r * block_bytes + off, len,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
      f->error = 1;
  }
  if (hipEventRecord(f->pulled[r], st) != hipSuccess) f->error = 1;
  fabric_barrier(f);  // host only: every rank has enqueued its pulls and recorded its event
  for (int p = 0; p < f->P; p++)  // my send buffer is free again once every peer has pulled from it
    if (p != r && hipStreamWaitEvent(st, f->pulled[p], 0) != hipSuccess) f->error = 1;
  // the events are re-recorded by the next exchange: every wait on this round's records must be enqueued first
  fabric_barrier(f);
  if (r == 0) f->exchanges++;
  return f->error;
}

static int fabric_alltoall(void *user, const void *send, void *recv, size_t bytes, void *stream) {
  return fabric_exchange((FabricLink *)user, send, recv, bytes, nullptr, nullptr, (hipStream_t)stream);
}

static int fabric_alltoallv(void *user, const void *send, void *recv, size_t block_bytes, size_t send_off, size_t send_bytes,
                            const size_t *recv_off, const size_t *recv_bytes, void *stream) {
  (void)send_off; (void)send_bytes;  // every rank derives the same ranges; the puller uses its own copy of them
  return fabric_exchange((FabricLink *)user, send, recv, block_bytes, recv_off, recv_bytes, (hipStream_t)stream);
}

static int fabric_allreduce(void *user, void *buf, size_t count, int is_u64, void *stream) {
  FabricLink *l = (FabricLink *)user;
  pf_fabric *f = l->f;
  hipStream_t st = (hipStream_t)stream;
  std::vector<unsigned long long> mine(count);  // (a handful of scalars a few times per run: through the host, synchronously)
  if (hipMemcpyAsync(mine.data(), buf, count * 8, hipMemcpyDeviceToHost, st) != hipSuccess) f->error = 1;
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;
  f->buf[l->rank] = mine.data();
  fabric_barrier(f);
  std::vector<unsigned long long> out(count);
  for (size_t i = 0; i < count; i++) {
    if (is_u64) {
      unsigned long long s = 0;
      for (int p = 0; p < f->P; p++) s += ((unsigned long long *)f->buf[p])[i];
      out[i] = s;
    } else {
      double s = 0;  // rank order: the same sum on every rank
      for (int p = 0; p < f->P; p++) s += ((double *)f->buf[p])[i];
      memcpy(&out[i], &s, 8);
    }
  }
  fabric_barrier(f);  // everybody has read everybody's host copy
  if (hipMemcpyAsync(buf, out.data(), count * 8, hipMemcpyHostToDevice, st) != hipSuccess) f->error = 1;
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;
  return f->error;
}

extern "C" int pf_ctx_rank_size(pf_ctx *ctx, int *rank, int *nranks);

extern "C" pf_fabric *pf_fabric_create(int nranks) {
  if (nranks < 1) return nullptr;
  pf_fabric *f = new pf_fabric();
  f->P = nranks;
  f->send.assign(nranks, nullptr);
  f->buf.assign(nranks, nullptr);
  f->ready.assign(nranks, nullptr);
  f->pulled.assign(nranks, nullptr);
  for (int p = 0; p < nranks; p++)
    if (hipEventCreateWithFlags(&f->ready[p], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&f->pulled[p], hipEventDisableTiming) != hipSuccess) {
      delete f;
      return nullptr;
    }
  return f;
}
extern "C" void pf_fabric_destroy(pf_fabric *f) {
  if (!f) return;
  for (auto e : f->ready) if (e) hipEventDestroy(e);
  for (auto e : f->pulled) if (e) hipEventDestroy(e);
  delete f;
}
extern "C" int pf_fabric_set_delay(pf_fabric *f, int microseconds) {
  if (!f || microseconds < 0) return 1;
  f->delay_us = microseconds;
  return 0;
}
extern "C" int pf_fabric_attach(pf_fabric *f, pf_ctx *ctx) {
  if (!f || !ctx) return 1;
  FabricLink *l = new FabricLink();
  int P = 0;
  if (pf_ctx_rank_size(ctx, &l->rank, &P) || P != f->P) { delete l; return 1; }
  l->f = f;
  pf_set_exchange(ctx, fabric_alltoall, l);
  pf_set_exchange_rows(ctx, fabric_alltoallv, l);
  pf_set_allreduce(ctx, fabric_allreduce, l);
  return 0;
}
