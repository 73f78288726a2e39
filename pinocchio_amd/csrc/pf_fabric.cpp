// pf_fabric.cpp -- in-process exchange between P contexts that share ONE GPU.
//
// A bring-up / test transport (what MPI calls a "self" BTL): P host threads,
// each driving its own pf_ctx (rank r of P) on the same device, meet in the
// all-to-all and the all-reduce through a mutex/condvar barrier and move the
// blocks with device-to-device copies.  It exercises exactly the slab code
// path of an 8-GPU run (KY/XS layouts, block addressing, wavenumber offsets,
// pack-free send blocks) on a 1-GPU box; production runs use pf_init_rccl.
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
  std::vector<void *> buf;
  std::vector<double> acc_d;
  std::vector<unsigned long long> acc_u;
  int error = 0;
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

static int fabric_alltoall(void *user, const void *send, void *recv, size_t bytes, void *stream) {
  FabricLink *l = (FabricLink *)user;
  pf_fabric *f = l->f;
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;  // my blocks are complete
  f->send[l->rank] = send;
  fabric_barrier(f);
  for (int p = 0; p < f->P; p++)  // pull block `rank` of every peer
    if (hipMemcpyAsync((char *)recv + (size_t)p * bytes, (const char *)f->send[p] + (size_t)l->rank * bytes, bytes,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
      f->error = 1;
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;
  fabric_barrier(f);  // peers may reuse their send buffers
  return f->error;
}

static int fabric_alltoallv(void *user, const void *send, void *recv, size_t block_bytes, size_t send_off, size_t send_bytes,
                            const size_t *recv_off, const size_t *recv_bytes, void *stream) {
  FabricLink *l = (FabricLink *)user;
  pf_fabric *f = l->f;
  hipStream_t st = (hipStream_t)stream;
  (void)send_off; (void)send_bytes;  // every rank derives the same ranges; the puller uses its own copy of them
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;
  f->send[l->rank] = send;
  fabric_barrier(f);
  for (int p = 0; p < f->P; p++)  // pull the row range of block `rank` of every peer that has one
    if (recv_bytes[p] &&
        hipMemcpyAsync((char *)recv + (size_t)p * block_bytes + recv_off[p], (const char *)f->send[p] + (size_t)l->rank * block_bytes + recv_off[p],
                       recv_bytes[p], hipMemcpyDeviceToDevice, st) != hipSuccess)
      f->error = 1;
  if (hipStreamSynchronize(st) != hipSuccess) f->error = 1;
  fabric_barrier(f);
  return f->error;
}

static int fabric_allreduce(void *user, void *buf, size_t count, int is_u64, void *stream) {
  FabricLink *l = (FabricLink *)user;
  pf_fabric *f = l->f;
  hipStream_t st = (hipStream_t)stream;
  std::vector<unsigned long long> mine(count);
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
  return f;
}
extern "C" void pf_fabric_destroy(pf_fabric *f) { delete f; }
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
