// pf_rccl.cpp -- built-in multi-GPU exchange over RCCL/xGMI (pf_init_rccl).
//
// One all-to-all per 3-D FFT, expressed as grouped ncclSend/ncclRecv: on the
// fully connected xGMI mesh every peer pair has its own link, so the P-1
// point-to-point transfers of a rank run concurrently.
// Replaces the MPI_Alltoall inside pfft_execute (src/fmax-pfft.c:197,211) and
// the small MPI_Reduce/MPI_Bcast calls (src/collapse_times.c:656-667, src/fmax.c:527).
//
// RCCL is bound at run time (dlopen) so that libpinfmax_hip.so loads on a
// single GPU without it and never clashes with an RCCL already in the process.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pinfmax.h"

typedef void *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclChar_ = 0, ncclUint64_ = 5, ncclDouble_ = 8, ncclSum_ = 0 };

struct RcclApi {
  void *h;
  int (*GetUniqueId)(ncclUniqueId *);
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  int (*CommDestroy)(ncclComm_t);
  int (*CommCount)(const ncclComm_t, int *);
  int (*Send)(const void *, size_t, int, int, ncclComm_t, void *);
  int (*Recv)(void *, size_t, int, int, ncclComm_t, void *);
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, void *);
  int (*GroupStart)();
  int (*GroupEnd)();
};
static RcclApi g_api = {};

static int load_rccl() {
  if (g_api.h) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char *nm : names) {
    if (h) break;
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) { printf("ERROR on task 0: cannot load librccl (%s)\n", dlerror()); return 1; }
#define SYM(field, name) *(void **)(&g_api.field) = dlsym(h, name); if (!g_api.field) { printf("ERROR on task 0: missing %s in librccl\n", name); return 1; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount")
  SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
  g_api.h = h;
  return 0;
}

struct RcclLink { ncclComm_t comm; int rank, nranks; };

// (a group that was opened is always closed, also when a call inside it fails: an open group would swallow every later
//  RCCL call of the process, the fallback exchange's included)
static int rccl_alltoall(void *user, const void *send, void *recv, size_t bytes, void *stream) {
  RcclLink *l = (RcclLink *)user;
  if (g_api.GroupStart()) return 1;
  int bad = 0;
  for (int q = 0; q < l->nranks && !bad; q++) {
    bad |= g_api.Send((const char *)send + (size_t)q * bytes, bytes, ncclChar_, q, l->comm, stream) != 0;
    if (!bad) bad |= g_api.Recv((char *)recv + (size_t)q * bytes, bytes, ncclChar_, q, l->comm, stream) != 0;
  }
  return (g_api.GroupEnd() != 0) | bad;
}
// the same exchange restricted to one row range per block (band-limited spectra): variable message sizes, empty ones skipped
static int rccl_alltoallv(void *user, const void *send, void *recv, size_t block_bytes, size_t send_off, size_t send_bytes,
                          const size_t *recv_off, const size_t *recv_bytes, void *stream) {
  RcclLink *l = (RcclLink *)user;
  if (g_api.GroupStart()) return 1;
  int bad = 0;
  for (int q = 0; q < l->nranks && !bad; q++) {
    if (send_bytes) bad |= g_api.Send((const char *)send + (size_t)q * block_bytes + send_off, send_bytes, ncclChar_, q, l->comm, stream) != 0;
    if (!bad && recv_bytes[q]) bad |= g_api.Recv((char *)recv + (size_t)q * block_bytes + recv_off[q], recv_bytes[q], ncclChar_, q, l->comm, stream) != 0;
  }
  return (g_api.GroupEnd() != 0) | bad;
}
static int rccl_allreduce(void *user, void *buf, size_t count, int is_u64, void *stream) {
  RcclLink *l = (RcclLink *)user;
  return g_api.AllReduce(buf, buf, count, is_u64 ? ncclUint64_ : ncclDouble_, ncclSum_, l->comm, stream) ? 1 : 0;
}

// host-side only: can this process bind RCCL at all?  No collective, no communicator -- safe to call before the ranks
// vote on the exchange kind (a rank that cannot must say so BEFORE its peers block inside ncclCommInitRank)
extern "C" int pf_rccl_available(void) { return load_rccl() ? 0 : 1; }

extern "C" int pf_rccl_unique_id(void *id128) {
  if (!id128 || load_rccl()) return 1;
  ncclUniqueId id;
  if (g_api.GetUniqueId(&id)) return 1;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// rank/nranks come from the context's own configuration; the context owns the link (pf_destroy / pf_release_rccl)
extern "C" int pf_ctx_rank_size(pf_ctx *ctx, int *rank, int *nranks);
extern "C" int pf_ctx_set_rccl(pf_ctx *ctx, void *link);

// ranks RCCL itself counts in the communicator of this link (ncclCommCount): what a bench line reports beside WORLD_SIZE
extern "C" int pf_rccl_link_count(void *link) {
  RcclLink *l = (RcclLink *)link;
  int n = 0;
  if (!l || !g_api.h || !l->comm || g_api.CommCount(l->comm, &n)) return -1;
  return n;
}

extern "C" void pf_rccl_release(void *link) {
  RcclLink *l = (RcclLink *)link;
  if (!l) return;
  if (g_api.h && l->comm) g_api.CommDestroy(l->comm);
  delete l;
}

extern "C" int pf_init_rccl(pf_ctx *ctx, const void *id128) {
  if (!ctx || !id128 || load_rccl()) return 1;
  RcclLink *l = new RcclLink();
  if (pf_ctx_rank_size(ctx, &l->rank, &l->nranks)) { delete l; return 1; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  if (g_api.CommInitRank(&l->comm, l->nranks, id, l->rank)) {
    printf("ERROR on task %d: ncclCommInitRank failed\n", l->rank);
    delete l;
    return 1;
  }
  pf_set_exchange(ctx, rccl_alltoall, l);
  pf_set_exchange_rows(ctx, rccl_alltoallv, l);
  pf_set_allreduce(ctx, rccl_allreduce, l);
  pf_ctx_set_rccl(ctx, l);
  return 0;
}
