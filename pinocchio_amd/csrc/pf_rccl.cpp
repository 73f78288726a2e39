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
#include <stdlib.h>
#include <string.h>

#include "../../include/pinfmax.h"

// Types, enumerators and prototypes come from the RCCL header this library is BUILT against (/opt/rocm/include/rccl/rccl.h:
// ncclUniqueId, ncclChar / ncclUint64 / ncclDouble, ncclSum ...), never from values copied by hand; the library bound at run time
// must then be of the same major version (ncclGetVersion against NCCL_MAJOR, checked in load_rccl): a drift of the ABI is refused
// with a message instead of passing a wrong enumerator to a collective.
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == PF_RCCL_ID_BYTES, "pf_rccl_unique_id / pf_init_rccl pass the id as PF_RCCL_ID_BYTES bytes");

struct RcclApi {
  void *h;
  int version;
  decltype(&ncclGetVersion) GetVersion;
  decltype(&ncclGetErrorString) GetErrorString;
  decltype(&ncclGetUniqueId) GetUniqueId;
  decltype(&ncclCommInitRank) CommInitRank;
  decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclCommCount) CommCount;
  decltype(&ncclSend) Send;
  decltype(&ncclRecv) Recv;
  decltype(&ncclAllReduce) AllReduce;
  decltype(&ncclGroupStart) GroupStart;
  decltype(&ncclGroupEnd) GroupEnd;
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
  SYM(GetVersion, "ncclGetVersion") SYM(GetErrorString, "ncclGetErrorString")
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount")
  SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
  // the library at run time against the header at build time: same major version, or nothing is bound
  int v = 0;
  if (g_api.GetVersion(&v) != ncclSuccess) { printf("ERROR on task 0: ncclGetVersion failed\n"); return 1; }
  const int major = v >= 10000 ? v / 10000 : v / 1000;  // (NCCL_VERSION_CODE: X * 10000 + Y * 100 + Z since 2.9)
  if (major != NCCL_MAJOR) {
    printf("ERROR on task 0: librccl at run time is version %d, libpinfmax_hip was built against %d.%d.x: refusing to bind it\n", v, NCCL_MAJOR, NCCL_MINOR);
    memset(&g_api, 0, sizeof(g_api));
    return 1;
  }
  g_api.version = v;
  g_api.h = h;
  return 0;
}
// run-time / build-time versions of RCCL (0 / the header's code when nothing is bound)
extern "C" int pf_rccl_version(int *build_code) {
  if (build_code) *build_code = NCCL_VERSION_CODE;
  return g_api.h ? g_api.version : 0;
}

struct RcclLink {
  ncclComm_t comm; int rank, nranks;
  bool self_send;  // PF_RCCL_SELF_SEND=1 (read once, in pf_init_rccl): the rank's own block goes through ncclSend / ncclRecv to itself
                   // inside the group, as in rounds 1-5; default: a device-to-device copy on the same stream, outside RCCL (no self
                   // connection, no proxy work for 1/P of every field) -- the first run on eight GPUs can A/B the two
};
static int self_copy(const void *src, void *dst, size_t bytes, void *stream) {
  return bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess;
}

// (a group that was opened is always closed, also when a call inside it fails: an open group would swallow every later
//  RCCL call of the process, the fallback exchange's included)
static int rccl_alltoall(void *user, const void *send, void *recv, size_t bytes, void *stream) {
  RcclLink *l = (RcclLink *)user;
  if (g_api.GroupStart()) return 1;
  int bad = 0;
  for (int q = 0; q < l->nranks && !bad; q++) {
    if (q == l->rank && !l->self_send) continue;
    bad |= g_api.Send((const char *)send + (size_t)q * bytes, bytes, ncclChar, q, l->comm, (hipStream_t)stream) != ncclSuccess;
    if (!bad) bad |= g_api.Recv((char *)recv + (size_t)q * bytes, bytes, ncclChar, q, l->comm, (hipStream_t)stream) != ncclSuccess;
  }
  bad |= g_api.GroupEnd() != 0;
  if (!bad && !l->self_send) bad |= self_copy((const char *)send + (size_t)l->rank * bytes, (char *)recv + (size_t)l->rank * bytes, bytes, stream);
  return bad;
}
// the same exchange restricted to one row range per block (band-limited spectra): variable message sizes, empty ones skipped
static int rccl_alltoallv(void *user, const void *send, void *recv, size_t block_bytes, size_t send_off, size_t send_bytes,
                          const size_t *recv_off, const size_t *recv_bytes, void *stream) {
  RcclLink *l = (RcclLink *)user;
  if (g_api.GroupStart()) return 1;
  int bad = 0;
  for (int q = 0; q < l->nranks && !bad; q++) {
    if (q == l->rank && !l->self_send) continue;
    if (send_bytes) bad |= g_api.Send((const char *)send + (size_t)q * block_bytes + send_off, send_bytes, ncclChar, q, l->comm, (hipStream_t)stream) != ncclSuccess;
    if (!bad && recv_bytes[q]) bad |= g_api.Recv((char *)recv + (size_t)q * block_bytes + recv_off[q], recv_bytes[q], ncclChar, q, l->comm, (hipStream_t)stream) != ncclSuccess;
  }
  bad |= g_api.GroupEnd() != 0;
  // (the rank's own row range: what it sends is what it receives from itself -- recv_off[rank] == send_off, recv_bytes[rank] == send_bytes)
  if (!bad && !l->self_send)
    bad |= self_copy((const char *)send + (size_t)l->rank * block_bytes + send_off, (char *)recv + (size_t)l->rank * block_bytes + recv_off[l->rank], recv_bytes[l->rank], stream);
  return bad;
}
static int rccl_allreduce(void *user, void *buf, size_t count, int is_u64, void *stream) {
  RcclLink *l = (RcclLink *)user;
  return g_api.AllReduce(buf, buf, count, is_u64 ? ncclUint64 : ncclDouble, ncclSum, l->comm, (hipStream_t)stream) != ncclSuccess ? 1 : 0;
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
  { const char *e = getenv("PF_RCCL_SELF_SEND"); l->self_send = e && atoi(e) != 0; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = g_api.CommInitRank(&l->comm, l->nranks, id, l->rank);
  if (rc != ncclSuccess) {
    printf("ERROR on task %d: ncclCommInitRank failed: %s (NCCL_DEBUG=WARN shows RCCL's own account)\n", l->rank, g_api.GetErrorString(rc));
    delete l;
    return 1;
  }
  pf_set_exchange(ctx, rccl_alltoall, l);
  pf_set_exchange_rows(ctx, rccl_alltoallv, l);
  pf_set_allreduce(ctx, rccl_allreduce, l);
  pf_ctx_set_rccl(ctx, l);
  return 0;
}
