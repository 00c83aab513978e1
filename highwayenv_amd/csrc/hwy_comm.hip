// hwy_comm.hip -- the multi-GPU exchange step behind the C-ABI (include/hwy_engine.h: hwy_comm_*, hwy_gather).
//
// Environments are independent, so the engines of a node never talk while stepping; the only exchange is the gather of every
// rank's (obs | reward | done) block to a root rank, one collective per batched step (or per K steps), over RCCL / xGMI
// (ncclGather, /opt/rocm/include/rccl/rccl.h:745).  One process per GPU: rank 0 creates a unique id,
// the caller ships its 128 bytes to the other ranks by any means (MPI, a TCP store, torch.distributed, a file), every rank
// calls hwy_comm_init.  librccl is loaded on first use (dlopen), so a single-GPU user never needs it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/hwy_engine.h"
#include "hwy_comm.h"

namespace {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[HWY_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { ncclUint8 = 1 };    // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (handle) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) { err = std::string("dlopen librccl: ") + dlerror(); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    Gather = (decltype(Gather))dlsym(handle, "ncclGather");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !Gather) { err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclGather"; return false; }
    return true;
  }
  std::string what(ncclResult_t r) { return GetErrorString ? GetErrorString(r) : std::to_string(r); }
};
Rccl g_rccl;
}  // namespace

namespace hwy {
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};
int comm_unique_id(uint8_t *id, std::string &err) {
  if (!g_rccl.load()) { err = g_rccl.err; return HWY_ERR_UNSUPPORTED; }
  ncclUniqueId u;
  if (ncclResult_t r = g_rccl.GetUniqueId(&u)) { err = "ncclGetUniqueId: " + g_rccl.what(r); return HWY_ERR_HIP; }
  std::memcpy(id, u.internal, HWY_COMM_ID_BYTES);
  return HWY_OK;
}
int comm_init(Comm **out, const uint8_t *id, int rank, int world, std::string &err) {
  if (!g_rccl.load()) { err = g_rccl.err; return HWY_ERR_UNSUPPORTED; }
  ncclUniqueId u;
  std::memcpy(u.internal, id, HWY_COMM_ID_BYTES);
  Comm *c = new Comm();
  c->rank = rank; c->world = world;
  if (ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank)) {
    err = "ncclCommInitRank: " + g_rccl.what(r);
    delete c;
    return HWY_ERR_HIP;
  }
  *out = c;
  return HWY_OK;
}
int comm_gather(Comm *c, const void *d_send, void *d_recv, size_t bytes, int root, hipStream_t stream, std::string &err) {
  if (ncclResult_t r = g_rccl.Gather(d_send, d_recv, bytes, ncclUint8, root, c->comm, stream)) {
    err = "ncclGather: " + g_rccl.what(r);
    return HWY_ERR_HIP;
  }
  return HWY_OK;
}
void comm_destroy(Comm *c) {
  if (!c) return;
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}
int comm_rank(const Comm *c) { return c->rank; }
int comm_world(const Comm *c) { return c->world; }
}  // namespace hwy
