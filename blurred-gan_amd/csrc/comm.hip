// Data-parallel exchange step of the C ABI: SUM all-reduce of the flat gradient buffers over RCCL.
//
// RCCL is bound lazily (dlopen at bg_comm_init) so libbgan_hip.so has no link-time dependency on it: a single-GPU host never
// loads the collective library, and a process that already holds torch's copy of librccl gets that one back from dlopen.
// One communicator per process (one process per GPU); the handle is the only state the library keeps between calls.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int bind_rccl() {
  if (g_rccl.handle) return BG_OK;
  const char* names[] = {getenv("BGAN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return bg::fail(BG_ERR_RCCL, "bg_comm: cannot load librccl (%s)", dlerror());
  Rccl r;
  r.handle = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));            // optional (bg_comm_query)
  r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
    dlclose(h);
    return bg::fail(BG_ERR_RCCL, "bg_comm: librccl lacks a required symbol");
  }
  g_rccl = r;
  return BG_OK;
}

int rccl_fail(const char* what, ncclResult_t e) { return bg::fail(BG_ERR_RCCL, "%s: %s", what, g_rccl.GetErrorString(e)); }

}  // namespace

struct bg_comm {
  ncclComm_t comm;
  int rank, nranks;
};

static_assert(BG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" {

int bg_comm_unique_id(unsigned char* id_out) {
  BG_REQUIRE(id_out, BG_ERR_NULL, "bg_comm_unique_id: null pointer");
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, id.internal, BG_COMM_ID_BYTES);
  return BG_OK;
}

int bg_comm_init(bg_comm** out, int rank, int nranks, const unsigned char* id_bytes) {
  BG_REQUIRE(out && id_bytes, BG_ERR_NULL, "bg_comm_init: null pointer");
  BG_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, BG_ERR_BAD_SHAPE, "bg_comm_init: rank=%d nranks=%d", rank, nranks);
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, BG_COMM_ID_BYTES);
  ncclComm_t c;
  ncclResult_t e = g_rccl.CommInitRank(&c, nranks, id, rank);   // binds the communicator to the current HIP device
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  *out = new bg_comm{c, rank, nranks};
  return BG_OK;
}

int bg_allreduce_sum_f32(bg_comm* comm, float* buf_d, size_t n, void* stream) {
  BG_REQUIRE(comm && buf_d, BG_ERR_NULL, "bg_allreduce_sum_f32: null pointer");
  BG_REQUIRE(n > 0, BG_ERR_BAD_SHAPE, "bg_allreduce_sum_f32: empty buffer");
  ncclResult_t e = g_rccl.AllReduce(buf_d, buf_d, n, ncclFloat32, ncclSum, comm->comm, static_cast<hipStream_t>(stream));
  if (e != ncclSuccess) return rccl_fail("ncclAllReduce", e);
  return BG_OK;
}

int bg_comm_query(bg_comm* comm, int* nranks, int* rank) {
  BG_REQUIRE(comm, BG_ERR_NULL, "bg_comm_query: null communicator");
  int n = comm->nranks, r = comm->rank;
  if (g_rccl.CommCount && g_rccl.CommUserRank) {            // what RCCL itself says, not what the caller passed to bg_comm_init
    ncclResult_t e = g_rccl.CommCount(comm->comm, &n);
    if (e != ncclSuccess) return rccl_fail("ncclCommCount", e);
    e = g_rccl.CommUserRank(comm->comm, &r);
    if (e != ncclSuccess) return rccl_fail("ncclCommUserRank", e);
  }
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  return BG_OK;
}

int bg_comm_destroy(bg_comm* comm) {
  if (!comm) return BG_OK;
  ncclResult_t e = g_rccl.CommDestroy(comm->comm);
  delete comm;
  if (e != ncclSuccess) return rccl_fail("ncclCommDestroy", e);
  return BG_OK;
}

}  // extern "C"
