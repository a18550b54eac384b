// group.hpp -- shard groups of the C ABI (ilqr_group_*): several handles, one batch; the one exchange of the path -- the gather
// of per-trajectory costs -- as an RCCL all-gather between distinct devices (librccl.so loaded on first use) or plain copies
// between shards that share a device.  Included once, by capi.hip.
#pragma once
#include "handle.hpp"

// ---- shard groups (include/ilqr_amd.h) -------------------------------------------------------
#include <dlfcn.h>
// RCCL is loaded at run time (dlopen below) and only when a group spans devices, so its header must not be a build dependency:
// the real declarations where the header exists, otherwise the six entry points and three types this file uses (nccl.h's ABI)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;  // ncclFloat64
#endif
namespace {
struct RcclApi {  // librccl.so, loaded on first use: a single-GPU user of the library never maps it
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString;
  }
};
RcclApi g_rccl;
}  // namespace
struct ilqr_group {
  std::vector<ilqr_batch*> shards;
  bool rccl = false;
  int per = 0;                       // padded shard length of the all-gather (the largest B)
  std::vector<ncclComm_t> comms;     // one per shard (= per device), in shard order
  std::vector<double*> send, recv;   // per device: [per], [n_shards * per]
};
#define NCCLCHK(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) return fail(ILQR_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString(r_)); \
  } while (0)
extern "C" {
int ilqr_group_create(ilqr_batch* const* shards, int n_shards, int flags, ilqr_group** out) {
  if (!shards || !out || n_shards < 1) return fail(ILQR_ERR_INVALID, "ilqr_group_create: null argument / no shards");
  for (int i = 0; i < n_shards; i++)
    if (!shards[i]) return fail(ILQR_ERR_INVALID, "ilqr_group_create: shard %d is null", i);
  ilqr_group* g = new ilqr_group();
  g->shards.assign(shards, shards + n_shards);
  bool distinct = true;
  for (int i = 0; i < n_shards; i++)
    for (int j = 0; j < i; j++) distinct = distinct && shards[i]->device != shards[j]->device;
  for (int i = 0; i < n_shards; i++) g->per = std::max(g->per, shards[i]->B);
  g->rccl = distinct && (n_shards > 1 || (flags & 1));
  if (g->rccl) {
    if (!g_rccl.load()) {
      delete g;
      return fail(ILQR_ERR_UNSUPPORTED, "shards on %d devices need librccl.so for their gather: %s", n_shards, dlerror());
    }
    std::vector<int> devs(n_shards);
    for (int i = 0; i < n_shards; i++) devs[i] = shards[i]->device;
    g->comms.resize(n_shards);
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_shards, devs.data());
    if (r != ncclSuccess) {
      g->comms.clear();
      delete g;
      return fail(ILQR_ERR_HIP, "ncclCommInitAll over %d devices: %s", n_shards, g_rccl.GetErrorString(r));
    }
    g->send.assign(n_shards, nullptr);
    g->recv.assign(n_shards, nullptr);
    for (int i = 0; i < n_shards; i++) {
      if (hipSetDevice(devs[i]) != hipSuccess || hipMalloc((void**)&g->send[i], (size_t)g->per * sizeof(double)) != hipSuccess ||
          hipMalloc((void**)&g->recv[i], (size_t)n_shards * g->per * sizeof(double)) != hipSuccess ||
          hipMemset(g->send[i], 0, (size_t)g->per * sizeof(double)) != hipSuccess) {
        ilqr_group_destroy(g);
        return fail(ILQR_ERR_HIP, "ilqr_group_create: device buffers of shard %d", i);
      }
    }
  }
  *out = g;
  return 0;
}
void ilqr_group_destroy(ilqr_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); i++) {
    (void)hipSetDevice(g->shards[i]->device);
    if (i < g->send.size() && g->send[i]) (void)hipFree(g->send[i]);
    if (i < g->recv.size() && g->recv[i]) (void)hipFree(g->recv[i]);
    if (g->comms[i]) (void)g_rccl.CommDestroy(g->comms[i]);
  }
  delete g;
}
int ilqr_group_uses_rccl(ilqr_group* g, int* n_ranks) {
  if (!g) return 0;
  if (n_ranks) *n_ranks = (int)g->comms.size();
  return g->rccl ? 1 : 0;
}
int ilqr_group_gather_costs(ilqr_group* g, double* cost_out) {
  if (!g || !cost_out) return fail(ILQR_ERR_INVALID, "null argument");
  const int n = (int)g->shards.size();
  if (!g->rccl) {  // shards share a device: plain copies, shard by shard
    size_t off = 0;
    for (int i = 0; i < n; i++) {
      if (int rc = ilqr_get_cost(g->shards[i], cost_out + off)) return rc;
      off += (size_t)g->shards[i]->B;
    }
    return 0;
  }
  // every shard's costs into its device's send buffer (on the shard's stream), then ONE all-gather over the devices' links
  for (int i = 0; i < n; i++)
    if (int rc = ilqr_copy_cost_to_device(g->shards[i], g->send[i])) return rc;
  NCCLCHK(g_rccl.GroupStart());
  for (int i = 0; i < n; i++) {  // (a failure inside the group closes it before returning: the calling thread must not be left inside an open NCCL group)
    const hipError_t he = hipSetDevice(g->shards[i]->device);
    const ncclResult_t nr = (he == hipSuccess) ? g_rccl.AllGather(g->send[i], g->recv[i], (size_t)g->per, ncclDouble, g->comms[i], g->shards[i]->stream)  // (stream order: after the copy)
                                               : ncclSuccess;
    if (he != hipSuccess || nr != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return he != hipSuccess ? fail(ILQR_ERR_HIP, "hipSetDevice(%d) inside the gather: %s", g->shards[i]->device, hipGetErrorString(he))
                              : fail(ILQR_ERR_HIP, "ncclAllGather of shard %d: %s", i, g_rccl.GetErrorString(nr));
    }
  }
  NCCLCHK(g_rccl.GroupEnd());
  std::vector<double> all((size_t)n * g->per);
  ilqr_batch* h0 = g->shards[0];
  HIPCHK(hipSetDevice(h0->device));
  HIPCHK(hipMemcpyAsync(all.data(), g->recv[0], all.size() * sizeof(double), hipMemcpyDeviceToHost, h0->stream));
  for (int i = 0; i < n; i++) {
    HIPCHK(hipSetDevice(g->shards[i]->device));
    HIPCHK(hipStreamSynchronize(g->shards[i]->stream));
  }
  size_t off = 0;
  for (int i = 0; i < n; i++) {  // drop the padding of ragged shards
    std::copy(all.begin() + (size_t)i * g->per, all.begin() + (size_t)i * g->per + g->shards[i]->B, cost_out + off);
    off += (size_t)g->shards[i]->B;
  }
  return 0;
}

}  // extern "C"

