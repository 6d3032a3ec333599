// rccl_leg.hip - RCCL binding of libsoslam_ba (host code only; see rccl_leg.h).
#include "rccl_leg.h"

#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace soslam {

namespace {

// the few RCCL declarations this leg needs (rccl.h: ncclUniqueId, ncclComm_t, ncclAllReduce ...), restated so that the
// library builds without the RCCL headers and binds at run time
struct NcclUniqueId { char internal[kRcclIdBytes]; };
typedef void* NcclCommT;
enum { kNcclSuccess = 0 };
enum { kNcclSum = 0, kNcclMax = 2 };
enum { kNcclFloat64 = 8 };

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclCommT*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclCommT) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclCommT, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

const RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
    });
    if (!ok) {
        set_last_error("RCCL not available: %s", api.lib ? "librccl lacks a required symbol" : dlerror());
        return nullptr;
    }
    return &api;
}

int rccl_fail(const RcclApi* api, const char* what, int rc)
{
    set_last_error("%s failed: %s", what, api->GetErrorString(rc));
    return SOSLAM_ERR_COMM;
}

}  // namespace

struct RcclComm {
    NcclCommT comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

int rccl_get_unique_id(void* id128)
{
    const RcclApi* api = rccl_api();
    if (!api) return SOSLAM_ERR_COMM;
    NcclUniqueId id;
    const int rc = api->GetUniqueId(&id);
    if (rc != kNcclSuccess) return rccl_fail(api, "ncclGetUniqueId", rc);
    std::memcpy(id128, id.internal, kRcclIdBytes);
    return SOSLAM_OK;
}

int rccl_comm_create(const void* id128, int rank, int world, int device, RcclComm** out)
{
    *out = nullptr;
    const RcclApi* api = rccl_api();
    if (!api) return SOSLAM_ERR_COMM;
    SOSLAM_HIP_CHECK(hipSetDevice(device));
    NcclUniqueId id;
    std::memcpy(id.internal, id128, kRcclIdBytes);
    RcclComm* c = new RcclComm();
    c->rank = rank; c->world = world; c->device = device;
    const int rc = api->CommInitRank(&c->comm, world, id, rank);   // collective over all ranks of the job
    if (rc != kNcclSuccess) {
        delete c;
        return rccl_fail(api, "ncclCommInitRank", rc);
    }
    *out = c;
    return SOSLAM_OK;
}

void rccl_comm_destroy(RcclComm* c)
{
    if (!c) return;
    const RcclApi* api = rccl_api();
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    delete c;
}

int rccl_allreduce_f64(RcclComm* c, double* buf, size_t count, int op, hipStream_t stream)
{
    const RcclApi* api = rccl_api();
    if (!api || !c) return SOSLAM_ERR_COMM;
    const int rc = api->AllReduce(buf, buf, count, kNcclFloat64, op == SOSLAM_REDUCE_MAX ? kNcclMax : kNcclSum, c->comm, stream);
    if (rc != kNcclSuccess) return rccl_fail(api, "ncclAllReduce", rc);
    return SOSLAM_OK;
}

}  // namespace soslam
