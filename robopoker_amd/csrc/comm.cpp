// comm.cpp — rp_comm: the RCCL communicator a non-Python host hands to the sharded entry points (SURVEY §8b proposed
// rp_mccfr_allreduce(h, rp_comm*); §8e: one all-gather of composed maps per exchange window for MCCFR, one integer
// all-reduce per Elkan iteration for k-means).  librccl is loaded on first use (dlopen), so single-GPU hosts never map it.
// One process per GPU; the collectives are enqueued on the solver's / layer's own HIP stream — no host synchronisation.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "rp_internal.h"

namespace {
// the slice of rccl.h this file needs (RCCL 2.x ABI: /opt/rocm/include/rccl/rccl.h:40-52,448-466)
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
enum { NCCL_SUCCESS = 0, NCCL_SUM = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_UINT32 = 3, NCCL_INT64 = 4, NCCL_UINT64 = 5 };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_p) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_p, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
int g_load_rc = RP_OK;

int load_rccl() {
    std::call_once(g_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) {
            g_load_rc = rp::fail(RP_ERR_UNSUPPORTED, "rp_comm: librccl.so not found (%s)", dlerror());
            return;
        }
#define RP_SYM(field, sym)                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, sym));       \
    if (!g_rccl.field) g_load_rc = rp::fail(RP_ERR_UNSUPPORTED, "rp_comm: librccl lacks %s", sym)
        RP_SYM(GetUniqueId, "ncclGetUniqueId");
        RP_SYM(CommInitRank, "ncclCommInitRank");
        RP_SYM(CommDestroy, "ncclCommDestroy");
        RP_SYM(AllGather, "ncclAllGather");
        RP_SYM(AllReduce, "ncclAllReduce");
        RP_SYM(GetErrorString, "ncclGetErrorString");
#undef RP_SYM
    });
    return g_load_rc;
}
int nccl_fail(const char* what, int code) {
    return rp::fail(RP_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "rccl error");
}
}  // namespace

struct rp_comm {
    ncclComm_p comm = nullptr;
    int rank = 0, world = 1, device = 0;
    bool owned = true;
};

namespace rp {
int comm_world(const rp_comm* c) { return c->world; }
int comm_rank(const rp_comm* c) { return c->rank; }
int comm_all_gather(rp_comm* c, const void* send, void* recv, size_t bytes, hipStream_t stream) {
    const int rc = g_rccl.AllGather(send, recv, bytes, NCCL_UINT8, c->comm, stream);
    return rc == NCCL_SUCCESS ? RP_OK : nccl_fail("ncclAllGather", rc);
}
// in place; kind: 0 = i32, 1 = i64
int comm_all_reduce_sum(rp_comm* c, void* buf, size_t count, int kind, hipStream_t stream) {
    const int rc = g_rccl.AllReduce(buf, buf, count, kind ? NCCL_INT64 : NCCL_INT32, NCCL_SUM, c->comm, stream);
    return rc == NCCL_SUCCESS ? RP_OK : nccl_fail("ncclAllReduce", rc);
}
}  // namespace rp

extern "C" {

int rp_comm_unique_id(uint8_t* id) {
    if (!id) return rp::fail(RP_ERR_INVALID, "rp_comm_unique_id: NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId_t u;
    const int n = g_rccl.GetUniqueId(&u);
    if (n != NCCL_SUCCESS) return nccl_fail("ncclGetUniqueId", n);
    memcpy(id, u.internal, RP_COMM_ID_BYTES);
    return RP_OK;
}

int rp_comm_create(const uint8_t* id, int rank, int world, int device, rp_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return rp::fail(RP_ERR_INVALID, "rp_comm_create: bad argument");
    if (rp_device_count() <= 0) return rp::fail(RP_ERR_NO_DEVICE, "rp_comm_create: no HIP device visible");
    int rc = load_rccl();
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return rp::fail(RP_ERR_HIP, "rp_comm_create: hipSetDevice(%d) failed", device);
    ncclUniqueId_t u;
    memcpy(u.internal, id, RP_COMM_ID_BYTES);
    rp_comm* c = new rp_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    const int n = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (n != NCCL_SUCCESS) {
        delete c;
        return nccl_fail("ncclCommInitRank", n);
    }
    *out = c;
    return RP_OK;
}

int rp_comm_adopt(void* nccl_comm, int rank, int world, int device, rp_comm** out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return rp::fail(RP_ERR_INVALID, "rp_comm_adopt: bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    rp_comm* c = new rp_comm();
    c->comm = nccl_comm;
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->owned = false;
    *out = c;
    return RP_OK;
}

int rp_comm_destroy(rp_comm* c) {
    if (!c) return RP_OK;
    if (c->owned && c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return RP_OK;
}

}  // extern "C"
