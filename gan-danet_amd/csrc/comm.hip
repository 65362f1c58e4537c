// RCCL communicator behind the C ABI (gd_comm_init / gd_allreduce / gd_comm_destroy, plus the
// reduce-scatter / all-gather pair of the sharded fc1 update).  For hosts that do NOT bring torch.distributed: the
// Python package here uses torch.distributed's "nccl" backend (= RCCL) and never calls these.
//
// librccl is dlopen()ed on first use (GD_RCCL_PATH, then librccl.so.1 / librccl.so): libgandanet_hip.so carries no link
// dependency on it, so a process that already holds another copy (PyTorch bundles one) is not disturbed at load time.
// One communicator per process (one process per GPU); calls are asynchronous on the caller's stream.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.h"
#include "../../include/gandanet.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g;
ncclComm_t g_comm = nullptr;
int g_world = 0;

bool load_rccl() {
    if (g.h) return true;
    const char* names[] = {getenv("GD_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n) continue;
        g.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g.h) break;
    }
    if (!g.h) {
        gd_set_error("gd_comm: librccl.so not found (set GD_RCCL_PATH)");
        return false;
    }
#define GD_SYM(field, name)                                                   \
    *(void**)(&g.field) = dlsym(g.h, name);                                   \
    if (!g.field) {                                                           \
        gd_set_error("gd_comm: librccl lacks " name);                         \
        dlclose(g.h);                                                         \
        g.h = nullptr;                                                        \
        return false;                                                         \
    }
    GD_SYM(GetUniqueId, "ncclGetUniqueId")
    GD_SYM(CommInitRank, "ncclCommInitRank")
    GD_SYM(CommDestroy, "ncclCommDestroy")
    GD_SYM(AllReduce, "ncclAllReduce")
    GD_SYM(ReduceScatter, "ncclReduceScatter")
    GD_SYM(AllGather, "ncclAllGather")
    GD_SYM(GetErrorString, "ncclGetErrorString")
#undef GD_SYM
    return true;
}
int fail(const char* what, ncclResult_t r) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "rccl error");
    gd_set_error(buf);
    return -3;
}
bool dtype_of(int dtype, ncclDataType_t* t) {
    if (dtype == 0) *t = ncclFloat32;
    else if (dtype == 1) *t = ncclBfloat16;
    else if (dtype == 2) *t = ncclFloat16;
    else return false;
    return true;
}
}  // namespace

extern "C" int gd_comm_unique_id(char* id128) {
    GD_CHECK_ARG(id128, "gd_comm_unique_id: null buffer");
    if (!load_rccl()) return -3;
    ncclUniqueId id;
    const ncclResult_t r = g.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}
extern "C" int gd_comm_init(int rank, int world, const char* id128) {
    GD_CHECK_ARG(id128 && world >= 1 && rank >= 0 && rank < world, "gd_comm_init: bad arguments");
    GD_CHECK_ARG(!g_comm, "gd_comm_init: a communicator already exists (one per process)");
    if (!load_rccl()) return -3;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = g.CommInitRank(&g_comm, world, id, rank);
    if (r != ncclSuccess) {
        g_comm = nullptr;
        return fail("ncclCommInitRank", r);
    }
    g_world = world;
    return 0;
}
extern "C" int gd_comm_world(void) { return g_comm ? g_world : 0; }
extern "C" int gd_allreduce(void* buf, size_t n, int dtype, void* stream) {
    ncclDataType_t t;
    GD_CHECK_ARG(g_comm, "gd_allreduce: no communicator (gd_comm_init)");
    GD_CHECK_ARG(buf && n > 0 && dtype_of(dtype, &t), "gd_allreduce: bad arguments");
    const ncclResult_t r = g.AllReduce(buf, buf, n, t, ncclSum, g_comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail("ncclAllReduce", r);
}
extern "C" int gd_reduce_scatter(const void* send, void* recv, size_t recv_n, int dtype, void* stream) {
    ncclDataType_t t;
    GD_CHECK_ARG(g_comm, "gd_reduce_scatter: no communicator (gd_comm_init)");
    GD_CHECK_ARG(send && recv && recv_n > 0 && dtype_of(dtype, &t), "gd_reduce_scatter: bad arguments");
    const ncclResult_t r = g.ReduceScatter(send, recv, recv_n, t, ncclSum, g_comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail("ncclReduceScatter", r);
}
extern "C" int gd_allgather(const void* send, void* recv, size_t send_n, int dtype, void* stream) {
    ncclDataType_t t;
    GD_CHECK_ARG(g_comm, "gd_allgather: no communicator (gd_comm_init)");
    GD_CHECK_ARG(send && recv && send_n > 0 && dtype_of(dtype, &t), "gd_allgather: bad arguments");
    const ncclResult_t r = g.AllGather(send, recv, send_n, t, g_comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail("ncclAllGather", r);
}
extern "C" int gd_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = g.CommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    return r == ncclSuccess ? 0 : fail("ncclCommDestroy", r);
}
