// Collectives of the data-parallel TTA step behind the C ABI: one RCCL communicator per process
// (= per GPU).  Replaces the role of reference src/ddp.py:6-70 (GatherLayer / all_gather helpers over torch.distributed)
// for callers that bind libneat_hip.so without torch.distributed: an all-gather of the per-rank [text | coarse] feature
// shards and of the gallery embeddings, and a sum all-reduce of the flat LayerNorm-gradient buffer.
//
// RCCL is resolved at RUN time (dlopen "librccl.so.1"): libneat_hip.so carries no link-time dependency on it, loads on
// a box without RCCL, and inside a PyTorch process binds to the copy PyTorch has already loaded (same SONAME) instead of
// bringing a second one.  Every failure returns NEAT_ERR_COMM with RCCL's own message in neat_last_error().
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.h"

namespace neat {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static ncclComm_t g_comm = nullptr;
static int g_world = 0, g_rank = -1;

static int load_rccl() {
    if (g_rccl.handle) return NEAT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_error(NEAT_ERR_COMM, "neat_comm: cannot load librccl.so (%s)", dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.AllReduce || !a.CommDestroy || !a.GetErrorString)
        return set_error(NEAT_ERR_COMM, "neat_comm: librccl.so lacks an expected symbol");
    g_rccl = a;
    return NEAT_OK;
}

static int check_rccl(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return NEAT_OK;
    return set_error(NEAT_ERR_COMM, "%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
}

}  // namespace neat

using namespace neat;

static_assert(sizeof(ncclUniqueId) == NEAT_COMM_ID_BYTES, "NEAT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int neat_comm_unique_id(void* id) {
    NEAT_REQUIRE(id, NEAT_ERR_ARG, "neat_comm_unique_id: null pointer");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId u;
    if (int rc = check_rccl(g_rccl.GetUniqueId(&u), "ncclGetUniqueId")) return rc;
    memcpy(id, &u, sizeof(u));
    return NEAT_OK;
}

extern "C" int neat_comm_init(const void* id, int world, int rank) {
    NEAT_REQUIRE(id, NEAT_ERR_ARG, "neat_comm_init: null id");
    NEAT_REQUIRE(world >= 1 && rank >= 0 && rank < world, NEAT_ERR_ARG, "neat_comm_init: rank %d of world %d", rank, world);
    NEAT_REQUIRE(!g_comm, NEAT_ERR_COMM, "neat_comm_init: a communicator already exists (one per process); call neat_comm_destroy first");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    if (int rc = check_rccl(g_rccl.CommInitRank(&c, world, u, rank), "ncclCommInitRank")) return rc;
    g_comm = c;
    g_world = world;
    g_rank = rank;
    return NEAT_OK;
}

extern "C" int neat_comm_world(void) { return g_comm ? g_world : 0; }
extern "C" int neat_comm_rank(void) { return g_comm ? g_rank : -1; }

extern "C" int neat_comm_allgather(const void* send, void* recv, long bytes, void* stream) {
    NEAT_REQUIRE(g_comm, NEAT_ERR_COMM, "neat_comm_allgather: no communicator (neat_comm_init)");
    NEAT_REQUIRE(send && recv && bytes > 0, NEAT_ERR_ARG, "neat_comm_allgather: null pointer or bytes=%ld", bytes);
    return check_rccl(g_rccl.AllGather(send, recv, (size_t)bytes, ncclInt8, g_comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int neat_comm_allreduce_f32(const float* send, float* recv, long n, void* stream) {
    NEAT_REQUIRE(g_comm, NEAT_ERR_COMM, "neat_comm_allreduce_f32: no communicator (neat_comm_init)");
    NEAT_REQUIRE(send && recv && n > 0, NEAT_ERR_ARG, "neat_comm_allreduce_f32: null pointer or n=%ld", n);
    return check_rccl(g_rccl.AllReduce(send, recv, (size_t)n, ncclFloat32, ncclSum, g_comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int neat_comm_destroy(void) {
    if (!g_comm) return NEAT_OK;
    const int rc = check_rccl(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
    g_comm = nullptr;
    g_world = 0;
    g_rank = -1;
    return rc;
}
