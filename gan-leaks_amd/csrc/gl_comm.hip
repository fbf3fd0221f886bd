// The one exchange step of the sharded attack: min over ranks of the packed (distance, global index) keys, as an RCCL
// all-reduce on the context's HIP stream (ncclAllReduce(ncclMin, ncclUint64) on 8 Q bytes, latency-bound over xGMI).
// The reference is single-device (attack_models/fbb.py:40); this is the multi-GPU extension the project brief names.
//
// RCCL is bound at run time (dlopen) so that the library loads, and every single-GPU entry point works, on a machine without it;
// the first gl_comm_* call resolves librccl.  Nothing here synchronises the host: the reduce is queued behind the search kernel
// that produced the keys and ahead of gl_keys_unpack on the same stream.
#include "gl_common.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>

struct gl_comm {
    gl_ctx *ctx;
    ncclComm_t comm;
    int rank, nranks;
};

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[4] = {getenv("GANLEAKS_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            snprintf(api.why, sizeof(api.why), "%s", dlerror());
        }
        if (!api.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if (!p) { ok = false; snprintf(api.why, sizeof(api.why), "librccl has no symbol %s", name); }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return &api;
}

}  // namespace

#define GL_RCCL_API(api)                                                                     \
    RcclApi *api = rccl();                                                                   \
    if (!api->handle) {                                                                      \
        gl_set_error("RCCL is not available: %s", api->why);                                 \
        return GL_ERR_RCCL;                                                                  \
    }

#define GL_RCCL(api, expr)                                                                   \
    do {                                                                                     \
        ncclResult_t _r = (expr);                                                            \
        if (_r != ncclSuccess) {                                                             \
            gl_set_error("%s failed: %s (%s:%d)", #expr, api->GetErrorString(_r), __FILE__, __LINE__); \
            return GL_ERR_RCCL;                                                              \
        }                                                                                    \
    } while (0)

extern "C" {

int gl_comm_unique_id(void *id_out)
{
    GL_REQUIRE(id_out, "gl_comm_unique_id: NULL id_out");
    static_assert(GL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "GL_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    GL_RCCL_API(api);
    ncclUniqueId id;
    GL_RCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, id.internal, GL_COMM_ID_BYTES);
    return GL_OK;
}

int gl_comm_init_rank(gl_ctx *ctx, const void *id, int rank, int nranks, gl_comm **out)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && id && out, "gl_comm_init_rank: NULL argument");
    GL_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "gl_comm_init_rank: rank %d of %d", rank, nranks);
    GL_RCCL_API(api);
    GL_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, GL_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    GL_RCCL(api, api->CommInitRank(&comm, nranks, uid, rank));
    gl_comm *c = new gl_comm();
    c->ctx = ctx; c->comm = comm; c->rank = rank; c->nranks = nranks;
    *out = c;
    return GL_OK;
}

int gl_comm_init_all(gl_ctx *const *ctxs, int n, gl_comm **out_comms)
{
    GL_REQUIRE(ctxs && out_comms && n >= 1 && n <= 64, "gl_comm_init_all: bad argument (n = %d)", n);
    int dev[64];
    for (int i = 0; i < n; ++i) {
        GL_REQUIRE(ctxs[i], "gl_comm_init_all: NULL context %d", i);
        dev[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (dev[j] == dev[i]) {
                // RCCL refuses a communicator with two ranks on one device; say so before it does (callers fall back to a host merge)
                gl_set_error("gl_comm_init_all: contexts %d and %d are both on device %d; one rank per GPU", j, i, dev[i]);
                return GL_ERR_RCCL;
            }
    }
    GL_RCCL_API(api);
    ncclComm_t comms[64];
    GL_RCCL(api, api->CommInitAll(comms, n, dev));
    for (int i = 0; i < n; ++i) {
        gl_comm *c = new gl_comm();
        c->ctx = ctxs[i]; c->comm = comms[i]; c->rank = i; c->nranks = n;
        out_comms[i] = c;
    }
    return GL_OK;
}

int gl_comm_destroy(gl_comm *c)
{
    if (!c) return GL_OK;
    gl_make_current(c->ctx);
    RcclApi *api = rccl();
    (void)hipStreamSynchronize(c->ctx->stream);
    if (api->handle && c->comm) (void)api->CommDestroy(c->comm);
    delete c;
    return GL_OK;
}

int gl_comm_abort(gl_comm *c)
{
    // no stream synchronisation here: the point is to end a reduce that can never complete (a peer rank failed before queueing its half)
    if (!c) return GL_OK;
    RcclApi *api = rccl();
    ncclResult_t r = ncclSuccess;
    if (api->handle && c->comm) r = api->CommAbort(c->comm);
    delete c;
    if (r != ncclSuccess) {
        gl_set_error("ncclCommAbort failed: %s", api->GetErrorString(r));
        return GL_ERR_RCCL;
    }
    return GL_OK;
}

int gl_comm_rank(const gl_comm *c, int *out_rank, int *out_nranks)
{
    GL_REQUIRE(c && out_rank && out_nranks, "gl_comm_rank: NULL argument");
    *out_rank = c->rank;
    *out_nranks = c->nranks;
    return GL_OK;
}

int gl_allreduce_min_keys(gl_comm *c, uint64_t *keys_dev, int64_t nq)
{
    GL_REQUIRE(c && nq >= 0, "gl_allreduce_min_keys: bad argument");
    if (nq == 0) return GL_OK;
    GL_REQUIRE(keys_dev, "gl_allreduce_min_keys: NULL keys_dev");
    gl_make_current(c->ctx);
    GL_RCCL_API(api);
    // in place, unsigned 64-bit minimum: smallest distance first, then smallest global index (torch.min's first occurrence, fbb.py:86)
    GL_RCCL(api, api->AllReduce(keys_dev, keys_dev, (size_t)nq, ncclUint64, ncclMin, c->comm, c->ctx->stream));
    return GL_OK;
}

int gl_allgather_rows(gl_comm *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank)
{
    GL_REQUIRE(c && bytes_per_rank >= 0, "gl_allgather_rows: bad argument");
    if (bytes_per_rank == 0) return GL_OK;
    GL_REQUIRE(send_dev && recv_dev, "gl_allgather_rows: NULL device pointer");
    gl_make_current(c->ctx);
    GL_RCCL_API(api);
    GL_RCCL(api, api->AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return GL_OK;
}

int gl_comm_group_start(void)
{
    GL_RCCL_API(api);
    GL_RCCL(api, api->GroupStart());
    return GL_OK;
}

int gl_comm_group_end(void)
{
    GL_RCCL_API(api);
    GL_RCCL(api, api->GroupEnd());
    return GL_OK;
}

}  // extern "C"
