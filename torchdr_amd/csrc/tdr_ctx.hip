// Multi-GPU context of the hot path: an RCCL communicator behind the C ABI ("no global mutable state
// except an opaque ctx* holding the RCCL communicator").  One process per GPU; the collectives are enqueued ON THE
// CALLER'S STREAM, so they order with the kernels around them without host synchronisation and can be captured into the
// HIP graphs of the loop object (tdr_umap_loop_*).
//
// Replaces, for the optimisation loop (citations under /root/reference/torchdr):
//   affinity_matcher.py:395-413   zero-padded all-reduce of the rows each rank stepped  -> in-place all-gather of the rows
//   affinity_matcher.py:425       all-reduce of a full gradient                         -> tdr_ctx_allreduce_f32
// librccl is opened with dlopen at context creation (the path is the caller's: normally the copy PyTorch has already
// loaded, so that one RCCL runtime serves both), not linked: the library keeps loading on boxes without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include "tdr_common.h"

namespace {

struct RcclApi {
    void* dl;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
};

int load_rccl(const char* path, RcclApi* api) {
    const char* p = (path && path[0]) ? path : "librccl.so";
    void* dl = dlopen(p, RTLD_NOW | RTLD_LOCAL);
    if (!dl && path && path[0]) dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!dl) return TDR_ERR_UNSUPPORTED;
    api->dl = dl;
#define TDR_SYM(field, name)                                              \
    *(void**)(&api->field) = dlsym(dl, name);                             \
    if (!api->field) { dlclose(dl); api->dl = nullptr; return TDR_ERR_UNSUPPORTED; }
    TDR_SYM(GetUniqueId, "ncclGetUniqueId")
    TDR_SYM(CommInitRank, "ncclCommInitRank")
    TDR_SYM(CommDestroy, "ncclCommDestroy")
    TDR_SYM(AllGather, "ncclAllGather")
    TDR_SYM(AllReduce, "ncclAllReduce")
    TDR_SYM(Broadcast, "ncclBroadcast")
    TDR_SYM(GroupStart, "ncclGroupStart")
    TDR_SYM(GroupEnd, "ncclGroupEnd")
#undef TDR_SYM
    return TDR_OK;
}

struct TdrCtx {
    RcclApi api;
    ncclComm_t comm;
    int rank, world;
    int64_t n_total;  // rows of the replicated embedding; chunks as distributed/__init__.py:209-219
};

inline void chunk_of(int64_t n, int world, int r, int64_t* start, int64_t* rows) {
    const int64_t base = n / world, rem = n % world;
    if (r < rem) { *start = r * (base + 1); *rows = base + 1; }
    else { *start = r * base + rem; *rows = base; }
}

// RCCL error codes are returned as 1000 + code so that they cannot be mistaken for a hipError_t
inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? TDR_OK : 1000 + (int)r; }

}  // namespace

extern "C" {

/* A fresh RCCL unique id (128 bytes) for tdr_ctx_create; called by ONE rank, which hands the bytes to the others (any
 * transport: the Python side broadcasts them through torch.distributed).  rccl_path: librccl to open ("" = default). */
int tdr_ctx_unique_id(const char* rccl_path, void* out128) {
    if (!out128) return TDR_ERR_BAD_ARG;
    RcclApi api;
    const int rc = load_rccl(rccl_path, &api);
    if (rc != TDR_OK) return rc;
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r == ncclSuccess) memcpy(out128, &id, NCCL_UNIQUE_ID_BYTES);
    dlclose(api.dl);
    return rc_of(r);
}

/* Communicator of `world` ranks (this process = `rank`, its GPU = the current HIP device) for an embedding of n_total
 * rows sharded by the reference's chunk rule.  Collective: every rank calls it with the same unique id. */
int tdr_ctx_create(void** ctx, int rank, int world, const void* unique_id128, const char* rccl_path, int64_t n_total) {
    if (!ctx || !unique_id128 || world <= 0 || rank < 0 || rank >= world || n_total <= 0) return TDR_ERR_BAD_ARG;
    TdrCtx* c = new TdrCtx();
    int rc = load_rccl(rccl_path, &c->api);
    if (rc != TDR_OK) { delete c; return rc; }
    ncclUniqueId id;
    memcpy(&id, unique_id128, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = c->api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { dlclose(c->api.dl); delete c; return rc_of(r); }
    c->rank = rank; c->world = world; c->n_total = n_total;
    *ctx = c;
    return TDR_OK;
}

/* Re-target the communicator at an embedding of n_total rows (the chunk rule is applied to it by the collectives below):
 * creating an RCCL communicator costs tens of milliseconds, so ONE is kept per process and reused by every fit. */
int tdr_ctx_set_rows(void* ctx, int64_t n_total) {
    TdrCtx* c = (TdrCtx*)ctx;
    if (!c || n_total <= 0) return TDR_ERR_BAD_ARG;
    c->n_total = n_total;
    return TDR_OK;
}

/* In-place all-gather of the row chunks of Z (n_total, nc): on entry rows [start_r, start_r + rows_r) of every rank r
 * hold what rank r computed; on return every rank holds all rows.  Equal chunks: one ncclAllGather whose send buffer is
 * the rank's own slot of the receive buffer; uneven chunks: one grouped ncclBroadcast per rank.  Enqueued on `stream`.
 * Signature = the `gather` callback of tdr_umap_loop_desc. */
int tdr_ctx_allgather_rows(void* ctx, float* Z, int nc, void* stream) {
    TdrCtx* c = (TdrCtx*)ctx;
    if (!c || !Z || nc <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (c->n_total % c->world == 0) {
        const int64_t rows = c->n_total / c->world;
        return rc_of(c->api.AllGather(Z + (size_t)c->rank * rows * nc, Z, (size_t)(rows * nc), ncclFloat, c->comm, st));
    }
    ncclResult_t r = c->api.GroupStart();
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
        int64_t start, rows;
        chunk_of(c->n_total, c->world, p, &start, &rows);
        float* buf = Z + (size_t)start * nc;
        r = c->api.Broadcast(buf, buf, (size_t)(rows * nc), ncclFloat, p, c->comm, st);
    }
    const ncclResult_t e = c->api.GroupEnd();
    return rc_of(r != ncclSuccess ? r : e);
}

/* In-place sum all-reduce of `count` floats (affinity_matcher.py:425; scalar partition functions, gradient norms). */
int tdr_ctx_allreduce_f32(void* ctx, float* buf, int64_t count, void* stream) {
    TdrCtx* c = (TdrCtx*)ctx;
    if (!c || !buf || count <= 0) return TDR_ERR_BAD_ARG;
    return rc_of(c->api.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
}

int tdr_ctx_destroy(void* ctx) {
    TdrCtx* c = (TdrCtx*)ctx;
    if (!c) return TDR_ERR_BAD_ARG;
    const ncclResult_t r = c->api.CommDestroy(c->comm);
    dlclose(c->api.dl);
    delete c;
    return rc_of(r);
}

}  // extern "C"
