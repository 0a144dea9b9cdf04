// Peer exchange of the rows every rank stepped (round 4): an in-place all-gather of the row chunks of
// the replicated embedding WITHOUT a collective library -- every rank WRITES its chunk straight into a staging block of every
// peer over xGMI (all W - 1 links of the fully connected node at once: a ring all-gather moves the same bytes hop by hop),
// raises a generation flag at each peer, waits for the W - 1 flags raised at it, and copies the staged chunks into its
// embedding.  Replaces affinity_matcher.py:395-413 (the reference all-reduces a zero-padded gradient) for the estimators in
// which only a rank's own rows move (UMAP, COSNE), as tdr_ctx_allgather_rows does over RCCL; same callback signature.
//
// Memory model (one process per GPU, peers mapped with hipIpcOpenMemHandle):
//   stage   (capacity floats)  per rank, FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) where the
//           runtime offers it: remote writes and local reads are coherent at system scope without relying on what a kernel
//           boundary does to the L2s.  The embedding itself stays an ordinary (coarse-grained, L2-cached) allocation -- it is
//           gathered 52 M times per iteration -- and is only written by this rank's own kernels.
//   flags   (W x 32 ints) per rank, fine-grained: flags[p * 32] = generation of the last chunk rank p delivered here.
// One exchange = two launches on the caller's stream:
//   push   : each block copies its share of this rank's rows into every peer's stage, __threadfence_system(), draws a
//            ticket; the last block raises the flags (system-scope release stores).
//   pull   : thread p of every block spins (system-scope acquire loads, s_sleep, bounded) until flag p has reached this
//            generation, then the block copies its share of the staged rows of the other ranks into the embedding.
// A rank never overwrites a peer's stage before the peer has consumed it: generation g + 1 is pushed only after this rank's
// pull of generation g returned, and a peer's pull of g cannot return before this rank's flag g arrived -- but a FAST rank
// could push g + 1 while a slow peer still copies g.  Hence two stages per rank, used alternately (g & 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "tdr_common.h"

namespace {

constexpr int PX_MAX_WORLD = 16;
constexpr int PX_FLAG_STRIDE = 32;    // ints between two ranks' flags (128 B: one line each)
constexpr long long PX_SPIN_LIMIT = 1LL << 22;   // bounded wait (a few seconds): a lost peer must not hang the device

struct PeerX {
    int rank, world;
    int64_t capacity;           // floats per stage
    int64_t n_total;
    float* stage[2][PX_MAX_WORLD];   // [parity][rank]: this rank's own stages at [..][rank], peers' mapped ones elsewhere
    int* flags[PX_MAX_WORLD];
    void* own_stage;            // allocation holding both of this rank's stages
    void* own_flags;
    void* mapped_stage[PX_MAX_WORLD];
    void* mapped_flags[PX_MAX_WORLD];
    int* ticket;                // device ints: [0] push kernel's last-block counter, [1] pull kernel's, [2] the GENERATION of the last
                                // completed exchange (device-resident since round 6: an exchange bakes nothing into its kernel arguments,
                                // so windows that contain it can be captured into HIP graphs and replayed)
    int* err;                   // device int: 1 = a wait ran into its limit
    int gen;
    int fine_grained;
    bool opened;
    long long spin_limit;       // bounded flag wait of the pull kernel (tdr_peerx_set_wait_limit)
};

struct PushParams {
    const float* src;           // this rank's rows in the embedding
    int64_t count;              // floats
    int64_t dst_off;            // offset of this rank's chunk in a stage
    float* dst[2][PX_MAX_WORLD];   // [parity][rank]
    int* flag[PX_MAX_WORLD];    // flag word of THIS rank at every peer
    int world, rank;
    int* ticket;                // [0] this kernel's counter, [2] generation of the last completed exchange
};

__global__ __launch_bounds__(256) void peerx_push_kernel(const PushParams P) {
    const int gen = P.ticket[2] + 1;      // this exchange (the pull kernel of the previous one stored its generation before this launch began)
    const int par = gen & 1;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.count; i += stride) {
        const float v = P.src[i];
        for (int p = 0; p < P.world; ++p)
            if (p != P.rank) P.dst[par][p][P.dst_off + i] = v;
    }
    __threadfence_system();
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(P.ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) *P.ticket = 0;
    __threadfence_system();
    if ((int)threadIdx.x < P.world && (int)threadIdx.x != P.rank)
        __hip_atomic_store(P.flag[threadIdx.x], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PullParams {
    float* Z;                   // the embedding (n_total x nc floats)
    const float* stage[2];      // this rank's stages, by parity
    const int* flags;           // this rank's flag block
    int64_t own_off, own_count; // floats of this rank's own chunk (not copied)
    int64_t total;              // n_total * nc
    int world, rank;
    int* ticket;                // [1] this kernel's counter, [2] generation of the last completed exchange
    int* err;
    long long spin_limit;
};

__global__ __launch_bounds__(256) void peerx_pull_kernel(const PullParams P) {
    const int gen = P.ticket[2] + 1;
    const float* stage = P.stage[gen & 1];
    if ((int)threadIdx.x < P.world && (int)threadIdx.x != P.rank) {
        const int* f = P.flags + (size_t)threadIdx.x * PX_FLAG_STRIDE;
        long long spins = 0;
        // generations only grow: "reached" = not behind (wrap-safe signed difference)
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - gen < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > P.spin_limit) { atomicExch(P.err, 1); break; }
        }
    }
    __syncthreads();
    __threadfence_system();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.total; i += stride) {
        if (i >= P.own_off && i < P.own_off + P.own_count) continue;
        P.Z[i] = __builtin_nontemporal_load(stage + i);
    }
    // the last block to finish closes the exchange: every block has read the generation by the time it draws its ticket
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(P.ticket + 1, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) { P.ticket[1] = 0; P.ticket[2] = gen; }
}

inline void chunk_of(int64_t n, int world, int r, int64_t* start, int64_t* rows) {
    const int64_t base = n / world, rem = n % world;
    if (r < rem) { *start = r * (base + 1); *rows = base + 1; }
    else { *start = r * base + rem; *rows = base; }
}

}  // namespace

// ---- loopback stand-in of one exchange (measurement of a rank's share on ONE GPU, tools/rank_share.py) -----------------------
// Rank r of a W-rank fit is run alone; what the exchange would move is moved locally: push = the rank's chunk written W - 1
// times (what it sends over its W - 1 links), pull = the N - chunk staged rows of the peers copied into the embedding (they
// hold the peers' rows as they stood at the start of the fit: the peers do not exist).  Same two launches, same bytes, no link.
struct EmulX {
    int rank, world;
    int64_t n_total, capacity;     // rows of the embedding; floats of the stage
    float* stage;                  // the peers' rows (n_total x nc), filled from the embedding at the first exchange
    float* scratch;                // (W - 1) copies of the rank's chunk
    bool primed;
};
__global__ __launch_bounds__(256) void emulx_push_kernel(const float* __restrict__ src, int64_t count, float* __restrict__ dst, int copies) {
    const int64_t n4 = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        for (int c = 0; c < copies; ++c) reinterpret_cast<float4*>(dst + (size_t)c * count)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3))
        for (int c = 0; c < copies; ++c) dst[(size_t)c * count + (n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
}
__global__ __launch_bounds__(256) void emulx_pull_kernel(const float* __restrict__ stage, float* __restrict__ Z, int64_t lo, int64_t hi, int64_t total) {
    // floats [0, lo) and [hi, total) of the stage into Z (the rank's own floats [lo, hi) stay)
    const int64_t n = total - (hi - lo);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t j = i < lo ? i : i + (hi - lo);
        Z[j] = stage[j];
    }
}

extern "C" {

/* Context of the peer exchange for `world` <= 16 ranks on one node (this process = `rank`, its GPU = the current HIP device):
 * two stages of capacity_floats floats and the flag block, fine-grained where the runtime allows. */
int tdr_peerx_create(void** out, int rank, int world, int64_t capacity_floats) {
    if (!out || world < 2 || world > PX_MAX_WORLD || rank < 0 || rank >= world || capacity_floats <= 0) return TDR_ERR_BAD_ARG;
    PeerX* c = new PeerX();
    memset(c, 0, sizeof(PeerX));
    c->rank = rank; c->world = world; c->capacity = capacity_floats; c->gen = 0; c->opened = false; c->spin_limit = PX_SPIN_LIMIT;
    const size_t sbytes = 2 * (size_t)capacity_floats * sizeof(float);
    const size_t fbytes = (size_t)(PX_MAX_WORLD * PX_FLAG_STRIDE + 64) * sizeof(int);
    c->fine_grained = 1;
    if (hipExtMallocWithFlags(&c->own_stage, sbytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        c->fine_grained = 0;
        if (hipMalloc(&c->own_stage, sbytes) != hipSuccess) { delete c; return TDR_ERR_WORKSPACE; }
    }
    if (hipExtMallocWithFlags(&c->own_flags, fbytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        c->fine_grained = 0;
        if (hipMalloc(&c->own_flags, fbytes) != hipSuccess) { (void)hipFree(c->own_stage); delete c; return TDR_ERR_WORKSPACE; }
    }
    if (hipMemset(c->own_flags, 0, fbytes) != hipSuccess) { (void)hipFree(c->own_stage); (void)hipFree(c->own_flags); delete c; return TDR_ERR_WORKSPACE; }
    c->stage[0][rank] = (float*)c->own_stage;
    c->stage[1][rank] = (float*)c->own_stage + capacity_floats;
    c->flags[rank] = (int*)c->own_flags;
    c->ticket = (int*)c->own_flags + PX_MAX_WORLD * PX_FLAG_STRIDE;
    c->err = c->ticket + 16;
    *out = c;
    return TDR_OK;
}

/* 128 bytes: the IPC handles of this rank's stage and flag allocations, to be handed to every peer (any transport). */
int tdr_peerx_handles(void* ctx, void* out128) {
    PeerX* c = (PeerX*)ctx;
    if (!c || !out128) return TDR_ERR_BAD_ARG;
    hipIpcMemHandle_t hs, hf;
    hipError_t e = hipIpcGetMemHandle(&hs, c->own_stage);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hf, c->own_flags);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(out128, &hs, 64);
    memcpy((char*)out128 + 64, &hf, 64);
    return TDR_OK;
}

/* all_handles: world x 128 bytes (rank r's tdr_peerx_handles output at offset 128 r).  Maps every peer's stage and flags. */
int tdr_peerx_open(void* ctx, const void* all_handles) {
    PeerX* c = (PeerX*)ctx;
    if (!c || !all_handles || c->opened) return TDR_ERR_BAD_ARG;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t hs, hf;
        memcpy(&hs, (const char*)all_handles + 128 * (size_t)p, 64);
        memcpy(&hf, (const char*)all_handles + 128 * (size_t)p + 64, 64);
        hipError_t e = hipIpcOpenMemHandle(&c->mapped_stage[p], hs, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&c->mapped_flags[p], hf, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
        c->stage[0][p] = (float*)c->mapped_stage[p];
        c->stage[1][p] = (float*)c->mapped_stage[p] + c->capacity;
        c->flags[p] = (int*)c->mapped_flags[p];
    }
    c->opened = true;
    return TDR_OK;
}

int tdr_peerx_set_rows(void* ctx, int64_t n_total) {
    PeerX* c = (PeerX*)ctx;
    if (!c || n_total <= 0) return TDR_ERR_BAD_ARG;
    c->n_total = n_total;
    return TDR_OK;
}

/* Spins (each ~ an s_sleep(8) + one system-scope load) a pull waits for a peer's flag before it gives up and sets the error
 * word; the default (2^22, a few seconds) bounds the time a lost peer can hold the device.  Ranks that time-slice one device
 * or a slow host may legitimately fall further behind: raise it there. */
int tdr_peerx_set_wait_limit(void* ctx, int64_t spins) {
    PeerX* c = (PeerX*)ctx;
    if (!c || spins < 1) return TDR_ERR_BAD_ARG;
    c->spin_limit = (long long)spins;
    return TDR_OK;
}

int tdr_peerx_fine_grained(void* ctx) { return ctx ? ((PeerX*)ctx)->fine_grained : 0; }

/* In-place all-gather of the row chunks of Z (n_total, nc) -- tdr_ctx_allgather_rows's contract and callback signature
 * (chunks by the reference's rule, distributed/__init__.py:209-219).  n_total * nc must fit the stage capacity.  Collective:
 * every rank calls it the same number of times.  Enqueued on `stream`; a peer that never arrives sets the error flag
 * (tdr_peerx_error) instead of hanging the device. */
int tdr_peerx_allgather_rows(void* ctx, float* Z, int nc, void* stream) {
    PeerX* c = (PeerX*)ctx;
    if (!c || !Z || nc <= 0 || !c->opened || c->n_total <= 0) return TDR_ERR_BAD_ARG;
    if (c->n_total * nc > c->capacity) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int64_t start, rows;
    chunk_of(c->n_total, c->world, c->rank, &start, &rows);
    ++c->gen;       // host-side count of enqueued exchanges (statistics only: the kernels read the generation from device memory)
    PushParams P;
    P.src = Z + (size_t)start * nc; P.count = rows * nc; P.dst_off = start * nc; P.world = c->world; P.rank = c->rank;
    P.ticket = c->ticket;
    for (int p = 0; p < c->world; ++p) {
        P.dst[0][p] = c->stage[0][p]; P.dst[1][p] = c->stage[1][p];
        P.flag[p] = c->flags[p] + (size_t)c->rank * PX_FLAG_STRIDE;
    }
    int64_t pb = (P.count + 1023) / 1024;
    if (pb < 1) pb = 1;
    if (pb > 256) pb = 256;
    hipLaunchKernelGGL(peerx_push_kernel, dim3((unsigned)pb), dim3(256), 0, st, P);
    PullParams Q;
    Q.Z = Z; Q.stage[0] = c->stage[0][c->rank]; Q.stage[1] = c->stage[1][c->rank]; Q.flags = c->flags[c->rank];
    Q.own_off = start * nc; Q.own_count = rows * nc;
    Q.total = c->n_total * nc; Q.world = c->world; Q.rank = c->rank; Q.ticket = c->ticket; Q.err = c->err; Q.spin_limit = c->spin_limit;
    int64_t qb = (Q.total + 2047) / 2048;
    if (qb < 1) qb = 1;
    if (qb > 128) qb = 128;      // few blocks: they spin, and ranks that share a device (tests) must still get CUs
    hipLaunchKernelGGL(peerx_pull_kernel, dim3((unsigned)qb), dim3(256), 0, st, Q);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* 1 when a wait of this context ran into its limit since creation (synchronises the device). */
int tdr_peerx_error(void* ctx) {
    PeerX* c = (PeerX*)ctx;
    if (!c) return TDR_ERR_BAD_ARG;
    int v = 0;
    if (hipMemcpy(&v, c->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return v;
}

int tdr_peerx_destroy(void* ctx) {
    PeerX* c = (PeerX*)ctx;
    if (!c) return TDR_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        if (c->mapped_stage[p]) (void)hipIpcCloseMemHandle(c->mapped_stage[p]);
        if (c->mapped_flags[p]) (void)hipIpcCloseMemHandle(c->mapped_flags[p]);
    }
    (void)hipFree(c->own_stage);
    (void)hipFree(c->own_flags);
    delete c;
    return TDR_OK;
}


/* Loopback stand-in of tdr_peerx_allgather_rows for a rank that runs ALONE (measurement only; same callback signature): the
 * bytes a rank of `world` would send (its chunk, world - 1 times) and receive (all other rows) are moved inside this GPU. */
int tdr_emulx_create(void** out, int rank, int world, int64_t n_total, int nc) {
    if (!out || world < 2 || world > PX_MAX_WORLD || rank < 0 || rank >= world || n_total <= 0 || nc <= 0) return TDR_ERR_BAD_ARG;
    EmulX* c = new EmulX();
    c->rank = rank; c->world = world; c->n_total = n_total; c->capacity = n_total * nc; c->primed = false;
    const int64_t chunk = (n_total + world - 1) / world * nc;
    if (hipMalloc((void**)&c->stage, (size_t)c->capacity * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->scratch, (size_t)chunk * (world - 1) * sizeof(float)) != hipSuccess) { delete c; return TDR_ERR_WORKSPACE; }
    *out = c;
    return TDR_OK;
}
int tdr_emulx_allgather_rows(void* ctx, float* Z, int nc, void* stream) {
    EmulX* c = (EmulX*)ctx;
    if (!c || !Z || nc <= 0 || c->n_total * nc > c->capacity) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t base = c->n_total / c->world, rem = c->n_total % c->world;
    const int64_t r0 = c->rank < rem ? c->rank * (base + 1) : c->rank * base + rem;
    const int64_t rows = base + (c->rank < rem ? 1 : 0);
    const int64_t total = c->n_total * nc, lo = r0 * nc, hi = (r0 + rows) * nc;
    if (!c->primed) {
        if (hipMemcpyAsync(c->stage, Z, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return TDR_ERR_WORKSPACE;
        c->primed = true;
    }
    hipLaunchKernelGGL(emulx_push_kernel, dim3(256), dim3(256), 0, st, (const float*)(Z + lo), hi - lo, c->scratch, c->world - 1);
    hipLaunchKernelGGL(emulx_pull_kernel, dim3(1024), dim3(256), 0, st, (const float*)c->stage, Z, lo, hi, total);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}
int tdr_emulx_destroy(void* ctx) {
    EmulX* c = (EmulX*)ctx;
    if (!c) return TDR_ERR_BAD_ARG;
    (void)hipFree(c->stage); (void)hipFree(c->scratch);
    delete c;
    return TDR_OK;
}

}  // extern "C"
