// K5s -- the UMAP gradient loop on per-iteration active-edge lists ("scheduled epochs").
//
// Replaces the same reference lines as the per-step kernels of tdr_embed.hip
//   neighbor_embedding/umap.py:236-264   attraction over the edges whose epoch counter fires (epoch_of_next_sample)
//   neighbor_embedding/umap.py:266-292   repulsion over 5 * (#fired edges) sampled negatives
//   neighbor_embedding/base.py:617-649   negative sampling
// with a different data flow.  Which edge fires at which iteration does not depend on the embedding:
// `act = next <= n_iter + 1; next[act] += eps_per[act]` (umap.py:243-247) is a recurrence on the edge alone.  The
// per-step kernel nevertheless streams (next, cols, eps_per) of ALL nnz edges every iteration (12 B x 49 M at N = 1M)
// to find the ~8.6 of ~49 edges per row that fire, and writes `next` back whole.  Here a SCHEDULE kernel advances the
// recurrence 32 iterations at a time (bit-exact: the same fp32 compare and add per iteration) and emits, per iteration
// and per L2 slice of the embedding, the compacted column lists of the edges that fire; the gradient kernel then reads
// ~8.6 x 4 B per row and iteration instead of ~49 x 12 B.
//
// Layout (one "schedule block" = 64 consecutive rows, one workgroup of the schedule kernel):
//   blk_base (n_blocks + 1) int64   static start of every block's region in `list` (capacity bound from eps_per)
//   list     int32                  column ids; inside a block's region the segments (t, slice, row) follow each other
//                                   in that order, so one gradient pass (fixed t, slice) reads a contiguous run per block
//   hdr      (B * S, n_rows) uint2  one record per (iteration, slice, row): .x = start of the segment in `list`
//                                   (absolute), .y = segment length (low 16 bits) | number of edges of the row that fire
//                                   at this iteration over all slices (high 16 bits): the row draws
//                                   min(5 * act, n_negatives) negatives (umap.py:283-288)
// Gradient pass = one launch per slice s of the embedding (Z slice <= 4 MiB = one XCD's L2): a row group walks ONE item
// stream made of its fired edges with column in slice s followed by its negatives drawn inside slice s (exact
// multinomial split of the row's negative count, tdr_embed_common.h); both kinds share the d^b evaluation and differ
// in a select.  Partial (attraction, repulsion) sums travel between the slice passes; the last pass clamps each to
// [-4, 4] (umap.py:262,290) and writes the gradient.
#include "tdr_embed_common.h"
#include "tdr_umap_pool.h"
#include "../../include/torchdr_amd.h"

namespace tdr {

constexpr int SCHED_RB = 64;    // rows per schedule block
constexpr int SCHED_BMAX = 32;  // iterations per schedule window (one mask bit each)
constexpr int SCHED_DENSE_MIN = 3;  // a register-resident chunk is counted / placed bit-sliced when some lane fires this often

// upper bound of the firings of one edge in ANY window of B iterations: its counter advances by eps_per per firing and
// fires at most once per iteration -> floor(B / eps_per) + 1, plus slack for the fp32 roundings of the additions and of
// this quotient
__device__ __forceinline__ int edge_capacity(float ep, int B) {
    if (!(ep < __builtin_inff())) return 0;
    const float q = (float)B / ep;
    if (!(q < (float)B)) return B;
    const int c = (int)q + 3;
    return c < B ? c : B;
}

__global__ __launch_bounds__(256) void umap_sched_plan_kernel(const int64_t* __restrict__ rowptr, const float* __restrict__ eps_per,
                                                              int64_t n_rows, int B, int rows_per_block, int64_t* __restrict__ cap) {
    __shared__ unsigned long long part[4];
    const int64_t rb = blockIdx.x;
    const int64_t r0 = rb * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n_rows) ? r0 + rows_per_block : n_rows;
    const int64_t e0 = rowptr[r0], e1 = rowptr[r1];
    unsigned long long c = 0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) c += (unsigned long long)edge_capacity(eps_per[e], B);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cap[rb] = (int64_t)(part[0] + part[1] + part[2] + part[3]);
}

// out[0..n] = exclusive scan of in[0..n) (out[n] = total); one workgroup, n is the number of schedule blocks (N / 64)
__global__ __launch_bounds__(256) void scan_i64_kernel(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    __shared__ long long tot[256];
    const int tid = threadIdx.x;
    const int64_t per = (n + 255) / 256;
    const int64_t i0 = tid * per, i1 = (i0 + per < n) ? i0 + per : n;
    long long s = 0;
    for (int64_t i = i0; i < i1; ++i) s += in[i];
    tot[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) { const long long t = tot[i]; tot[i] = run; run += t; }
        out[n] = run;
    }
    __syncthreads();
    long long run = tot[tid];
    for (int64_t i = i0; i < i1; ++i) { const long long t = in[i]; out[i] = run; run += t; }
}

struct SchedBuildParams {
    const int64_t* rowptr;
    const int32_t* cols;
    const float* eps_per;
    float* next;              // epoch_of_next_sample, advanced by B iterations
    int64_t n_rows;
    uint32_t slice_step;      // ceil((N - 1) / S): column j belongs to slice min(S - 1, j / slice_step)
    int t0, B, S;             // window = iterations t0 .. t0 + B - 1 (B <= 32), S slices
    int stash;                // 1: keep the first chunks' (mask, column | slice << 29) in registers (needs N <= 2^29)
    const int* iter_base;     // optional device int added to t0 (graph replays: one captured window serves every window)
    const int64_t* blk_base;
    int32_t* list;
    uint2* hdr;               // (B * S, n_rows) segment records, see the file header
    int* err;                 // device flag: 1 = a block's region would overflow (never with the plan's bound),
                              // 2 = a segment longer than 65535 entries or a list beyond 2^32 entries (unsupported)
};

// The edge's firings in the window [t0, t0 + B) as a bit mask; nx advances exactly as umap.py:243-247 does step by step.
// `nx <= t + 1` first holds at t = ceil(nx) - 1 (or at once when the counter lags behind the iteration), so the loop
// runs once per FIRING, not once per iteration: rarely-firing edges cost one compare.
__device__ __forceinline__ uint32_t fire_mask(float& nx, float ep, int t0, int B) {
    uint32_t m = 0;
    const float tend = (float)(t0 + B);
    int tcur = t0;
    while (nx <= tend) {
        int tf = (int)ceilf(nx) - 1;
        if (tf < tcur) tf = tcur;
        if (tf >= t0 + B) break;
        m |= 1u << (tf - t0);
        nx = __fadd_rn(nx, ep);
        tcur = tf + 1;
    }
    return m;
}

// ---- bit-sliced counting over a 16-lane row group -------------------------------------------------------------------
// A lane's firing mask holds one bit per iteration of the window.  "How many lanes of the row group fire at iteration t"
// (the segment sizes) is a sum of 1-bit values over the 16 lanes -- for all 32 iterations AT ONCE when the sums are kept
// bit-sliced: plane p of a value holds bit p of the count of every iteration.  Adding two bit-sliced numbers is a ripple of
// full adders on 32-bit words (xor / majority), moving a number to another lane one DPP instruction per plane: ~50 vector
// instructions give all 32 totals of a row group, with no LDS traffic, where the per-firing loop runs as many rounds as the
// busiest lane fires (up to 32) and its LDS atomics hit the SAME counter from up to 16 lanes at once.  Used for the
// counting pass of the register-resident chunks (phase 1: 1.008 -> 0.935 ms of a 1.9 ms window).  The placement pass keeps
// the returning atomics: the bit-sliced prefix-count form (position = pointer + number of lanes below that fire into the
// same segment) was built, passed the same tests and measured SLOWER (1.99 vs 1.91 ms per window): the kernel is bound by
// vector instructions (11.6 k per wavefront, 62 % of the SIMD cycles; profiles/r03_sched_build_pmc.json), and extracting a
// 4-plane rank per firing costs more instructions than the atomic it replaces.
template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

#define TDR_FA(x, y, c, sum, carry)      \
    do {                                 \
        const uint32_t xy__ = (x) ^ (y); \
        sum = xy__ ^ (c);                \
        carry = ((x) & (y)) | ((c) & xy__); \
    } while (0)

// totals over the row group of a 1-bit-per-iteration value: every lane of the group gets the 5 planes (counts 0..16)
__device__ __forceinline__ void bs_total16(uint32_t m, uint32_t (&T)[5]) {
    uint32_t b = dppu<0xB1>(m);                       // pairs
    const uint32_t s0 = m ^ b, s1 = m & b;
    uint32_t b0 = dppu<0x4E>(s0), b1 = dppu<0x4E>(s1);  // quads: 2-bit + 2-bit
    const uint32_t r0 = s0 ^ b0, c0 = s0 & b0;
    uint32_t r1, r2;
    TDR_FA(s1, b1, c0, r1, r2);
    b0 = dppu<0x141>(r0); b1 = dppu<0x141>(r1); uint32_t b2 = dppu<0x141>(r2);   // half rows: 3-bit + 3-bit
    const uint32_t u0 = r0 ^ b0, k0 = r0 & b0;
    uint32_t u1, k1, u2, u3;
    TDR_FA(r1, b1, k0, u1, k1);
    TDR_FA(r2, b2, k1, u2, u3);
    b0 = dppu<0x140>(u0); b1 = dppu<0x140>(u1); b2 = dppu<0x140>(u2); const uint32_t b3 = dppu<0x140>(u3);   // row: 4-bit + 4-bit
    uint32_t q1, q2, q3;
    T[0] = u0 ^ b0; q1 = u0 & b0;
    TDR_FA(u1, b1, q1, T[1], q2);
    TDR_FA(u2, b2, q2, T[2], q3);
    TDR_FA(u3, b3, q3, T[3], T[4]);
}
template <int NP>
__device__ __forceinline__ uint32_t bs_get(const uint32_t (&V)[NP], int t) {
    uint32_t c = 0;
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) c |= ((V[p2] >> t) & 1u) << p2;
    return c;
}

// One workgroup = one schedule block (64 rows); a wavefront owns 16 rows and walks them 4 at a time with 16 lanes per
// row.  Phase 1 counts the firings per (t, slice, row) in LDS, the counts are scanned into segment starts, phase 2
// places every firing with a returning LDS atomic on its segment's write pointer: from registers for the first 32
// edges of a row (kept from phase 1), by recomputing the masks for the rest (the 64 rows' edge state is ~25 KB, an L2
// hit).  A row's counters are touched by one wavefront only, in program order, so the
// placement is a function of the input alone (same lists on every run); the order inside a segment is the order the
// LDS unit serialises the lanes of one instruction in, which only permutes the terms of the force sum.
// Rows are laid out by tdr_umap_sched_layout_f32 with their often-firing edges first: the per-lane loops run as many
// times as the busiest lane of the wavefront fires, so homogeneous chunks matter.
#define CNT_STRIDE 65  // odd stride: the 16 lanes of a row group hit different (t, slice) -> different banks
// scratch builds only (tools/sched_build_ablate.py): bit 0 stop after phase 1, bit 1 no counting atomics, bit 2 no row
// records, bit 3 no list stores, bit 4 no phase 2
#ifndef TDR_SCHED_ABLATE
#define TDR_SCHED_ABLATE 0
#endif

template <int STASH>
__global__ __launch_bounds__(256) void umap_sched_build_kernel(const SchedBuildParams P) {
    extern __shared__ uint32_t cnt[];  // [B * S][65]
    __shared__ uint32_t wave_tot[4];
    __shared__ uint16_t actl[SCHED_BMAX * 64];
    const int K = P.B * P.S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gl = lane & 15, gq = lane >> 4;
    const int64_t rb = blockIdx.x;
    const float INF = __builtin_inff();
    const int t0 = P.t0 + (P.iter_base ? *P.iter_base : 0);
    for (int i = tid; i < K * CNT_STRIDE; i += 256) cnt[i] = 0;
    __syncthreads();

    // (mask, column | slice) of the first STASH chunks of every row stay in registers for phase 2: the rows are laid out
    // with their often-firing edges first, so these chunks carry most of the firings -- and most of the cost of
    // advancing the counters, which phase 2 would otherwise repeat
    uint32_t m_st[4 * STASH], cs_st[4 * STASH];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lr = 16 * w + 4 * q + gq;
        const int64_t r = rb * SCHED_RB + lr;
        int64_t e0 = 0, e1 = 0;
        if (r < P.n_rows) { e0 = P.rowptr[r]; e1 = P.rowptr[r + 1]; }
        const int len = (int)(e1 - e0);
        int maxlen = len;
        maxlen = max(maxlen, __shfl_xor(maxlen, 16, 64));
        maxlen = max(maxlen, __shfl_xor(maxlen, 32, 64));
        // segment sizes / write pointers of a chunk whose lanes fire often: bit-sliced totals per slice, lane gl adds the
        // counts of iterations gl and gl + 16 (64 distinct LDS words per instruction: no conflicts)
        auto add_totals = [&](uint32_t m, uint32_t s, int lr_) {
            for (int sg = 0; sg < P.S; ++sg) {
                const uint32_t ms = (s == (uint32_t)sg) ? m : 0u;
                if (__ballot(ms != 0u) == 0ull) continue;
                uint32_t T[5];
                bs_total16(ms, T);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int t = gl + 16 * hh;
                    const uint32_t c = (t < P.B) ? bs_get<5>(T, t) : 0u;
                    if (c && !(TDR_SCHED_ABLATE & 2)) atomicAdd(&cnt[(t * P.S + sg) * CNT_STRIDE + lr_], c);
                }
            }
        };
        auto count_chunk = [&](int c, uint32_t& m_out, uint32_t& cs_out, bool keep) {
            const bool valid = c + gl < len;
            const int64_t e = e0 + c + gl;
            float nx = valid ? P.next[e] : INF;
            const float ep = valid ? P.eps_per[e] : INF;
            const uint32_t col = valid ? (uint32_t)P.cols[e] : 0u;
            uint32_t m = fire_mask(nx, ep, t0, P.B);
            if (keep && m) P.next[e] = nx;  // phase 2 does not visit this edge again
            uint32_t s = col / P.slice_step;
            if (s > (uint32_t)(P.S - 1)) s = (uint32_t)(P.S - 1);
            m_out = m;
            cs_out = col | (s << 29);  // kept form only (stash = 1 requires column ids below 2^29)
            const int sbase = (int)s * CNT_STRIDE + lr;
            if (keep && __any(__popc(m) >= SCHED_DENSE_MIN)) {   // wavefront-uniform; phase 2 takes the same decision
                add_totals(m, s, lr);
                return;
            }
            while (m) {  // four firings per round: the LDS operations of a round are independent
                int tt[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ok[u] = m != 0u; tt[u] = ok[u] ? __ffs(m) - 1 : 0; m &= m - 1u; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && !(TDR_SCHED_ABLATE & 2)) atomicAdd(&cnt[tt[u] * P.S * CNT_STRIDE + sbase], 1u);  // a count: order-independent
            }
        };
#pragma unroll
        for (int ci = 0; ci < STASH; ++ci) {
            m_st[q * STASH + ci] = 0u; cs_st[q * STASH + ci] = 0u;
            if (16 * ci < maxlen) count_chunk(16 * ci, m_st[q * STASH + ci], cs_st[q * STASH + ci], P.stash != 0);
        }
        for (int c = 16 * STASH; c < maxlen; c += 16) {
            uint32_t mm, cc;
            count_chunk(c, mm, cc, false);
        }
    }
    __syncthreads();
    if (TDR_SCHED_ABLATE & 1) return;

    // rows' active counts per iteration (all slices)
    for (int i = tid; i < P.B * 64; i += 256) {
        const int t = i >> 6, lr = i & 63;
        uint32_t a = 0;
        for (int s = 0; s < P.S; ++s) a += cnt[(t * P.S + s) * CNT_STRIDE + lr];
        actl[i] = (uint16_t)(a > 65535u ? 65535u : a);
    }
    // exclusive scan of the counts in (segment k = t * S + slice, row) order; wavefront w owns segments [k0, k1)
    const int kper = (K + 3) / 4;
    const int k0 = w * kper;
    const int k1 = (k0 + kper < K) ? k0 + kper : K;
    uint32_t sum = 0;
    for (int k = k0; k < k1; ++k) sum += cnt[k * CNT_STRIDE + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) wave_tot[w] = sum;
    __syncthreads();
    uint32_t carry = 0;
    for (int i = 0; i < w; ++i) carry += wave_tot[i];
    const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    const int64_t base = P.blk_base[rb];
    const int64_t capacity64 = P.blk_base[rb + 1] - base;
    const uint32_t capacity = capacity64 > 0xffffffffLL ? 0xffffffffu : (uint32_t)capacity64;
    const int64_t row = rb * SCHED_RB + lane;
    bool bad = base + capacity64 > 0xffffffffLL;
    for (int k = k0; k < k1; ++k) {
        const uint32_t v = cnt[k * CNT_STRIDE + lane];
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        const uint32_t ex = carry + inc - v;
        cnt[k * CNT_STRIDE + lane] = ex;  // from here on: the write pointer of segment (k, row)
        bad = bad || v > 65535u;
        if (row < P.n_rows && !(TDR_SCHED_ABLATE & 4))
            P.hdr[(size_t)k * P.n_rows + row] = make_uint2((uint32_t)base + ex, (v & 0xffffu) | ((uint32_t)actl[(k / P.S) * 64 + lane] << 16));
        carry += __shfl(inc, 63, 64);
    }
    if (bad) atomicMax(P.err, 2);
    if (total > capacity && tid == 0) atomicMax(P.err, 1);
    __syncthreads();
    if (TDR_SCHED_ABLATE & 16) return;

#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lr = 16 * w + 4 * q + gq;
        const int64_t r = rb * SCHED_RB + lr;
        int64_t e0 = 0, e1 = 0;
        if (r < P.n_rows) { e0 = P.rowptr[r]; e1 = P.rowptr[r + 1]; }
        const int len = (int)(e1 - e0);
        int maxlen = len;
        maxlen = max(maxlen, __shfl_xor(maxlen, 16, 64));
        maxlen = max(maxlen, __shfl_xor(maxlen, 32, 64));
        auto place = [&](uint32_t m, uint32_t col, uint32_t s) {
            const int sbase = (int)s * CNT_STRIDE + lr;
            while (m) {  // four firings per round: four returning LDS atomics in flight, then four stores
                int tt[4];
                bool ok[4];
                uint32_t pos[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ok[u] = m != 0u; tt[u] = ok[u] ? __ffs(m) - 1 : 0; m &= m - 1u; }
#pragma unroll
                for (int u = 0; u < 4; ++u) pos[u] = ok[u] ? atomicAdd(&cnt[tt[u] * P.S * CNT_STRIDE + sbase], 1u) : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && pos[u] < capacity && !(TDR_SCHED_ABLATE & 8)) P.list[base + pos[u]] = (int32_t)col;
            }
        };
        int c_first = 0;
        if (P.stash) {
#pragma unroll
            for (int ci = 0; ci < STASH; ++ci) place(m_st[q * STASH + ci], cs_st[q * STASH + ci] & 0x1fffffffu, cs_st[q * STASH + ci] >> 29);
            c_first = 16 * STASH;
        }
        for (int c = c_first; c < maxlen; c += 16) {
            const bool valid = c + gl < len;
            const int64_t e = e0 + c + gl;
            float nx = valid ? P.next[e] : INF;
            const float ep = valid ? P.eps_per[e] : INF;
            const uint32_t col = valid ? (uint32_t)P.cols[e] : 0u;
            const uint32_t m = fire_mask(nx, ep, t0, P.B);
            if (m) P.next[e] = nx;
            uint32_t s = col / P.slice_step;
            if (s > (uint32_t)(P.S - 1)) s = (uint32_t)(P.S - 1);
            place(m, col, s);
        }
    }
}

// Loop layout of a row's edges: ascending eps_per (= often-firing edges first, never-firing ones last), ties by column
// id, then position -- the order does not depend on how the CSR row was arranged, so a row-sharded fit (rows symmetrised
// in the loop's numbering) and a single-process one (rows permuted into it) sum a row's forces in the same order.  Rank sort per row, one wavefront per row: the row's periods sit in registers (lane p holds entries p,
// p + 64, ...) and every entry is broadcast once through the scalar unit (v_readlane); rows of more than 2048 edges
// keep their order.
// Rows of 65 .. 64 K edges: K entries per lane in registers (lane p holds entries p, p + 64, ...), every entry broadcast ONCE and
// compared with all K of the lane -- len x (2 broadcasts + K compares) instead of (len / 64)^2 passes of 64 broadcasts that re-read
// the row from memory each time.  On the headline's graph 16 % of the rows hold more than 64 edges (4 % more than 128, the longest
// 829) and took 70 % of the kernel: 2.5 -> 1.1 ms (tools/layout_perf.py real).  Keys as in the one-entry form: (period bits, column)
// as one 64-bit integer, ties by position; the caller has checked that periods are positive or +inf and columns non-negative.
template <int K>
__device__ __forceinline__ void layout_row_regs(const int64_t b, const int len, const int lane, const int32_t* __restrict__ cols,
                                                const float* __restrict__ eps_per, int32_t* __restrict__ cols_out,
                                                float* __restrict__ eps_out) {
    uint32_t mb[K];
    int32_t col[K];
    unsigned long long key[K];
    int rank[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = lane + 64 * k;
        const bool have = p < len;
        mb[k] = have ? __float_as_uint(eps_per[b + p]) : 0x7f800000u;
        col[k] = have ? cols[b + p] : 0x7fffffff;
        key[k] = ((unsigned long long)mb[k] << 32) | (uint32_t)col[k];
        rank[k] = 0;
    }
#pragma unroll
    for (int kq = 0; kq < K; ++kq) {
        const int nq = len - 64 * kq < 64 ? len - 64 * kq : 64;      // wavefront-uniform
        for (int q = 0; q < nq; ++q) {
            const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mb[kq], q) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane(col[kq], q);
#pragma unroll
            for (int k = 0; k < K; ++k)
                rank[k] += (ok < key[k] || (ok == key[k] && (kq < k || (kq == k && q < lane)))) ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < len) { cols_out[b + rank[k]] = col[k]; eps_out[b + rank[k]] = __uint_as_float(mb[k]); }
}

__global__ __launch_bounds__(256) void umap_sched_layout_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                                const float* __restrict__ eps_per, int64_t n_rows,
                                                                int32_t* __restrict__ cols_out, float* __restrict__ eps_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    const int len = (int)(e - b);
    if (len > 2048) {
        for (int p = lane; p < len; p += 64) { cols_out[b + p] = cols[b + p]; eps_out[b + p] = eps_per[b + p]; }
        return;
    }
    if (len <= 64) {  // the common case: one entry per lane
        const bool have = lane < len;
        const float mine = have ? eps_per[b + lane] : __builtin_inff();
        const int32_t col = have ? cols[b + lane] : 0;
        int rank = 0;
        // periods are positive (or +inf) and columns non-negative: (period bits, column) as ONE 64-bit key orders like the pair, and
        // a rank step is two broadcasts, one 64-bit compare and an add instead of five compares joined through scalar masks (the
        // kernel was 2.9 ms of a 117 ms fit at N = 1M).  Anything else (a negative or NaN period) takes the general comparison.
        const uint32_t mbits = __float_as_uint(mine);
        if (__builtin_amdgcn_ballot_w64(have && ((mbits >> 31) != 0u || mine != mine || col < 0)) == 0ull) {
            const unsigned long long key = ((unsigned long long)mbits << 32) | (uint32_t)col;
            for (int q = 0; q < len; ++q) {
                const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mbits, q) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane(col, q);
                rank += (ok < key || (ok == key && q < lane)) ? 1 : 0;
            }
        } else {
            for (int q = 0; q < len; ++q) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), q));
                const int32_t oc = __builtin_amdgcn_readlane(col, q);
                rank += (o < mine || (o == mine && (oc < col || (oc == col && q < lane)))) ? 1 : 0;
            }
        }
        if (have) { cols_out[b + rank] = col; eps_out[b + rank] = mine; }
        return;
    }
    if (len <= 1024) {
        // register-resident form when every period is positive (or +inf) and every column non-negative: the 64-bit key orders like the pair
        bool odd = false;
        for (int p = lane; p < len; p += 64) {
            const float v = eps_per[b + p];
            odd = odd || (__float_as_uint(v) >> 31) != 0u || v != v || cols[b + p] < 0;
        }
        if (__builtin_amdgcn_ballot_w64(odd) == 0ull) {
            if (len <= 128) layout_row_regs<2>(b, len, lane, cols, eps_per, cols_out, eps_out);
            else if (len <= 256) layout_row_regs<4>(b, len, lane, cols, eps_per, cols_out, eps_out);
            else if (len <= 512) layout_row_regs<8>(b, len, lane, cols, eps_per, cols_out, eps_out);
            else layout_row_regs<16>(b, len, lane, cols, eps_per, cols_out, eps_out);
            return;
        }
    }
    for (int p0 = 0; p0 < len; p0 += 64) {
        const int p = p0 + lane;
        const bool have = p < len;
        const float mine = have ? eps_per[b + p] : 0.f;
        const int32_t col = have ? cols[b + p] : 0;
        int rank = 0;
        for (int q0 = 0; q0 < len; q0 += 64) {
            const float theirs = (q0 + lane < len) ? eps_per[b + q0 + lane] : __builtin_inff();
            const int32_t tcol = (q0 + lane < len) ? cols[b + q0 + lane] : 0;
            const int nq = (len - q0 < 64) ? len - q0 : 64;
            for (int q = 0; q < nq; ++q) {
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, theirs), q));
                const int32_t oc = __builtin_amdgcn_readlane(tcol, q);
                rank += (o < mine || (o == mine && (oc < col || (oc == col && q0 + q < p)))) ? 1 : 0;
            }
        }
        if (have) { cols_out[b + rank] = col; eps_out[b + rank] = mine; }
    }
}

// ---- the schedule build on GROUP-ORDERED loop state (round 4) ----------------------------------------------------------
// What bounds the row-chunk kernel above (profiles/r03_sched_build_pmc.json, r03_sched_build_ablation.json): its per-firing
// loops run as many rounds as the busiest lane of the wavefront fires (a row's 16 hottest edges sit in one 16-lane group:
// ~40 % of the lanes do work), and its list stores are 275 M scattered 4-byte requests per window (0.72 of the 1.9 ms).
// Here the loop state of every GROUP of 16 consecutive rows is laid out once per fit in order of the firing period
// (tdr_umap_sched_group_f32: a stable counting sort of the group's edges by period class, 4 classes per octave, rows
// pre-sorted by (period, column) -- so a row's edges keep their (period, column) order among themselves whatever the
// other rows of the group are), one wavefront owns one group, and
//   * a lane's edge and its 63 neighbours fire (nearly) equally often: the recurrence / counting / placement loops run
//     with most lanes busy;
//   * the group's list region is filled through an LDS stage, 8 iterations (one contiguous run of the region: segments
//     follow each other in (iteration, slice, row) order) at a time, and flushed with coalesced stores; entries beyond
//     the stage go straight to memory;
//   * nothing is shared between wavefronts: no workgroup barrier, the scan of the segment sizes is a wavefront scan.
// A row's counters are touched by one wavefront only, in program order, lanes of one LDS instruction in lane order, so a
// segment lists the row's firing edges in (period, column) order on every run and for every composition of the group
// (a row-sharded fit cuts the groups elsewhere and still sums a row's forces in the same order).
// Records and lists have the format of the row-chunk kernel with 16-row blocks (grp_base instead of blk_base).
constexpr int G2_ROWS = 16;    // rows per group (one wavefront)
constexpr int G2_STRIDE = 17;  // counters of one (iteration, slice): 16 rows, odd stride
constexpr int G2_STASH = 16;   // 64-edge chunks whose (mask, column, row | slice) stay in registers between the phases
constexpr int G2_TC = 4;       // iterations per staged run (part of the order key inside a segment: one value)
constexpr int G2_STAGE_DEFAULT = 768;   // staged list entries (LDS, >= 512); the mean run is 16 rows x 8.6 firings x 4 iterations = 550

// firing period -> class: 4 per octave (epochs_per_sample = max weight / weight >= 1); 2^15.5 and beyond, incl. the
// never-firing edges (inf), share class 63.  Monotone in the period.
__device__ __forceinline__ int period_class(float ep) {
    const uint32_t b = __float_as_uint(ep);
    if (b < 0x3F800000u) return 0;
    const uint32_t k = (b - 0x3F800000u) >> 21;
    return k > 63u ? 63 : (int)k;
}

// One wavefront per group: stable counting sort of the group's edges (row-major, every row already in (period, column)
// order) by period class.  Outputs in group order: column, period, (local row | slice << 4), and the edge's position in
// the row-major order relative to the group's first edge (to move per-edge state between the two orders).
__global__ __launch_bounds__(64) void umap_sched_group_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                              const float* __restrict__ eps_per, int64_t n_rows, uint32_t slice_step,
                                                              int S, int32_t* __restrict__ cols_g, float* __restrict__ eps_g,
                                                              uint8_t* __restrict__ rs_g, int32_t* __restrict__ order_g,
                                                              int* __restrict__ err) {
    __shared__ int hist[64];
    __shared__ int rpl[17];
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * G2_ROWS;
    const int nr = (int)((n_rows - r0 < G2_ROWS) ? n_rows - r0 : G2_ROWS);
    const int64_t e0 = rowptr[r0], e1 = rowptr[r0 + nr];
    if (e1 - e0 > 0x7fffffffLL - 64) {
        if (lane == 0) atomicMax(err, 2);
        return;
    }
    const int ne = (int)(e1 - e0);
    if (lane <= 16) rpl[lane] = (int)(rowptr[r0 + (lane < nr ? lane : nr)] - e0);
    hist[lane] = 0;
    __syncthreads();
    for (int c = 0; c < ne; c += 64)
        if (c + lane < ne) atomicAdd(&hist[period_class(eps_per[e0 + c + lane])], 1);  // a count: order-independent
    __syncthreads();
    // lane b keeps the write position of class b
    const int cntb = hist[lane];
    int incl = cntb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    int binpos = incl - cntb;
    for (int c = 0; c < ne; c += 64) {
        const int i = c + lane;
        const bool valid = i < ne;
        const float ep = valid ? eps_per[e0 + i] : 0.f;
        const int32_t col = valid ? cols[e0 + i] : 0;
        const int cls = valid ? period_class(ep) : -1;
        unsigned long long rem = __ballot(valid);
        int dst = 0;
        while (rem) {  // one round per distinct class of the chunk; lanes of a class keep their order (stable)
            const int l0 = __ffsll((long long)rem) - 1;
            const int c0 = __builtin_amdgcn_readlane(cls, l0);
            const unsigned long long mk = __ballot(cls == c0);
            const int base = __builtin_amdgcn_readlane(binpos, c0);
            if (cls == c0) dst = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (lane == c0) binpos += __popcll(mk);
            rem &= ~mk;
        }
        if (valid) {
            int lr = 0;
#pragma unroll
            for (int q = 1; q <= 16; ++q) lr += (rpl[q] <= i) ? 1 : 0;
            uint32_t s = (uint32_t)col / slice_step;
            if (s > (uint32_t)(S - 1)) s = (uint32_t)(S - 1);
            cols_g[e0 + dst] = col;
            eps_g[e0 + dst] = ep;
            rs_g[e0 + dst] = (uint8_t)(lr | (int)(s << 4));
            order_g[e0 + dst] = i;
        }
    }
}

// per-edge values from group order back to the row-major order (inspection / parity: epoch_of_next_sample)
__global__ __launch_bounds__(64) void umap_sched_ungroup_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ order_g,
                                                                const float* __restrict__ vals_g, int64_t n_rows,
                                                                float* __restrict__ vals_rm) {
    const int64_t r0 = (int64_t)blockIdx.x * G2_ROWS;
    const int64_t r1 = (r0 + G2_ROWS < n_rows) ? r0 + G2_ROWS : n_rows;
    const int64_t e0 = rowptr[r0], e1 = rowptr[r1];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 64) vals_rm[e0 + order_g[e]] = vals_g[e];
}

struct SchedBuild2Params {
    const int64_t* rowptr;
    const int32_t* cols;      // group order (tdr_umap_sched_group_f32)
    const float* eps_per;
    const uint8_t* rs;        // local row | slice << 4
    float* next;              // epoch_of_next_sample in group order, advanced by B iterations
    int64_t n_rows;
    int t0, B, S;
    int stage;                // LDS stage entries
    const int* iter_base;
    const int64_t* grp_base;  // (n_groups + 1) static list regions (tdr_umap_sched_plan_groups_f32)
    int32_t* list;
    uint2* hdr;
    int* err;
};

template <int WAVES, int TC>   // WAVES: wavefronts per SIMD the register allocation aims at; TC: iterations per staged run
__global__ __launch_bounds__(64, WAVES) void umap_sched_build2_kernel(const SchedBuild2Params P) {
    extern __shared__ uint32_t sm2[];  // cnt [B * S][17] | stage [P.stage >= 512] (the rows' active counts until phase 2)
    __shared__ uint32_t tb[SCHED_BMAX / TC + 2];
    const int K = P.B * P.S, KS = P.S * G2_STRIDE;
    uint32_t* cnt = sm2;
    uint32_t* stage = sm2 + ((K * G2_STRIDE + 3) & ~3);
    uint32_t* actl = stage;
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    const int64_t r0 = g * G2_ROWS;
    const int nr = (int)((P.n_rows - r0 < G2_ROWS) ? P.n_rows - r0 : G2_ROWS);
    const int64_t e0 = P.rowptr[r0], e1 = P.rowptr[r0 + nr];
    const float INF = __builtin_inff();
    const int t0 = P.t0 + (P.iter_base ? *P.iter_base : 0);
    if (e1 - e0 > 0x7fffffffLL - 64) {
        if (lane == 0) atomicMax(P.err, 2);
        return;
    }
    const int ne = (int)(e1 - e0);
    for (int i = lane; i < K * G2_STRIDE; i += 64) cnt[i] = 0;
    __syncthreads();

    // fire_mask() with the count of every firing added to its (iteration, slice, row) counter as it is found (a count:
    // order-independent); idx = slice * 17 + local row
    auto fire_count = [&](float& nx, float ep, int idx) {
        uint32_t m = 0;
        const float tend = (float)(t0 + P.B);
        int tcur = t0;
        while (nx <= tend) {
            int tf = (int)ceilf(nx) - 1;
            if (tf < tcur) tf = tcur;
            if (tf >= t0 + P.B) break;
            m |= 1u << (tf - t0);
            atomicAdd(&cnt[(tf - t0) * KS + idx], 1u);
            nx = __fadd_rn(nx, ep);
            tcur = tf + 1;
        }
        return m;
    };
    // phase 1: advance the counters, count the firings per (iteration, slice, row).  ALL loads of the register-resident
    // chunks are issued before the first counter is written back: a store to `next` orders every later load of `next`
    // behind it, and chunk-by-chunk the wavefront would sit out one memory latency per chunk (12 per group).
    uint32_t m_st[G2_STASH], c_st[G2_STASH], rs_st[G2_STASH / 4];
    {
        float nxv[G2_STASH], epv[G2_STASH];
        uint32_t rbv[G2_STASH];
#pragma unroll
        for (int ci = 0; ci < G2_STASH; ++ci) {
            const bool valid = ci * 64 + lane < ne;
            const int64_t e = e0 + ci * 64 + lane;
            nxv[ci] = valid ? P.next[e] : INF;
            epv[ci] = valid ? P.eps_per[e] : INF;
            rbv[ci] = valid ? (uint32_t)P.rs[e] : 0u;
            c_st[ci] = valid ? (uint32_t)P.cols[e] : 0u;
        }
#pragma unroll
        for (int q = 0; q < G2_STASH / 4; ++q)
            rs_st[q] = rbv[4 * q] | (rbv[4 * q + 1] << 8) | (rbv[4 * q + 2] << 16) | (rbv[4 * q + 3] << 24);
#pragma unroll
        for (int ci = 0; ci < G2_STASH; ++ci) {
            m_st[ci] = 0u;
            if (ci * 64 < ne) {
                float nx = nxv[ci];
                const uint32_t m = fire_count(nx, epv[ci], (int)(rbv[ci] >> 4) * G2_STRIDE + (int)(rbv[ci] & 15u));
                if (m) P.next[e0 + ci * 64 + lane] = nx;
                m_st[ci] = m;
            }
        }
    }
    for (int c = G2_STASH * 64; c < ne; c += 64) {  // groups of more than 1024 edges: the tail is advanced again in phase 3
        const bool valid = c + lane < ne;
        const int64_t e = e0 + c + lane;
        float nx = valid ? P.next[e] : INF;
        const float ep = valid ? P.eps_per[e] : INF;
        const uint32_t rb = valid ? (uint32_t)P.rs[e] : 0u;
        (void)fire_count(nx, ep, (int)(rb >> 4) * G2_STRIDE + (int)(rb & 15u));
    }
    __syncthreads();

    // rows' active counts per iteration (all slices)
    for (int i = lane; i < P.B * G2_ROWS; i += 64) {
        const int t = i >> 4, lr = i & 15;
        uint32_t a = 0;
        for (int s = 0; s < P.S; ++s) a += cnt[(t * P.S + s) * G2_STRIDE + lr];
        actl[i] = a > 65535u ? 65535u : a;
    }
    __syncthreads();
    // exclusive scan of the counts in (segment k = t * S + slice, row) order, four segments per step; records
    const int64_t gbase = P.grp_base[g];
    const int64_t capacity64 = P.grp_base[g + 1] - gbase;
    const uint32_t capacity = capacity64 > 0xffffffffLL ? 0xffffffffu : (uint32_t)capacity64;
    bool bad = gbase + capacity64 > 0xffffffffLL;
    const int n_runs = (P.B + TC - 1) / TC;
    uint32_t carry = 0;
    for (int k4 = 0; k4 < K; k4 += 4) {
        const int k = k4 + (lane >> 4), lr = lane & 15;
        const uint32_t v = k < K ? cnt[k * G2_STRIDE + lr] : 0u;
        // inclusive scan inside the 16-lane DPP row (= one segment index), rows chained through the scalar unit
        uint32_t inc = v;
        inc += dppu<0x111>(inc);
        inc += dppu<0x112>(inc);
        inc += dppu<0x114>(inc);
        inc += dppu<0x118>(inc);
        const uint32_t t0r = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15), t1r = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
        const uint32_t t2r = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47), t3r = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        const int qrow = lane >> 4;
        const uint32_t below = qrow == 0 ? 0u : (qrow == 1 ? t0r : (qrow == 2 ? t0r + t1r : t0r + t1r + t2r));
        const uint32_t ex = carry + below + inc - v;
        bad = bad || v > 65535u;
        if (k < K) {
            cnt[k * G2_STRIDE + lr] = ex;  // from here on: the write pointer of segment (k, row)
            // records and lists are written once and read by the gradient launches of the window: streamed past the L2
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            if (lr < nr) __builtin_nontemporal_store(u32x2_t{(uint32_t)gbase + ex, (v & 0xffffu) | (actl[(k / P.S) * G2_ROWS + lr] << 16)},
                                                     reinterpret_cast<u32x2_t*>(P.hdr + (size_t)k * P.n_rows + r0 + lr));
            if (lr == 0 && k % (TC * P.S) == 0) tb[k / (TC * P.S)] = ex;
        }
        carry += t0r + t1r + t2r + t3r;
    }
    if (lane == 0) tb[n_runs] = carry;
    if (bad) atomicMax(P.err, 2);
    if (carry > capacity && lane == 0) atomicMax(P.err, 1);
    __syncthreads();

    // phase 2: place the firings, one run of TC iterations at a time, through the LDS stage.  Inside a run the firings
    // are taken RANK-OUTER: pass p places, for every lane, the p-th firing of its edge in this run.  A segment then lists
    // a row's edges by (rank of the firing in the edge's own mask, period, column) -- a key made of the edge alone, so the
    // order does not depend on which chunk or lane the edge sits in (i.e. on the other rows of the group).
    for (int j = 0; j < n_runs; ++j) {
        const uint32_t sbase = tb[j], send = tb[j + 1];
        const int tlo = j * TC;
        const uint32_t runmask = ((1u << TC) - 1u) << tlo;
        auto put = [&](uint32_t pos, uint32_t col) {
            const uint32_t rel = pos - sbase;
            if (rel < (uint32_t)P.stage) stage[rel] = col;
            else if (pos < capacity) P.list[gbase + pos] = (int32_t)col;
        };
        uint32_t live = 0xfu;   // wavefront-uniform: chunk groups that may still hold firings of this run
        for (int pass = 0; pass < TC; ++pass) {
            bool any = false;
            // every atomic of the pass is issued before the first position is used (LDS returns in order)
            uint32_t pos[G2_STASH];
            uint32_t has = 0u;   // wavefront-uniform: chunk groups that placed something
#pragma unroll
            for (int cg = 0; cg < G2_STASH; cg += 4) {
                if (cg * 64 >= ne || !(live & (1u << (cg >> 2)))) continue;
                uint32_t bits[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) bits[u] = m_st[cg + u] & runmask;
                if (__ballot((bits[0] | bits[1] | bits[2] | bits[3]) != 0u) == 0ull) { live &= ~(1u << (cg >> 2)); continue; }
                has |= 1u << (cg >> 2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pos[cg + u] = 0xffffffffu;
                    if (bits[u]) {
                        const uint32_t rb = (rs_st[(cg + u) >> 2] >> (8 * ((cg + u) & 3))) & 0xffu;
                        const int t = __ffs(bits[u]) - 1;
                        m_st[cg + u] &= m_st[cg + u] - 1u;  // earlier runs are used up: the lowest set bit is this one
                        pos[cg + u] = atomicAdd(&cnt[t * KS + (int)(rb >> 4) * G2_STRIDE + (int)(rb & 15u)], 1u);
                    }
                }
            }
            any = has != 0u;
#pragma unroll
            for (int cg = 0; cg < G2_STASH; cg += 4) {
                if (!(has & (1u << (cg >> 2)))) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (pos[cg + u] != 0xffffffffu) put(pos[cg + u], c_st[cg + u]);
            }
            for (int c = G2_STASH * 64; c < ne; c += 64) {  // beyond the register-resident chunks: the mask is formed again
                const bool valid = c + lane < ne;
                const int64_t e = e0 + c + lane;
                float nx = valid ? P.next[e] : INF;
                const float ep = valid ? P.eps_per[e] : INF;
                uint32_t m = fire_mask(nx, ep, t0, P.B) & runmask;
                for (int q = 0; q < pass; ++q) m &= m - 1u;
                if (__ballot(m != 0u) == 0ull) continue;
                any = true;
                if (m) {
                    const uint32_t rb = (uint32_t)P.rs[e];
                    const int t = __ffs(m) - 1;
                    put(atomicAdd(&cnt[t * KS + (int)(rb >> 4) * G2_STRIDE + (int)(rb & 15u)], 1u), (uint32_t)P.cols[e]);
                }
            }
            if (!any) break;
        }
        __syncthreads();
        uint32_t n_st = send - sbase;
        if (n_st > (uint32_t)P.stage) n_st = (uint32_t)P.stage;
        if (sbase >= capacity) n_st = 0;
        else if (n_st > capacity - sbase) n_st = capacity - sbase;
        for (uint32_t i = lane; i < n_st; i += 64) __builtin_nontemporal_store((int32_t)stage[i], P.list + gbase + sbase + i);
        __syncthreads();
    }
    // the counters of the chunks beyond the register-resident ones (phase 1 left them as they were)
    for (int c = G2_STASH * 64; c < ne; c += 64) {
        if (c + lane < ne) {
            const int64_t e = e0 + c + lane;
            float nx = P.next[e];
            if (fire_mask(nx, P.eps_per[e], t0, P.B)) P.next[e] = nx;
        }
    }
}

struct SchedGradParams {
    const float* Z;
    int64_t n_total, row0, n_rows;
    const int32_t* list;
    const uint2* hdr;
    int t_local, S, slice;
    float a, b;
    int neg_rate, n_negatives;
    const int64_t* neg_inj;   // optional (n_rows, n_negatives) injected negatives (parity tests), else the counter hash
    uint64_t seed;
    uint32_t iter;
    const int* iter_base;     // optional device int added to iter (graph replays)
    float exag, rep, eps;
    float* grad;              // (n_rows, NC)
    float* acc;               // (n_rows, 2 NC) partial sums between the slice passes (S > 1)
    // per-pass constants of the negative split / sampler, prepared on the host (sched_pass_constants)
    uint32_t r_lo, r_len, step;     // this slice of the reduced index range [0, N-1): start, length; ceil((N-1)/S)
    int n_levels;                   // log2(S)
    uint32_t lvl_xor[3];            // hash key modifier of the binomial split at each level (split_key)
    int lvl_upper[3];               // 1: this slice lies in the upper half at that level
    // joint launch: ALL slices in one grid, workgroup b on XCD b % 8 (round-robin dispatch) takes slice (b % 8) / (8 / S),
    // so every XCD's L2 still holds one slice; partial sums go to plane `slice` of acc and a combine kernel finishes
    int joint;
    int nc;                         // actual row width of Z / grad / acc (PAD instances: NC is the padded register width)
    uint32_t j_r_lo[8], j_r_len[8], j_lvl_xor[8][3];
    int j_lvl_upper[8][3];
};

typedef float sched_f32x4 __attribute__((ext_vector_type(4)));
// this slice's share of the row's n_use negatives: binomial halving level by level (= slice_count(),
// tdr_embed_common.h) with the hash words spread over the G lanes of the row group and a DPP reduction.  Every level
// flips FAIR coins, which is the exact multinomial split when the S slices of the reduced index range [0, N - 1) are
// equally long; the last slice is up to S - 1 indices shorter (step = ceil((N - 1) / S)), so its points are drawn with
// probability 1 / (S len_last) instead of 1 / (N - 1): a relative bias of at most S / N (8e-6 at the smallest N that has
// two slices), far below the sampling noise of 150 draws per row.  S > 1 only when Z exceeds one L2 (N >= 512 k rows of
// two floats), so an empty last slice (N - 1 < S) cannot occur there; it is guarded (nneg = 0) for direct callers.
template <int G>
__device__ __forceinline__ int pass_negative_count(int n_levels, const uint32_t (&lvl_xor)[3], const int (&lvl_upper)[3], uint32_t rkey,
                                                   int n_use, int gl) {
    int mine = n_use;
    for (int l = 0; l < n_levels; ++l) {
        const uint32_t key = rkey ^ lvl_xor[l];
        int m = 0;
        for (int t = gl; t * 32 < mine; t += G) {
            uint32_t w = mix32(key + 0x7F4A7C15u * (uint32_t)(t + 1));
            const int rem = mine - t * 32;
            if (rem < 32) w &= (1u << rem) - 1u;
            m += __popc(w);
        }
        m = group_sum_dpp<G>(m);
        mine = lvl_upper[l] ? mine - m : m;
    }
    return mine;
}

// One row group (G lanes) walks ONE item stream: the row's fired edges with column in this slice, then its negatives
// drawn inside this slice; lane gl owns the four consecutive items 4 gl .. 4 gl + 3 of every round of 4 G items.
// What bounds it (N = 1M, 2 slices, 26 items per row and slice; DESIGN.md section 3, profiles/r02_umap_sched_ablation.json):
// the vector ALUs and the L2 at once.  691 vector instructions per wavefront and slice (a wave64 instruction holds a
// 16-lane SIMD for 4 cycles, 16 for 32-bit integer multiplies and transcendentals) keep the ALUs ~80 % busy; the 26 M
// random 8-byte gathers per slice keep the L2 ~85 % busy (its request rate, 128 channels x clock, is the ceiling: 268 G/s
// in tools/gather_bench.hip at 2.1 GHz, ~180 G/s at the 1.43 GHz the chip holds under this kernel).  Removing either
// alone (ablations) leaves the other: 0.310 -> 0.266 (no random gathers) / 0.301 (no hash, pow, rcp) / 0.223 ms (both).
// Hence the shape: one 8-byte record per row instead of four header words, one 16-byte list read per lane instead of
// four, a branch-free body (INJ = false: every lane evaluates both address forms and selects), a two-multiply item
// hash, and all slices in ONE launch spread over the XCDs (see the joint fields of SchedGradParams).
// PAD instances serve any n_components <= NC: rows are P.nc floats wide, registers hold NC (zeros beyond P.nc contribute
// nothing to distances or forces)
// scratch builds only (tools/grad_ablate.sh): bit 0 row key without the hash rounds, bit 1 no binomial split (half of the
// negatives per slice), bit 2 no pow, bit 3 one-multiply item hash, bit 4 rows not dealt by load
#ifndef TDR_GRAD_ABLATE
#define TDR_GRAD_ABLATE 0
#endif
template <int NC, int G, bool INJ, bool PAD = false>
__global__ __launch_bounds__(256) void umap_sched_grad_kernel(const SchedGradParams P) {
    constexpr int U = 4;
    const int nc = PAD ? P.nc : NC;
    auto load_row = [&](int64_t i) {
        if (!PAD) return load_z<NC>(P.Z, i);
        Vec<NC> v;
#pragma unroll
        for (int c = 0; c < NC; ++c) v.v[c] = c < nc ? P.Z[(size_t)i * nc + c] : 0.f;
        return v;
    };
    const int gl = threadIdx.x % G;
    int slice = P.slice;
    int64_t blk = blockIdx.x;
    uint32_t r_lo = P.r_lo, r_len = P.r_len, lvl_xor[3] = {P.lvl_xor[0], P.lvl_xor[1], P.lvl_xor[2]};
    int lvl_upper[3] = {P.lvl_upper[0], P.lvl_upper[1], P.lvl_upper[2]};
    if (P.joint) {
        const int xcd = blockIdx.x & 7, per = 8 / P.S;
        slice = xcd / per;
        blk = (int64_t)(blockIdx.x >> 3) * per + (xcd % per);
        r_lo = P.j_r_lo[slice]; r_len = P.j_r_len[slice];
#pragma unroll
        for (int l = 0; l < 3; ++l) { lvl_xor[l] = P.j_lvl_xor[slice][l]; lvl_upper[l] = P.j_lvl_upper[slice][l]; }
    }
    int64_t r = (blk * 256 + threadIdx.x) / G;
    uint2 h;
    if (r < P.n_rows) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t hv = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(P.hdr + (size_t)(P.t_local * P.S + slice) * P.n_rows + r));
        h = make_uint2(hv.x, hv.y);
    }
    if (r >= P.n_rows) return;   // rows beyond the chunk
    const uint32_t gi = (uint32_t)(P.row0 + r);
    const Vec<NC> zi = load_row(gi);
    const int32_t* lst = P.list + h.x;
    const int npos = (int)(h.y & 0xffffu);
    int n_use = (int)(h.y >> 16) * P.neg_rate;
    if (n_use > P.n_negatives) n_use = P.n_negatives;
    const uint32_t rkey = (TDR_GRAD_ABLATE & 1) ? (gi * 0x9E3779B9u) ^ (P.iter * 0x85EBCA6Bu)
                                                : neg_row_key(P.seed, P.iter + (P.iter_base ? (uint32_t)*P.iter_base : 0u), (int64_t)gi);
    // injected negatives: every column is visited and the ones outside this slice are masked
    int nneg = INJ ? n_use : ((TDR_GRAD_ABLATE & 2) ? n_use / P.S : pass_negative_count<G>(P.n_levels, lvl_xor, lvl_upper, rkey, n_use, gl));
    if (!INJ && r_len == 0u) nneg = 0;
    const uint32_t ckey = rkey + 0x632BE5ABu * (uint32_t)(slice + 1) - (uint32_t)npos * 0x9E3779B9u;
    const int total = npos + nneg;
    const float two_ab = 2.0f * P.a * P.b, m2b = -2.0f * P.b;
    float ga[NC], gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = 0.f; gr[c] = 0.f; }
    for (int base = 0; base < total; base += U * G) {
        const int i0 = base + gl * U;
        // the lane's four list entries in one read (the list carries 64 entries of slack behind its last segment, so
        // reading past a segment is harmless; positions >= npos are not used)
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 l4 = {0, 0, 0, 0};
        if (__ballot(i0 < npos)) {  // wavefront-uniform: rounds made of negatives only skip the read
            const int32_t* lp = lst + (i0 < npos ? i0 : 0);
            __builtin_memcpy(&l4, lp, 16);     // (not streamed: the next round of a long row reads on in the same lines -- measured)
        }
        uint32_t jn[U];
        bool v[U], isp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            v[u] = i < total;
            isp[u] = i < npos;
            const uint32_t jl = (uint32_t)l4[u];
            uint32_t jneg;
            if (INJ) {
                jneg = gi;
                if (v[u] && !isp[u]) {
                    const uint32_t j = (uint32_t)P.neg_inj[(size_t)r * P.n_negatives + (i - npos)];
                    uint32_t sl = j / P.step;
                    if (sl > (uint32_t)(P.S - 1)) sl = (uint32_t)(P.S - 1);
                    v[u] = sl == (uint32_t)slice;
                    jneg = j;
                }
            } else {
                const uint32_t x = (TDR_GRAD_ABLATE & 8) ? (ckey + (uint32_t)i * 0x9E3779B9u) * 0x7feb352du
                                                         : mix32_item(ckey + (uint32_t)i * 0x9E3779B9u);  // column i - npos of this slice
                const uint32_t rr = r_lo + __umulhi(x, r_len);
                jneg = rr + (rr >= gi ? 1u : 0u);
            }
            jn[u] = v[u] ? (isp[u] ? jl : jneg) : gi;
        }
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zj[u] = load_row((int64_t)jn[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            const float pb = (TDR_GRAD_ABLATE & 4) ? d : fast_pow(d, P.b);  // d = 0: exp2(b * log2 0) = exp2(-inf) = 0, no branch needed
            const float den = 1.0f + P.a * pb;
            // attraction 2ab d^(b-1) / (1 + a d^b) (0 where d <= 0, umap.py:252-256) | repulsion -2b / ((d + eps)(1 + a d^b))
            const float num = isp[u] ? pb * two_ab : m2b;
            const float dd = isp[u] ? d : d + P.eps;
            float coef = num * fast_rcp(dd * den);
            if (!v[u] || (isp[u] && !(d > 0.f))) coef = 0.f;
            const float ca = isp[u] ? coef : 0.f, cr = isp[u] ? 0.f : coef;
#pragma unroll
            for (int c = 0; c < NC; ++c) { ga[c] += ca * df[c]; gr[c] += cr * df[c]; }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = group_sum_dpp<G>(ga[c]); gr[c] = group_sum_dpp<G>(gr[c]); }
    if (gl == 0 && P.joint) {  // this slice's partial sums; umap_sched_combine_kernel adds the planes in slice order
        float* acc = P.acc + ((size_t)slice * P.n_rows + r) * 2 * nc;
        if (NC == 2 && !PAD) {
            // streamed out: the planes are read next by the combine kernel (other workgroups, other XCDs) and would only push
            // lines of this XCD's slice of Z out of its L2
            __builtin_nontemporal_store(sched_f32x4{ga[0], ga[1], gr[0], gr[1]}, reinterpret_cast<sched_f32x4*>(acc));
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < nc) { acc[c] = ga[c]; acc[nc + c] = gr[c]; }
        }
    } else if (gl == 0) {
        float* acc = P.acc + (size_t)r * 2 * nc;
        if (P.slice > 0) {
            if (NC == 2 && !PAD) {
                const float4 t = *reinterpret_cast<const float4*>(acc);
                ga[0] += t.x; ga[1] += t.y; gr[0] += t.z; gr[1] += t.w;
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (c < nc) { ga[c] += acc[c]; gr[c] += acc[nc + c]; }
            }
        }
        if (P.slice == P.S - 1) {
            float g[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] = P.exag * fminf(fmaxf(ga[c], -4.f), 4.f) + P.rep * fminf(fmaxf(gr[c], -4.f), 4.f);
            if (NC == 2 && !PAD) {
                *reinterpret_cast<float2*>(P.grad + (size_t)r * 2) = make_float2(g[0], g[1]);
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (c < nc) P.grad[(size_t)r * nc + c] = g[c];
            }
        } else if (NC == 2 && !PAD) {
            *reinterpret_cast<float4*>(acc) = make_float4(ga[0], ga[1], gr[0], gr[1]);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < nc) { acc[c] = ga[c]; acc[nc + c] = gr[c]; }
        }
    }
}

// elementwise forms of the two finishing kernels for any row width nc (thread = one component of one row)
__global__ __launch_bounds__(256) void umap_sched_combine_any_kernel(const float* __restrict__ acc, int S, int64_t n_rows, int nc, float exag,
                                                                     float rep, float* __restrict__ grad, float* __restrict__ Z,
                                                                     float* __restrict__ buf, float lr, float momentum, int first,
                                                                     int* __restrict__ nan_flag, int iter) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * nc) return;
    const int64_t r = i / nc;
    const int c = (int)(i - r * nc);
    float ga = acc[(size_t)r * 2 * nc + c], gr = acc[(size_t)r * 2 * nc + nc + c];
    for (int s = 1; s < S; ++s) {
        const float* a = acc + ((size_t)s * n_rows + r) * 2 * nc;
        ga = a[c] + ga;
        gr = a[nc + c] + gr;
    }
    float g = exag * fminf(fmaxf(ga, -4.f), 4.f) + rep * fminf(fmaxf(gr, -4.f), 4.f);
    grad[i] = g;
    if (!Z) return;  // combine only
    if (momentum != 0.f) {
        const float bprev = first ? 0.f : buf[i];
        g = first ? g : __fadd_rn(__fmul_rn(bprev, momentum), g);
        buf[i] = g;
    }
    const float z = fmaf(-lr, g, Z[i]);
    Z[i] = z;
    if (z != z) atomicCAS(nan_flag, 0, iter + 1);
}

// joint launch: gradient = exag * clamp(sum of the attraction planes) + rep * clamp(sum of the repulsion planes), the planes
// added in slice order (the association of the per-slice launches: bit-identical results)
template <int NC>
__global__ __launch_bounds__(256) void umap_sched_combine_kernel(const float* __restrict__ acc, int S, int64_t n_rows, float exag,
                                                                 float rep, float* __restrict__ grad) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float ga[NC], gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = acc[(size_t)r * 2 * NC + c]; gr[c] = acc[(size_t)r * 2 * NC + NC + c]; }
    for (int s = 1; s < S; ++s) {
        const float* a = acc + ((size_t)s * n_rows + r) * 2 * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) { ga[c] = a[c] + ga[c]; gr[c] = a[NC + c] + gr[c]; }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) grad[(size_t)r * NC + c] = exag * fminf(fmaxf(ga[c], -4.f), 4.f) + rep * fminf(fmaxf(gr[c], -4.f), 4.f);
}

// combine + torch.optim.SGD step in one pass over the rows (same arithmetic as umap_sched_combine_kernel followed by
// sgd_step_kernel of tdr_embed.hip): saves a launch and the round trip of the gradient through memory
template <int NC>
__global__ __launch_bounds__(256) void umap_sched_combine_sgd_kernel(const float* __restrict__ acc, int S, int64_t n_rows, float exag,
                                                                     float rep, float* __restrict__ grad, float* __restrict__ Z,
                                                                     float* __restrict__ buf, float lr, float momentum, int first,
                                                                     int* __restrict__ nan_flag, int iter) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float ga[NC], gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = acc[(size_t)r * 2 * NC + c]; gr[c] = acc[(size_t)r * 2 * NC + NC + c]; }
    for (int s = 1; s < S; ++s) {
        const float* a = acc + ((size_t)s * n_rows + r) * 2 * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) { ga[c] = a[c] + ga[c]; gr[c] = a[NC + c] + gr[c]; }
    }
    bool nan = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int64_t i = r * NC + c;
        float g = exag * fminf(fmaxf(ga[c], -4.f), 4.f) + rep * fminf(fmaxf(gr[c], -4.f), 4.f);
        if (grad) grad[i] = g;      // NULL where nobody reads the gradient of this iteration (8 of the pass's 56 MB at N = 1M)
        if (momentum != 0.f) {
            const float bprev = first ? 0.f : buf[i];
            g = first ? g : __fadd_rn(__fmul_rn(bprev, momentum), g);
            buf[i] = g;
        }
        const float z = fmaf(-lr, g, Z[i]);
        Z[i] = z;
        nan = nan || z != z;
    }
    if (nan) atomicCAS(nan_flag, 0, iter + 1);
}

// host side of the per-pass constants (must mirror slice_count / slice_negative of tdr_embed_common.h)
static void sched_pass_constants(SchedGradParams& P, int slice) {
    const uint32_t nred = (uint32_t)(P.n_total - 1);
    const uint32_t S = (uint32_t)P.S;
    P.slice = slice;
    P.step = (nred + S - 1u) / S;
    P.r_lo = (uint32_t)slice * P.step;
    P.r_len = (P.r_lo < nred) ? ((nred - P.r_lo < P.step) ? nred - P.r_lo : P.step) : 0u;
    int level = 0;
    for (int span = P.S; span > 1; span >>= 1, ++level) {
        const uint32_t group = (uint32_t)(slice / span);
        P.lvl_xor[level] = level == 0 ? 0x9E3779B9u
                         : level == 1 ? 0x85EBCA6Bu + 0x27D4EB2Fu * group
                                      : 0xC2B2AE35u + 0x165667B1u * group;
        P.lvl_upper[level] = (slice / (span >> 1)) & 1;
    }
    P.n_levels = level;
    for (; level < 3; ++level) { P.lvl_xor[level] = 0; P.lvl_upper[level] = 0; }
}

template <int NC, int G>
static int launch_sched_grad(const SchedGradParams& P, hipStream_t st) {
    const int rpb = 256 / G;
    int64_t blocks = (P.n_rows + rpb - 1) / rpb;
    if (P.joint) { const int per = 8 / P.S; blocks = ((blocks + per - 1) / per) * 8; }
    const dim3 grid((unsigned)blocks);
    if (P.neg_inj) hipLaunchKernelGGL((umap_sched_grad_kernel<NC, G, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((umap_sched_grad_kernel<NC, G, false>), grid, dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

// geom: lanes per row (0 = default 4: 16 items per round and row, 0.303 ms at N = 1M with 2 slices; 8 lanes 0.309,
// 16 lanes 0.326, 2 lanes 0.319; 4 slices: 0.325 vs 0.375 with 8 lanes) -- ablation knob of tools/umap_sched_perf.py
template <int NC>
static int launch_sched_grad_geom(const SchedGradParams& P, int geom, hipStream_t st) {
    switch (geom) {
        case 1: return launch_sched_grad<NC, 8>(P, st);
        case 2: return launch_sched_grad<NC, 16>(P, st);
        case 3: return launch_sched_grad<NC, 2>(P, st);
        default: return launch_sched_grad<NC, 4>(P, st);
    }
}

// padded instances (any n_components <= NC, default lane geometry)
template <int NC>
static int launch_sched_grad_pad(const SchedGradParams& P, hipStream_t st) {
    constexpr int G = 4;
    const int rpb = 256 / G;
    int64_t blocks = (P.n_rows + rpb - 1) / rpb;
    if (P.joint) { const int per = 8 / P.S; blocks = ((blocks + per - 1) / per) * 8; }
    const dim3 grid((unsigned)blocks);
    if (P.neg_inj) hipLaunchKernelGGL((umap_sched_grad_kernel<NC, G, true, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((umap_sched_grad_kernel<NC, G, false, true>), grid, dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}
// one launch of the gradient kernel for row width P.nc: exact instances for 2 and 3, padded ones up to SCHED_NC_MAX
constexpr int SCHED_NC_MAX = 32;
static int launch_sched_grad_nc(const SchedGradParams& P, int geom, hipStream_t st) {
    if (P.nc == 2) return launch_sched_grad_geom<2>(P, geom, st);
    if (P.nc == 3) return launch_sched_grad_geom<3>(P, geom, st);
    if (P.nc <= 4) return launch_sched_grad_pad<4>(P, st);
    if (P.nc <= 8) return launch_sched_grad_pad<8>(P, st);
    if (P.nc <= 16) return launch_sched_grad_pad<16>(P, st);
    return launch_sched_grad_pad<32>(P, st);
}

// launch the schedule kernel (stash depth: chunks of 16 edges per row kept in registers between the two phases)
static int launch_sched_build(const SchedBuildParams& P0, hipStream_t st, bool set_attr) {
    SchedBuildParams P = P0;
    const int64_t n_blocks = (P.n_rows + SCHED_RB - 1) / SCHED_RB;
    const size_t lds = (size_t)P.B * P.S * CNT_STRIDE * sizeof(uint32_t);
    // depth 2 measured best at N = 1M (2.02 ms per window; 0: 2.29, 1: 2.16, 3: 2.04 -- registers cost occupancy)
    const int depth = P.stash ? 2 : 0;
#define TDR_BUILD(D)                                                                                                  \
    do {                                                                                                              \
        if (set_attr && lds > 32 * 1024) {                                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(umap_sched_build_kernel<D>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
            if (e != hipSuccess) return (int)e;                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL(umap_sched_build_kernel<D>, dim3((unsigned)n_blocks), dim3(256), lds, st, P);              \
    } while (0)
    if (depth == 0) TDR_BUILD(1);  // column ids beyond 2^29: the depth-1 instance with stash = 0 keeps nothing
    else TDR_BUILD(2);
#undef TDR_BUILD
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

// launch the group-ordered schedule kernel: one wavefront per group of 16 rows
static size_t sched_build2_lds(int B, int S, int stage) { return ((size_t)((B * S * G2_STRIDE + 3) & ~3) + (size_t)stage) * sizeof(uint32_t); }
static int launch_sched_build2(const SchedBuild2Params& P0, hipStream_t st, bool set_attr) {
    SchedBuild2Params P = P0;
    const int variant = (P.stage >> 16) & 15;   // tuning knob of tools/sched_build2_perf.py: register budget of the kernel
    P.stage &= 0xffff;
    if (P.stage == 0) P.stage = G2_STAGE_DEFAULT;
    const int64_t n_groups = (P.n_rows + G2_ROWS - 1) / G2_ROWS;
    const size_t lds = sched_build2_lds(P.B, P.S, P.stage);
#define TDR_BUILD2(W, T)                                                                                               \
    do {                                                                                                              \
        if (set_attr && lds > 32 * 1024) {                                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(umap_sched_build2_kernel<W, T>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
            if (e != hipSuccess) return (int)e;                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((umap_sched_build2_kernel<W, T>), dim3((unsigned)n_groups), dim3(64), lds, st, P);        \
    } while (0)
    // runs of G2_TC = 4 iterations with a 768-entry stage: 1.08 ms per window at N = 1M against 1.15 with runs of 8 and 1024
    // entries (same box; the mean run is 550 entries, nearly all of them staged).  The run length enters the ORDER inside a
    // segment (rank of a firing inside its run), so there is one production value; the register budget is a pure speed knob.
    if (variant == 4) TDR_BUILD2(4, G2_TC);
    else if (variant == 6) TDR_BUILD2(6, G2_TC);
    else TDR_BUILD2(5, G2_TC);   // 96 registers (8 spilled); 4 -> 126 registers: +8 %, 6 -> 80 registers (39 spilled): +50 %
#undef TDR_BUILD2
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

// all slice passes of one evaluation: one launch per slice (geom < 16), or -- geom & 16, S > 1 -- ONE joint launch with the
// slices spread over the XCDs and a combine kernel; acc must then hold S planes of (n_rows, 2 nc) floats
static int launch_sched_grad_all(SchedGradParams& P, int geom, hipStream_t st) {
    P.joint = 0;
    if ((geom & 16) && P.S > 1) {
        for (int s = 0; s < P.S; ++s) {
            sched_pass_constants(P, s);
            P.j_r_lo[s] = P.r_lo; P.j_r_len[s] = P.r_len;
            for (int l = 0; l < 3; ++l) { P.j_lvl_xor[s][l] = P.lvl_xor[l]; P.j_lvl_upper[s][l] = P.lvl_upper[l]; }
        }
        P.joint = 1;
        const int rc = launch_sched_grad_nc(P, geom & 15, st);
        if (rc != TDR_OK) return rc;
        if (geom & 32) return TDR_OK;  // the caller finishes with tdr_umap_sched_step_f32 (combine + SGD step)
        if (P.nc == 2) hipLaunchKernelGGL(umap_sched_combine_kernel<2>, dim3((unsigned)((P.n_rows + 255) / 256)), dim3(256), 0, st,
                                          (const float*)P.acc, P.S, P.n_rows, P.exag, P.rep, P.grad);
        else if (P.nc == 3) hipLaunchKernelGGL(umap_sched_combine_kernel<3>, dim3((unsigned)((P.n_rows + 255) / 256)), dim3(256), 0, st,
                                               (const float*)P.acc, P.S, P.n_rows, P.exag, P.rep, P.grad);
        else hipLaunchKernelGGL(umap_sched_combine_any_kernel, dim3((unsigned)((P.n_rows * P.nc + 255) / 256)), dim3(256), 0, st,
                                (const float*)P.acc, P.S, P.n_rows, P.nc, P.exag, P.rep, P.grad, (float*)nullptr, (float*)nullptr, 0.f, 0.f, 0,
                                (int*)nullptr, 0);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? TDR_OK : (int)e;
    }
    for (int s = 0; s < P.S; ++s) {
        sched_pass_constants(P, s);
        const int rc = launch_sched_grad_nc(P, geom & 15, st);
        if (rc != TDR_OK) return rc;
    }
    return TDR_OK;
}

// ---- the optimisation loop as one object ------------------------------------------------------------------------------
// SGD step of the loop runner: torch.optim.SGD semantics (affinity_matcher.py:427) with the learning rate read from a
// device table at the global iteration (device iteration base + offset), the NaN flag of check_NaNs (:315), and -- at
// the iterations the reference inspects the gradient (n_iter % check_interval == 0, :331-349) -- the squared gradient
// norm accumulated into norm2[n_iter / check_interval] and the stepped rows copied to `snap`, so that a host that runs
// whole windows ahead can still return exactly the state of the iteration at which the reference would have stopped.
__global__ __launch_bounds__(256) void sgd_table_step_kernel(float* __restrict__ Z, const float* __restrict__ grad,
                                                             float* __restrict__ buf, int64_t n, const float* __restrict__ lr_table,
                                                             const int* __restrict__ iter_base, int iter_off, float momentum,
                                                             int first_iter, int check_interval, float* __restrict__ norm2,
                                                             float* __restrict__ snap, int* __restrict__ nan_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int it = *iter_base + iter_off;
    float g = i < n ? grad[i] : 0.f;
    const bool inspected = check_interval > 0 && it % check_interval == 0;
    if (inspected) {
        float q = g * g;
        q = wave_sum(q);
        if ((threadIdx.x & 63) == 0 && q != 0.f) atomicAdd(&norm2[it / check_interval], q);
    }
    if (i >= n) return;
    if (momentum != 0.f) {
        const bool first = it == first_iter;
        const float bprev = first ? 0.f : buf[i];
        g = first ? g : __fadd_rn(__fmul_rn(bprev, momentum), g);
        buf[i] = g;
    }
    const float z = fmaf(-lr_table[it], g, Z[i]);
    Z[i] = z;
    if (inspected && snap) snap[i] = z;  // the state the reference returns if it stops here (:343-349)
    if (z != z) atomicCAS(nan_flag, 0, it + 1);
}

// The two kernels above in one pass for the joint launch of the loop object (round 4): umap_sched_combine_kernel's sums and
// clamps followed by sgd_table_step_kernel's step, element for element the same operations -- one launch and the round trip of
// the gradient through memory less per iteration (a rank of an 8-GPU fit at N = 1M spends ~40 us per iteration in kernels).
// The gradient itself is written at the inspected iterations only (nobody reads it elsewhere).
template <int NC>
__global__ __launch_bounds__(256) void umap_sched_combine_table_step_kernel(const float* __restrict__ acc, int S, int64_t n_rows, float exag,
                                                                            float rep, float* __restrict__ grad, float* __restrict__ Z,
                                                                            float* __restrict__ buf, const float* __restrict__ lr_table,
                                                                            const int* __restrict__ iter_base, int iter_off, float momentum,
                                                                            int first_iter, int check_interval, float* __restrict__ norm2,
                                                                            float* __restrict__ snap, int* __restrict__ nan_flag) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int it = *iter_base + iter_off;
    const bool inspected = check_interval > 0 && it % check_interval == 0;
    const bool have = r < n_rows;
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    if (have) {
        float ga[NC], gr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { ga[c] = acc[(size_t)r * 2 * NC + c]; gr[c] = acc[(size_t)r * 2 * NC + NC + c]; }
        for (int s = 1; s < S; ++s) {
            const float* a = acc + ((size_t)s * n_rows + r) * 2 * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c) { ga[c] = a[c] + ga[c]; gr[c] = a[NC + c] + gr[c]; }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) g[c] = exag * fminf(fmaxf(ga[c], -4.f), 4.f) + rep * fminf(fmaxf(gr[c], -4.f), 4.f);
    }
    if (inspected) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) q += g[c] * g[c];
        q = wave_sum(q);
        if ((threadIdx.x & 63) == 0 && q != 0.f) atomicAdd(&norm2[it / check_interval], q);
    }
    if (!have) return;
    const float lr = lr_table[it];
    const bool first = it == first_iter;
    bool nan = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int64_t i = r * NC + c;
        float gc = g[c];
        if (inspected) grad[i] = gc;
        if (momentum != 0.f) {
            const float bprev = first ? 0.f : buf[i];
            gc = first ? gc : __fadd_rn(__fmul_rn(bprev, momentum), gc);
            buf[i] = gc;
        }
        const float z = fmaf(-lr, gc, Z[i]);
        Z[i] = z;
        if (inspected && snap) snap[i] = z;
        nan = nan || z != z;
    }
    if (nan) atomicCAS(nan_flag, 0, it + 1);
}

typedef int (*tdr_collective_fn)(void* ctx, float* Z, int nc, void* stream);  // all-gather of the rows every rank stepped

struct UmapLoop {
    // static description of the problem (device pointers are the caller's)
    float* Z; int nc; int64_t n_total, row0, n_rows;
    const int64_t* rowptr; const int32_t* cols; const float* eps_per; float* next;
    const int64_t* blk_base; int32_t* list; uint2* hdr; int* err; float* acc; float* grad; float* mom_buf;
    float a, b; int neg_rate, n_negatives; uint64_t seed; float exag, rep, eps; int S, B;
    const float* lr_table; int max_iter; float momentum; int first_iter; int check_interval; float* norm2; float* snap; int* nan_flag;
    int* iter_base;                 // device int (caller's scratch)
    tdr_collective_fn gather; void* gather_ctx;
    int geom;
    const uint8_t* rs;              // non-null: group-ordered loop state (cols / eps_per / next / blk_base of tdr_umap_sched_group_f32 / _plan_groups_f32)
    int pool;                       // g + 1: negatives from the LDS pool (tdr_umap_pool.hip, geometry g), 0: i.i.d. gathers
    int gather_capturable;          // the gather callback may be captured into the window graphs
    float* Z_alt;                   // second embedding buffer: the pool gradient launch carries the SGD step (momentum 0) and the two swap roles
    // captured windows: graph_len[i] iterations each
    hipGraphExec_t graphs[2]; int graph_len[2];
};

// enqueue one window: schedule build for iterations [base + 0, base + n) and n x (S gradient passes + SGD step [+ row
// all-gather]); `base` lives in L->iter_base on the device, so the SAME enqueued sequence serves any window
// host_base >= 0 (plain launches): the window's first iteration is known here and goes to the gradient kernel as an argument --
// its row key, the head of every wavefront's dependency chain, then does not wait for a load of the device-side base
static int umap_loop_enqueue_window(UmapLoop* L, int n, hipStream_t st, int host_base = -1) {
    if (L->rs) {
        SchedBuild2Params B2;
        B2.rowptr = L->rowptr; B2.cols = L->cols; B2.eps_per = L->eps_per; B2.rs = L->rs; B2.next = L->next; B2.n_rows = L->n_rows;
        B2.t0 = 0; B2.B = n; B2.S = L->S; B2.stage = G2_STAGE_DEFAULT; B2.iter_base = L->iter_base; B2.grp_base = L->blk_base;
        B2.list = L->list; B2.hdr = L->hdr; B2.err = L->err;
        const int rc2 = launch_sched_build2(B2, st, false);
        if (rc2 != TDR_OK) return rc2;
    }
    SchedBuildParams Bp;
    Bp.rowptr = L->rowptr; Bp.cols = L->cols; Bp.eps_per = L->eps_per; Bp.next = L->next; Bp.n_rows = L->n_rows;
    const uint32_t nred = (uint32_t)(L->n_total - 1);
    Bp.slice_step = (nred + (uint32_t)L->S - 1u) / (uint32_t)L->S;
    Bp.stash = L->n_total <= (1LL << 29) ? 1 : 0;
    Bp.t0 = 0; Bp.iter_base = L->iter_base; Bp.B = n; Bp.S = L->S; Bp.blk_base = L->blk_base; Bp.list = L->list; Bp.hdr = L->hdr;
    Bp.err = L->err;
    const int rcb = L->rs ? TDR_OK : launch_sched_build(Bp, st, false);
    if (rcb != TDR_OK) return rcb;
    SchedGradParams G = {};
    G.Z = L->Z; G.n_total = L->n_total; G.row0 = L->row0; G.n_rows = L->n_rows; G.list = L->list; G.hdr = L->hdr; G.S = L->S;
    G.a = L->a; G.b = L->b; G.neg_rate = L->neg_rate; G.n_negatives = L->n_negatives; G.neg_inj = nullptr; G.seed = L->seed;
    G.iter_base = L->iter_base; G.exag = L->exag; G.rep = L->rep; G.eps = L->eps; G.grad = L->grad; G.acc = L->acc;
    const int64_t n_el = L->n_rows * L->nc;
    if (host_base >= 0) G.iter_base = nullptr;
    PoolGradParams Pp = {};
    Pp.Z = L->Z; Pp.nc = L->nc; Pp.n_total = L->n_total; Pp.row0 = L->row0; Pp.n_rows = L->n_rows; Pp.list = L->list; Pp.hdr = L->hdr;
    Pp.a = L->a; Pp.b = L->b; Pp.neg_rate = L->neg_rate; Pp.n_negatives = L->n_negatives; Pp.seed = L->seed; Pp.iter_base = G.iter_base;
    Pp.exag = L->exag; Pp.rep = L->rep; Pp.eps = L->eps; Pp.grad = L->grad; Pp.exact5 = (L->neg_rate == 5 && L->n_negatives % 5 == 0) ? 1 : 0;
    for (int t = 0; t < n; ++t) {
        G.t_local = t; G.iter = (uint32_t)(t + (host_base >= 0 ? host_base : 0));
        G.nc = L->nc;
        if (L->pool && L->Z_alt && L->momentum == 0.f) {
            // gradient + torch.optim.SGD step in ONE launch (round 6): iteration t of the window reads buffer t & 1 and writes the
            // stepped rows of this rank into the other one, where the exchange then delivers the other ranks' rows; learning rate,
            // NaN flag, and at the inspected iterations gradient / norm / snapshot as sgd_table_step_kernel leaves them.  A window
            // of odd length ends with the current rows in Z_alt: they are copied back, so every window starts from L->Z.
            float* cur = (t & 1) ? L->Z_alt : L->Z;
            float* nxt = (t & 1) ? L->Z : L->Z_alt;
            Pp.t_local = t; Pp.iter = G.iter; Pp.Z = cur; Pp.Z_out = nxt; Pp.lr_table = L->lr_table; Pp.check_interval = L->check_interval;
            Pp.norm2 = L->norm2; Pp.snap = L->snap; Pp.nan_flag = L->nan_flag;
            const int rcp = launch_pool_grad(Pp, L->pool - 1, st);
            if (rcp != TDR_OK) return rcp;
            if (L->gather) {
                const int rc = L->gather(L->gather_ctx, nxt, L->nc, (void*)st);
                if (rc != TDR_OK) return rc;
            }
            if (t == n - 1 && (n & 1)) {
                hipError_t ec = hipMemcpyAsync(L->Z, L->Z_alt, (size_t)L->n_total * L->nc * sizeof(float), hipMemcpyDeviceToDevice, st);
                if (ec != hipSuccess) return (int)ec;
            }
            continue;
        }
        if (L->pool) {
            Pp.t_local = t; Pp.iter = G.iter;
            const int rcp = launch_pool_grad(Pp, L->pool - 1, st);
            if (rcp != TDR_OK) return rcp;
            hipLaunchKernelGGL(sgd_table_step_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st,
                               L->Z + L->row0 * L->nc, (const float*)L->grad, L->mom_buf, n_el, L->lr_table, (const int*)L->iter_base, t,
                               L->momentum, L->first_iter, L->check_interval, L->norm2, L->snap, L->nan_flag);
            if (L->gather) {
                const int rc = L->gather(L->gather_ctx, L->Z, L->nc, (void*)st);
                if (rc != TDR_OK) return rc;
            }
            continue;
        }
        // joint launch, 2 or 3 components: the gradient kernel leaves the per-slice planes (geom bit 32) and ONE kernel combines
        // them and steps the rows
        const bool fused = (L->geom & 16) && L->S > 1 && (L->nc == 2 || L->nc == 3);
        const int rcg = launch_sched_grad_all(G, fused ? (L->geom | 32) : L->geom, st);
        if (rcg != TDR_OK) return rcg;
        if (fused && L->nc == 2)
            hipLaunchKernelGGL(umap_sched_combine_table_step_kernel<2>, dim3((unsigned)((L->n_rows + 255) / 256)), dim3(256), 0, st,
                               (const float*)L->acc, L->S, L->n_rows, L->exag, L->rep, L->grad, L->Z + L->row0 * L->nc, L->mom_buf, L->lr_table,
                               (const int*)L->iter_base, t, L->momentum, L->first_iter, L->check_interval, L->norm2, L->snap, L->nan_flag);
        else if (fused)
            hipLaunchKernelGGL(umap_sched_combine_table_step_kernel<3>, dim3((unsigned)((L->n_rows + 255) / 256)), dim3(256), 0, st,
                               (const float*)L->acc, L->S, L->n_rows, L->exag, L->rep, L->grad, L->Z + L->row0 * L->nc, L->mom_buf, L->lr_table,
                               (const int*)L->iter_base, t, L->momentum, L->first_iter, L->check_interval, L->norm2, L->snap, L->nan_flag);
        else
            hipLaunchKernelGGL(sgd_table_step_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st,
                               L->Z + L->row0 * L->nc, (const float*)L->grad, L->mom_buf, n_el, L->lr_table, (const int*)L->iter_base, t,
                               L->momentum, L->first_iter, L->check_interval, L->norm2, L->snap, L->nan_flag);
        if (L->gather) {
            const int rc = L->gather(L->gather_ctx, L->Z, L->nc, (void*)st);
            if (rc != TDR_OK) return rc;
        }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* Number of L2 slices of the embedding the scheduled gradient passes use: 1 while Z fits an XCD's L2 (<= 3 MiB), else
 * the power of two (<= 8) that brings a slice to <= 4 MiB. */
int tdr_umap_sched_slices(int64_t n_total, int nc) {
    const int64_t bytes = n_total * nc * (int64_t)sizeof(float);
    if (bytes <= (int64_t)3 << 20) return 1;
    int s = 2;
    while (s < 8 && bytes > (int64_t)s * ((int64_t)4 << 20)) s *= 2;
    return s;
}

/* uint2 (8-byte) records of the `hdr` table for a window of block_iters iterations and n_slices slices. */
int64_t tdr_umap_sched_hdr_entries(int64_t n_rows, int block_iters, int n_slices) {
    if (n_rows <= 0 || block_iters <= 0 || n_slices <= 0) return 0;
    return (int64_t)block_iters * n_slices * n_rows;
}

/* Static plan: blk_base (n_blocks + 1, n_blocks = ceil(n_rows / 64)) = exclusive scan of the blocks' list capacities for
 * windows of block_iters (<= 32) iterations; blk_base[n_blocks] = int32 entries `list` must hold, PLUS 64 entries of slack (idle lanes of the gradient kernel read entry 0 of a segment).  scratch: n_blocks int64. */
int tdr_umap_sched_plan_f32(const int64_t* rowptr, const float* eps_per, int64_t n_rows, int block_iters, int64_t* scratch,
                            int64_t* blk_base, void* stream) {
    if (!rowptr || !eps_per || !scratch || !blk_base || n_rows <= 0 || block_iters <= 0 || block_iters > SCHED_BMAX)
        return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_blocks = (n_rows + SCHED_RB - 1) / SCHED_RB;
    hipLaunchKernelGGL(umap_sched_plan_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, rowptr, eps_per, n_rows, block_iters,
                       SCHED_RB, scratch);
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(256), 0, st, (const int64_t*)scratch, n_blocks, blk_base);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Loop layout of the edges: every row's (cols, eps_per) reordered by ascending eps_per (often-firing edges first), which
 * makes the schedule kernel's per-lane firing loops homogeneous.  Any edge order gives the same gradient up to the
 * order of the fp32 force sum; the CSR itself (column-sorted, utils/sparse.py semantics) is left untouched. */
int tdr_umap_sched_layout_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, int64_t n_rows,
                              int32_t* cols_out, float* eps_out, void* stream) {
    if (!rowptr || !cols || !eps_per || !cols_out || !eps_out || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(umap_sched_layout_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr, cols,
                       eps_per, n_rows, cols_out, eps_out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Advance epoch_of_next_sample (`next`) by n_iters (<= 32) iterations starting at iteration t0 and emit the firing lists
 * (layout: file header).  hdr: tdr_umap_sched_hdr_entries(...) 8-byte records.  err: device int, set to 1 if a block's
 * region would overflow, 2 if a segment exceeds 65535 entries or the list 2^32 entries. */
int tdr_umap_sched_build_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, float* next, int64_t n_rows,
                             int64_t n_total, int t0, int n_iters, int n_slices, const int64_t* blk_base, int32_t* list,
                             void* hdr, int* err, void* stream) {
    if (!rowptr || !cols || !eps_per || !next || !blk_base || !list || !hdr || !err) return TDR_ERR_BAD_ARG;
    if (n_rows <= 0 || n_total < 2 || n_total >= 0x7fffffffLL || t0 < 0 || n_iters <= 0 || n_iters > SCHED_BMAX || t0 > (1 << 24) - 64) return TDR_ERR_BAD_ARG;
    if (n_slices != 1 && n_slices != 2 && n_slices != 4 && n_slices != 8) return TDR_ERR_BAD_ARG;
    SchedBuildParams P;
    P.rowptr = rowptr; P.cols = cols; P.eps_per = eps_per; P.next = next; P.n_rows = n_rows;
    const uint32_t nred = (uint32_t)(n_total - 1);
    P.slice_step = (nred + (uint32_t)n_slices - 1u) / (uint32_t)n_slices;
    P.stash = n_total <= (1LL << 29) ? 1 : 0;
    P.t0 = t0; P.iter_base = nullptr; P.B = n_iters; P.S = n_slices; P.blk_base = blk_base; P.list = list; P.hdr = (uint2*)hdr; P.err = err;
    return launch_sched_build(P, (hipStream_t)stream, true);
}

/* Group order of the loop state (round 4; see umap_sched_build2_kernel): the edges of every group of 16 consecutive rows,
 * given row-major with every row in (period, column) order (tdr_umap_sched_layout_f32), sorted stably by period class.
 * cols_g / eps_g (nnz), rs_g (nnz bytes: local row | slice << 4 for n_slices slices of n_total points), order_g (nnz int32:
 * the edge's row-major position relative to its group's first edge).  err: device int, 2 = a group beyond 2^31 edges. */
int tdr_umap_sched_group_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, int64_t n_rows, int64_t n_total,
                             int n_slices, int32_t* cols_g, float* eps_g, uint8_t* rs_g, int32_t* order_g, int* err, void* stream) {
    if (!rowptr || !cols || !eps_per || !cols_g || !eps_g || !rs_g || !order_g || !err || n_rows <= 0 || n_total < 2 || n_total >= 0x7fffffffLL)
        return TDR_ERR_BAD_ARG;
    if (n_slices != 1 && n_slices != 2 && n_slices != 4 && n_slices != 8) return TDR_ERR_BAD_ARG;
    const uint32_t nred = (uint32_t)(n_total - 1);
    const uint32_t slice_step = (nred + (uint32_t)n_slices - 1u) / (uint32_t)n_slices;
    hipLaunchKernelGGL(umap_sched_group_kernel, dim3((unsigned)((n_rows + G2_ROWS - 1) / G2_ROWS)), dim3(64), 0, (hipStream_t)stream, rowptr, cols,
                       eps_per, n_rows, slice_step, n_slices, cols_g, eps_g, rs_g, order_g, err);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* vals_rm[e] = the group-ordered per-edge value of row-major edge e (e.g. epoch_of_next_sample for inspection). */
int tdr_umap_sched_ungroup_f32(const int64_t* rowptr, const int32_t* order_g, const float* vals_g, int64_t n_rows, float* vals_rm,
                               void* stream) {
    if (!rowptr || !order_g || !vals_g || !vals_rm || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(umap_sched_ungroup_kernel, dim3((unsigned)((n_rows + G2_ROWS - 1) / G2_ROWS)), dim3(64), 0, (hipStream_t)stream, rowptr,
                       order_g, vals_g, n_rows, vals_rm);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Static plan of the group-ordered build: grp_base (n_groups + 1, n_groups = ceil(n_rows / 16)), as tdr_umap_sched_plan_f32
 * with 16-row blocks (eps_per in either order: a group's capacity is a sum over its edges).  scratch: n_groups int64. */
int tdr_umap_sched_plan_groups_f32(const int64_t* rowptr, const float* eps_per, int64_t n_rows, int block_iters, int64_t* scratch,
                                   int64_t* grp_base, void* stream) {
    if (!rowptr || !eps_per || !scratch || !grp_base || n_rows <= 0 || block_iters <= 0 || block_iters > SCHED_BMAX)
        return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_groups = (n_rows + G2_ROWS - 1) / G2_ROWS;
    hipLaunchKernelGGL(umap_sched_plan_kernel, dim3((unsigned)n_groups), dim3(256), 0, st, rowptr, eps_per, n_rows, block_iters,
                       G2_ROWS, scratch);
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(256), 0, st, (const int64_t*)scratch, n_groups, grp_base);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* tdr_umap_sched_build_f32 on group-ordered state: same records and lists (regions per 16-row group), `next_g` advanced
 * bit-exactly; a segment lists the row's firing edges in (period, column) order.  stage: LDS stage entries (0 = default). */
int tdr_umap_sched_build_groups_f32(const int64_t* rowptr, const int32_t* cols_g, const float* eps_g, const uint8_t* rs_g, float* next_g,
                                    int64_t n_rows, int t0, int n_iters, int n_slices, const int64_t* grp_base, int32_t* list,
                                    void* hdr, int* err, int stage, void* stream) {
    if (!rowptr || !cols_g || !eps_g || !rs_g || !next_g || !grp_base || !list || !hdr || !err) return TDR_ERR_BAD_ARG;
    if (n_rows <= 0 || t0 < 0 || n_iters <= 0 || n_iters > SCHED_BMAX || t0 > (1 << 24) - 64) return TDR_ERR_BAD_ARG;
    if (n_slices != 1 && n_slices != 2 && n_slices != 4 && n_slices != 8) return TDR_ERR_BAD_ARG;
    if ((stage & 0xffff) != 0 && ((stage & 0xffff) < 512 || (stage & 0xffff) > 24576)) return TDR_ERR_BAD_ARG;
    SchedBuild2Params P;
    P.rowptr = rowptr; P.cols = cols_g; P.eps_per = eps_g; P.rs = rs_g; P.next = next_g; P.n_rows = n_rows;
    P.t0 = t0; P.B = n_iters; P.S = n_slices; P.stage = stage; P.iter_base = nullptr; P.grp_base = grp_base;
    P.list = list; P.hdr = (uint2*)hdr; P.err = err;
    return launch_sched_build2(P, (hipStream_t)stream, true);
}

/* One evaluation of UMAP's closed-form gradient (umap.py:236-292) for rows [row0, row0 + n_rows) from the lists of
 * tdr_umap_sched_build_f32: t_local = iteration index inside the window, n_iter = global iteration (hash counter).
 * acc: (n_rows, 2 nc) floats (used when n_slices > 1).  geom: low 4 bits = lane geometry (0 = default; tuning knob);
 * bit 4 (16) = all slices in ONE launch spread over the XCDs + a combine kernel: acc then holds n_slices planes of
 * (n_rows, 2 nc) floats; bit 5 (32, with bit 4) = leave the planes in acc: tdr_umap_sched_step_f32 combines and steps. */
int tdr_umap_sched_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list,
                            const void* hdr, int t_local, int n_slices, float a, float b, int n_iter, int neg_rate,
                            int n_negatives, const int64_t* neg_inj, uint64_t seed, float exag, float rep, float eps,
                            float* grad, float* acc, int geom, void* stream) {
    if (!Z || !list || !hdr || !grad || n_rows <= 0 || n_total < 2 || n_total >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    if (t_local < 0 || t_local >= SCHED_BMAX || neg_rate < 0 || n_negatives < 0) return TDR_ERR_BAD_ARG;
    if (n_slices != 1 && n_slices != 2 && n_slices != 4 && n_slices != 8) return TDR_ERR_BAD_ARG;
    if (n_slices > 1 && !acc) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > SCHED_NC_MAX) return TDR_ERR_UNSUPPORTED;
    SchedGradParams P = {};
    P.Z = Z; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.list = list; P.hdr = (const uint2*)hdr;
    P.t_local = t_local; P.S = n_slices; P.a = a; P.b = b; P.neg_rate = neg_rate; P.n_negatives = n_negatives;
    P.neg_inj = neg_inj; P.seed = seed; P.iter = (uint32_t)n_iter; P.iter_base = nullptr; P.exag = exag; P.rep = rep; P.eps = eps; P.grad = grad;
    P.acc = acc;
    hipStream_t st = (hipStream_t)stream;
    P.nc = nc;
    return launch_sched_grad_all(P, geom, st);
}

/* Finish a joint evaluation (tdr_umap_sched_grad_f32 with geom & 48 == 48): gradient = exag * clamp(attraction) + rep *
 * clamp(repulsion) from the n_slices planes of acc (umap.py:262,290), written to grad (n_rows, nc), then the
 * torch.optim.SGD(momentum) step of tdr_sgd_step_f32 on the rows Z (n_rows, nc) -- one kernel, same bits as the two. */
int tdr_umap_sched_step_f32(const float* acc, int n_slices, int nc, int64_t n_rows, float exag, float rep, float* grad, float* Z,
                            float* buf, float lr, float momentum, int first, int* nan_flag, int n_iter, void* stream) {
    if (!acc || !Z || !nan_flag || n_rows <= 0) return TDR_ERR_BAD_ARG;
    if (n_slices != 2 && n_slices != 4 && n_slices != 8) return TDR_ERR_BAD_ARG;
    if (momentum != 0.f && !buf) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > SCHED_NC_MAX) return TDR_ERR_UNSUPPORTED;
    if (!grad && nc != 2 && nc != 3) return TDR_ERR_BAD_ARG;     // the elementwise form always writes the gradient
    const dim3 grid((unsigned)((n_rows + 255) / 256));
    if (nc != 2 && nc != 3) {
        hipLaunchKernelGGL(umap_sched_combine_any_kernel, dim3((unsigned)((n_rows * nc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc,
                           n_slices, n_rows, nc, exag, rep, grad, Z, buf, lr, momentum, first, nan_flag, n_iter);
        TDR_CHECK_LAUNCH();
        return TDR_OK;
    }
    if (nc == 2) hipLaunchKernelGGL(umap_sched_combine_sgd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, acc, n_slices, n_rows, exag, rep,
                                    grad, Z, buf, lr, momentum, first, nan_flag, n_iter);
    else hipLaunchKernelGGL(umap_sched_combine_sgd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, acc, n_slices, n_rows, exag, rep, grad,
                            Z, buf, lr, momentum, first, nan_flag, n_iter);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* ---- the whole optimisation loop behind one handle -------------------------------------------------------------------
 * affinity_matcher.py:288-352 (the training loop) for UMAP's closed-form step with torch.optim.SGD: every window of up to
 * `block_iters` iterations is ONE enqueued sequence (schedule build, then per iteration S gradient passes + the SGD step
 * [+ the all-gather of the rows this rank stepped]) whose iteration base lives in device memory, captured once into a HIP
 * graph and replayed for every window of the same length -- the host issues one graph launch per window instead of
 * ~4 kernel launches per iteration.
 *   desc: device pointers of the problem (see tdr_umap_loop_desc); lr_table: max_iter floats on the device (the learning
 *   rate of every iteration); norm2: ceil(max_iter / check_interval) floats, caller-zeroed (squared gradient norms at the
 *   iterations the reference inspects); scratch: >= 4 bytes of device memory (the iteration base).
 *   gather / gather_ctx: optional collective run after every step (tdr_ctx_allgather_rows of a tdr_ctx), NULL = none. */
int tdr_umap_loop_create(void** out, const tdr_umap_loop_desc* d) {
    if (!out || !d || !d->Z || !d->rowptr || !d->cols || !d->eps_per || !d->next || !d->blk_base || !d->list || !d->hdr || !d->err ||
        !d->grad || !d->lr_table || !d->norm2 || !d->nan_flag || !d->scratch) return TDR_ERR_BAD_ARG;
    if (d->n_rows <= 0 || d->n_total < 2 || d->n_total >= 0x7fffffffLL || d->block_iters <= 0 || d->block_iters > SCHED_BMAX) return TDR_ERR_BAD_ARG;
    if (d->n_slices != 1 && d->n_slices != 2 && d->n_slices != 4 && d->n_slices != 8) return TDR_ERR_BAD_ARG;
    if (d->n_slices > 1 && !d->acc) return TDR_ERR_BAD_ARG;
    if (d->momentum != 0.f && !d->mom_buf) return TDR_ERR_BAD_ARG;
    if (d->nc < 1 || d->nc > SCHED_NC_MAX) return TDR_ERR_UNSUPPORTED;
    UmapLoop* L = new UmapLoop();
    L->Z = d->Z; L->nc = d->nc; L->n_total = d->n_total; L->row0 = d->row0; L->n_rows = d->n_rows; L->rowptr = d->rowptr;
    L->cols = d->cols; L->eps_per = d->eps_per; L->next = d->next; L->blk_base = d->blk_base; L->list = d->list;
    L->hdr = (uint2*)d->hdr; L->err = d->err; L->acc = d->acc; L->grad = d->grad; L->mom_buf = d->mom_buf; L->a = d->a; L->b = d->b;
    L->neg_rate = d->neg_rate; L->n_negatives = d->n_negatives; L->seed = d->seed; L->exag = d->exag; L->rep = d->rep; L->eps = d->eps;
    L->S = d->n_slices; L->B = d->block_iters; L->lr_table = d->lr_table; L->max_iter = d->max_iter; L->momentum = d->momentum;
    L->first_iter = d->first_iter; L->check_interval = d->check_interval; L->norm2 = d->norm2; L->snap = d->snap; L->nan_flag = d->nan_flag;
    L->iter_base = (int*)d->scratch; L->gather = (tdr_collective_fn)d->gather; L->gather_ctx = d->gather_ctx; L->geom = d->geom;
    L->rs = d->rs;
    L->pool = d->pool;
    L->gather_capturable = d->gather_capturable;
    L->Z_alt = d->Z_alt;
    if (L->Z_alt && (L->Z_alt == L->Z || ((uintptr_t)L->Z_alt & 15u))) { delete L; return TDR_ERR_BAD_ARG; }
    if (L->pool < 0 || (L->pool && !tdr_pool_geom_ok(L->pool - 1)) || (L->pool && (L->S != 1 || !tdr_umap_pool_supported(L->nc) || ((uintptr_t)L->Z & 15u) || L->n_total * L->nc * 4 >= 0xffffffffLL))) { delete L; return TDR_ERR_BAD_ARG; }
    L->graphs[0] = L->graphs[1] = nullptr; L->graph_len[0] = L->graph_len[1] = 0;
    const size_t lds = (size_t)L->B * L->S * CNT_STRIDE * sizeof(uint32_t);
    if (lds > 32 * 1024) {  // raised here, outside graph capture, for every instance the launcher may pick
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(umap_sched_build_kernel<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(umap_sched_build_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { delete L; return (int)e; }
    }
    if (L->rs && sched_build2_lds(L->B, L->S, G2_STAGE_DEFAULT) > 32 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(umap_sched_build2_kernel<5, G2_TC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sched_build2_lds(L->B, L->S, G2_STAGE_DEFAULT));
        if (e != hipSuccess) { delete L; return (int)e; }
    }
    *out = L;
    return TDR_OK;
}

/* Run iterations [it0, it0 + n_iters) (n_iters >= 1; it0 + n_iters <= max_iter).  use_graph != 0: windows are captured
 * into HIP graphs on first use (at most two distinct window lengths are kept) and replayed; 0: plain launches.  A loop
 * created with a `gather` callback always runs plain launches (use_graph is ignored): see below. */
int tdr_umap_loop_run(void* loop, int it0, int n_iters, int use_graph, void* stream) {
    UmapLoop* L = (UmapLoop*)loop;
    if (!L || it0 < 0 || n_iters <= 0 || it0 + n_iters > L->max_iter) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // A window with a row exchange is captured only when the exchange says it may be (`gather_capturable`: kernels only, nothing
    // of the call baked into their arguments -- the peer exchange reads its generation and stage parity from device memory since
    // round 6; a communicator call belongs to its library's own capture rules and stays out).
    if (L->gather && !L->gather_capturable) use_graph = 0;
    int it = it0;
    while (it < it0 + n_iters) {
        const int n = (it0 + n_iters - it < L->B) ? it0 + n_iters - it : L->B;
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)L->iter_base, it, 1, st);
        if (e != hipSuccess) return (int)e;
        int slot = -1;
        if (use_graph) {
            for (int i = 0; i < 2; ++i)
                if (L->graphs[i] && L->graph_len[i] == n) slot = i;
            if (slot < 0) {
                int free_slot = !L->graphs[0] ? 0 : (!L->graphs[1] ? 1 : -1);
                if (free_slot >= 0) {
                    hipGraph_t g = nullptr;
                    e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                    if (e != hipSuccess) {  // e.g. the legacy default stream cannot be captured: plain launches from now on
                        (void)hipGetLastError();
                        use_graph = 0;
                        const int rc0 = umap_loop_enqueue_window(L, n, st, it);
                        if (rc0 != TDR_OK) return rc0;
                        it += n;
                        continue;
                    }
                    const int rc = umap_loop_enqueue_window(L, n, st);
                    e = hipStreamEndCapture(st, &g);
                    if (rc != TDR_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
                    if (e != hipSuccess) return (int)e;
                    hipGraphExec_t ge = nullptr;
                    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(g);
                    if (e != hipSuccess) return (int)e;
                    L->graphs[free_slot] = ge; L->graph_len[free_slot] = n; slot = free_slot;
                }
            }
        }
        if (slot >= 0) {
            e = hipGraphLaunch(L->graphs[slot], st);
            if (e != hipSuccess) return (int)e;
        } else {
            const int rc = umap_loop_enqueue_window(L, n, st, it);
            if (rc != TDR_OK) return rc;
        }
        it += n;
    }
    return TDR_OK;
}

int tdr_umap_loop_destroy(void* loop) {
    UmapLoop* L = (UmapLoop*)loop;
    if (!L) return TDR_ERR_BAD_ARG;
    for (int i = 0; i < 2; ++i)
        if (L->graphs[i]) (void)hipGraphExecDestroy(L->graphs[i]);
    delete L;
    return TDR_OK;
}

}  // extern "C"
