// K5 / K6 / K9 -- the attraction-repulsion gradient loop of the neighbor-embedding methods.
//
// Replaces (citations under /root/reference/torchdr):
//   neighbor_embedding/umap.py:215-234   epochs_per_sample / epoch_of_next_sample set-up
//   neighbor_embedding/umap.py:236-292   UMAP closed-form attractive / repulsive gradients
//   neighbor_embedding/base.py:617-649   per-step negative sampling (torch.randint)
//   neighbor_embedding/largevis.py:181-201, tsne.py:162-180   losses whose autograd gradients are the
//                                        closed forms below
//   affinity_matcher.py:427-429          torch.optim.SGD(momentum) step
//
// Layout: embedding Z (N, NC) fp32 row-major, replicated per GPU; the affinity graph is CSR (UMAP) or the
// rectangular (n, k) kNN block (LargeVis / TSNE).  One row group of G lanes walks a row's edges with
// coalesced loads of (col, eps_per, next), gathers z_j from the L2 / Infinity-Cache resident Z, and
// reduces the NC-dimensional force with wave shuffles.  Negatives are generated in-kernel with a
// counter-based hash generator keyed by (seed, iteration, row, column), only for the 5*active columns the
// reference actually uses (it samples 150 and masks ~110 of them); an injected index table reproduces
// the reference's per-step arithmetic exactly for the parity tests.
#include "tdr_embed_common.h"

namespace tdr {

// ---- umap.py:215-234 ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ v, int64_t n, unsigned* __restrict__ amax_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));  // values are >= 0
}

__global__ __launch_bounds__(256) void umap_prepare_kernel(const float* __restrict__ v, int64_t n,
                                                           const unsigned* __restrict__ amax_bits, float max_iter,
                                                           float* __restrict__ eps_per, float* __restrict__ next) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float amax = __uint_as_float(*amax_bits);
    const float thr = amax / max_iter;
    const float a = v[i];
    float e = __fmul_rn(1.0f / __fadd_rn(a, 1e-3f), amax);
    if (a <= thr) e = __builtin_inff();
    eps_per[i] = e;
    next[i] = e;
}


struct UmapStepParams {
    const float* Z;          // (N, NC)
    int64_t n_total;         // N
    int64_t row0;            // first global row of this chunk
    int64_t n_rows;          // rows in this chunk
    const int64_t* rowptr;   // (n_rows + 1)
    const int32_t* cols;     // global column ids
    const float* eps_per;    // epochs_per_sample   (nnz)
    float* next;             // epoch_of_next_sample (nnz), updated in place
    float a, b;
    float t1;                // n_iter + 1
    int neg_rate;            // negative_sample_rate
    int n_negatives;         // int(neg_rate * n_neighbors)
    const int64_t* neg_inj;  // optional (n_rows, n_negatives) injected negatives, else the counter hash
    uint64_t seed;
    uint32_t iter;
    float exag, rep;         // early_exaggeration_coeff_, repulsion_strength
    float eps;               // 1e-3
    float* grad;             // (n_rows, NC)
    // sliced negative phase (large N): the positive pass stores the row's negative count and the slice passes
    // accumulate the repulsion in gr_acc before the clamp
    int32_t* nuse;           // (n_rows, 2) row headers {n_use | slice counts << 8.., hash key} or NULL = single pass
    float* gr_acc;           // (n_rows, NC)
    int64_t j_lo, j_hi;      // slice of negative indices handled by this pass
    int first, last;
    int slice, n_slices;     // this pass's slice index; dense passes need n_slices in {2, 4}
};


// Both loops run U group-widths per pass with every load issued before the first use: the dependent chain
// per row is rowptr -> {next, cols} -> z_j gather -> math (3 memory levels) and U random gathers are in
// flight per lane -- the kernel is bound by the latency of the random 8-byte reads of Z (8 MB at N = 1M,
// larger than one XCD's L2), so memory-level parallelism is what buys throughput.  `cols` is read for every
// edge (4 B) so that the gather does not wait for the activity test; eps_per only where the edge fires.
template <int NC, int G, int U, bool POS_ONLY = false>
__global__ __launch_bounds__(256) void umap_grad_kernel(const UmapStepParams P) {
    const int gl = threadIdx.x % G;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= P.n_rows) return;
    const int64_t gi = P.row0 + r;
    const int64_t e0 = P.rowptr[r], e1 = P.rowptr[r + 1];
    const Vec<NC> zi = load_z<NC>(P.Z, gi);
    const float two_ab = 2.0f * P.a * P.b;
    const float INF = __builtin_inff();

    float ga[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ga[c] = 0.f;
    int act = 0;
    for (int64_t base = e0; base < e1; base += U * G) {
        float nx[U];
        int32_t cj[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = base + u * G + gl;
            const bool v = e < e1;
            nx[u] = v ? P.next[e] : INF;
            cj[u] = v ? P.cols[e] : (int32_t)gi;
        }
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            on[u] = nx[u] <= P.t1;
            zj[u] = load_z<NC>(P.Z, on[u] ? cj[u] : (int32_t)gi);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (on[u]) {
                const int64_t e = base + u * G + gl;
                P.next[e] = nx[u] + P.eps_per[e];
                act++;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            if (on[u] && d > 0.f) {
                const float pb = fast_pow(d, P.b);
                const float coef = (pb * two_ab) * fast_rcp(d * (1.0f + P.a * pb));  // 2ab d^(b-1) / (1 + a d^b)
#pragma unroll
                for (int c = 0; c < NC; ++c) ga[c] += coef * df[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) ga[c] = group_sum<G>(ga[c]);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) act += __shfl_xor(act, o, 64);

    int n_use = act * P.neg_rate;
    if (n_use > P.n_negatives) n_use = P.n_negatives;
    if (POS_ONLY) {  // the negatives are evaluated by the slice passes (umap_neg_slice_kernel)
        if (gl == 0) {
            // row header for the negative passes: n_use and the counts of slices 0..2 packed in one word (<= 150 < 256
            // each; the last slice gets the rest), and the row's hash key -- computed here, where the VALU has slack
            // (this pass is HBM-bound), so that the VALU-bound dense passes do not redo it
            const uint32_t rkey = neg_row_key(P.seed, P.iter, gi);
            uint32_t info = (uint32_t)n_use;
            if (P.n_slices == 2 || P.n_slices == 4) {
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    if (sl < P.n_slices - 1) info |= (uint32_t)slice_count(rkey, n_use, sl, P.n_slices) << (8 * (sl + 1));
            }
            P.nuse[2 * r] = (int32_t)info;
            P.nuse[2 * r + 1] = (int32_t)rkey;
#pragma unroll
            for (int c = 0; c < NC; ++c) P.grad[(size_t)r * NC + c] = P.exag * fminf(fmaxf(ga[c], -4.f), 4.f);
        }
        return;
    }
    float gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = 0.f;
    const uint32_t rkey = neg_row_key(P.seed, P.iter, gi);
    const float m2b = -2.0f * P.b;
    for (int base = 0; base < n_use; base += U * G) {
        int64_t jn[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int col = base + u * G + gl;
            v[u] = col < n_use;
            jn[u] = gi;
            if (v[u]) jn[u] = P.neg_inj ? P.neg_inj[(size_t)r * P.n_negatives + col] : sample_negative(rkey, gi, col, P.n_total);
        }
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zj[u] = load_z<NC>(P.Z, jn[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            if (v[u]) {
                const float den = 1.0f + P.a * (d > 0.f ? fast_pow(d, P.b) : 0.f);
                const float coef = fast_rcp((d + P.eps) * den) * m2b;
#pragma unroll
                for (int c = 0; c < NC; ++c) gr[c] += coef * df[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = group_sum<G>(gr[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float a_ = fminf(fmaxf(ga[c], -4.f), 4.f);
            const float r_ = fminf(fmaxf(gr[c], -4.f), 4.f);
            P.grad[(size_t)r * NC + c] = P.exag * a_ + P.rep * r_;
        }
    }
}

// Negative phase, one slice of the index range per launch.  At N = 1M the embedding (8 MB) does not fit an XCD's
// 4 MB L2 and the ~43 uniformly random 8-byte gathers per row go to the fabric (random negatives cost 0.46 us per
// 1000 rows there, 0.19 when Z fits L2, 0.07 when they are sequential).  Restricted to a slice of Z the gathers are
// L2 hits.  Every pass regenerates the row's negatives from the counter hash (or reads the injected ones) and keeps
// those inside [j_lo, j_hi); the partial sums meet in gr_acc and the last pass applies the clamp.  A pass is VALU
// bound (hash + force math issue for every column slot whatever the number of live lanes: ~1050 issue cycles per
// wavefront, 0.14 ms per 1M rows), so few, large slices win: 2 slices at N = 1M (0.96 -> 0.72 ms in
// tools/umap_perf.py; 4 slices 1.02 ms).  Compacting the live items through LDS or pipelining rows in a persistent
// grid did not pay (measured: the compaction costs what it saves at 16 lanes per row).
template <int NC, int G, int U>
__global__ __launch_bounds__(256) void umap_neg_slice_kernel(const UmapStepParams P) {
    const int gl = threadIdx.x % G;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= P.n_rows) return;
    const int64_t gi = P.row0 + r;
    const Vec<NC> zi = load_z<NC>(P.Z, gi);
    const int n_use = P.nuse[2 * r] & 0xff;
    float gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = 0.f;
    const uint32_t rkey = neg_row_key(P.seed, P.iter, gi);
    const float m2b = -2.0f * P.b;
    for (int base = 0; base < n_use; base += U * G) {
        int64_t jn[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int col = base + u * G + gl;
            v[u] = col < n_use;
            jn[u] = gi;
            if (v[u]) jn[u] = P.neg_inj ? P.neg_inj[(size_t)r * P.n_negatives + col] : sample_negative(rkey, gi, col, P.n_total);
            v[u] = v[u] && jn[u] >= P.j_lo && jn[u] < P.j_hi;
        }
        Vec<NC> zj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            zj[u] = zi;
            if (v[u]) zj[u] = load_z<NC>(P.Z, jn[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float df[NC];
            const float d = sqdist<NC>(zi, zj[u], df);
            if (v[u]) {
                const float den = 1.0f + P.a * (d > 0.f ? fast_pow(d, P.b) : 0.f);
                const float coef = fast_rcp((d + P.eps) * den) * m2b;
#pragma unroll
                for (int c = 0; c < NC; ++c) gr[c] += coef * df[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = group_sum<G>(gr[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float tot = (P.first ? 0.f : P.gr_acc[(size_t)r * NC + c]) + gr[c];
            if (P.last) P.grad[(size_t)r * NC + c] += P.rep * fminf(fmaxf(tot, -4.f), 4.f);
            else P.gr_acc[(size_t)r * NC + c] = tot;
        }
    }
}

// ---- dense negative slice pass --------------------------------------------------------------------------------
// PMC on the masking pass above: 484 VALU instructions per wavefront, VALU busy 95 % -- it is VALU-issue bound, and
// half of its column slots (all but 1/S of them) are generated only to be masked out.  The dense pass draws exactly
// the row's share for THIS slice instead.  The share is an exact multinomial split: the number of a row's n
// negatives falling into the lower of two equal halves of the index range is the population count of n fair random
// bits (Binomial(n, 1/2)), applied once for 2 slices and twice for 4; inside its slice every negative is uniform.
// "Split the count, then draw uniformly inside the part" has the same distribution as n i.i.d. uniform draws from
// {0..N-1} minus the row itself (reference: r ~ U{0..N-2}, j = r + (r >= i); neighbor_embedding/base.py:628-636).
template <int NC, int G, int U>
__global__ __launch_bounds__(256) void umap_neg_dense_kernel(const UmapStepParams P) {
    const int gl = threadIdx.x % G;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= P.n_rows) return;
    const uint32_t gi = (uint32_t)(P.row0 + r);
    const Vec<NC> zi = load_z<NC>(P.Z, gi);
    const uint32_t info = (uint32_t)P.nuse[2 * r];
    const uint32_t rkey = (uint32_t)P.nuse[2 * r + 1];
    int n_cols;
    if (P.slice < P.n_slices - 1) n_cols = (int)((info >> (8 * (P.slice + 1))) & 0xffu);
    else {
        n_cols = (int)(info & 0xffu);
        for (int sl = 0; sl < P.n_slices - 1; ++sl) n_cols -= (int)((info >> (8 * (sl + 1))) & 0xffu);
    }
    // this slice of the reduced index range [0, N-1)
    const uint32_t nred = (uint32_t)(P.n_total - 1);
    const uint32_t step = (nred + (uint32_t)P.n_slices - 1u) / (uint32_t)P.n_slices;
    const uint32_t r_lo = (uint32_t)P.slice * step;
    const uint32_t r_len = (r_lo < nred) ? ((nred - r_lo < step) ? nred - r_lo : step) : 0u;
    const uint32_t ckey = rkey + 0x632BE5ABu * (uint32_t)(P.slice + 1) + (uint32_t)gl * 0x9E3779B9u;
    const float m2b = -2.0f * P.b;
    float gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = 0.f;
    if (r_len) {
        for (int base = 0; base < n_cols; base += U * G) {
            uint32_t jn[U];
            bool v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int col = base + u * G + gl;
                v[u] = col < n_cols;
                const uint32_t x = mix32_item(ckey + (uint32_t)(base + u * G) * 0x9E3779B9u);  // = slice_negative()
                const uint32_t rr = r_lo + __umulhi(x, r_len);
                jn[u] = v[u] ? rr + (rr >= gi ? 1u : 0u) : gi;
            }
            Vec<NC> zj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) zj[u] = load_z<NC>(P.Z, jn[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float df[NC];
                const float d = sqdist<NC>(zi, zj[u], df);
                if (v[u]) {
                    const float den = 1.0f + P.a * (d > 0.f ? fast_pow(d, P.b) : 0.f);
                    const float coef = fast_rcp((d + P.eps) * den) * m2b;
#pragma unroll
                    for (int c = 0; c < NC; ++c) gr[c] += coef * df[c];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = group_sum<G>(gr[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float tot = (P.first ? 0.f : P.gr_acc[(size_t)r * NC + c]) + gr[c];
            if (P.last) P.grad[(size_t)r * NC + c] += P.rep * fminf(fmaxf(tot, -4.f), 4.f);
            else P.gr_acc[(size_t)r * NC + c] = tot;
        }
    }
}

// Test hook: the negatives the dense slice passes draw for each row (same device functions and hashing), written slice
// after slice into out (n_rows, width); unused slots = -1.  Lets the tests check the sampler's distribution.
__global__ __launch_bounds__(256) void umap_debug_negatives_kernel(uint64_t seed, uint32_t iter, int64_t n_total, int64_t row0,
                                                                   int64_t n_rows, const int32_t* __restrict__ nuse,
                                                                   int n_slices, int width, int64_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t gi = (uint32_t)(row0 + r);
    const uint32_t rkey = neg_row_key(seed, iter, (int64_t)gi);
    const uint32_t nred = (uint32_t)(n_total - 1);
    int pos = 0;
    for (int sl = 0; sl < n_slices; ++sl) {
        const int cnt = slice_count(rkey, nuse[r], sl, n_slices);
        for (int col = 0; col < cnt && pos < width; ++col)
            out[(size_t)r * width + pos++] = (int64_t)slice_negative(rkey, gi, col, sl, n_slices, nred);
    }
    for (; pos < width; ++pos) out[(size_t)r * width + pos] = -1;
}

// ---- LargeVis / TSNE sparse terms -------------------------------------------------------------------
struct NeStepParams {
    const float* Z;
    int64_t n_total, row0, n_rows;
    const int32_t* nn;       // (n_rows, k)
    const float* P;          // (n_rows, k) affinities (not log)
    int k;
    int kind;                // 0 largevis, 1 tsne (attraction), 2 sne (attraction), 3 infotsne
    float exag;              // multiplies the attractive term
    float rep_coef;          // largevis: repulsion_strength * 2 / N
    int n_neg;               // negatives per row (0 = none)
    const int64_t* neg_inj;  // optional (n_rows, n_neg)
    uint64_t seed;
    uint32_t iter;
    float* grad;             // (N, NC) zero-initialised; both endpoints receive atomics
    // optional transposed graph (in-edges of this chunk's rows): when present the neighbour edges are
    // evaluated pull-style by BOTH endpoints' rows and no atomics are issued for them
    const int64_t* t_rowptr; // (n_rows + 1) or NULL
    const int32_t* t_src;    // global source row of each in-edge
    const float* t_val;      // P of each in-edge
    int nc;                  // row width of Z / grad (PAD instances)
    int perm_neg;            // 1: negatives are keyed permutations of the rows and BOTH shares of every pair are pulled
                             // (tdr_embed_common.h: permutation sampler); kinds 0 and 3, no injected table
    const float* rowsum;     // kind 3 with perm_neg: (n_total) row normalisers sum_n q of EVERY row (ne_rowsum_kernel)
    int neg_halves;          // perm_neg, many negatives per row (InfoTSNE: 2 x 300 items): ONE launch in which every row is visited
                             // by two workgroups, one per half of the index range; a visit takes the row's negative items whose
                             // OTHER endpoint lies in its half (and, when the row itself does, its neighbour edges).  Workgroups
                             // go round the 8 XCDs, XCDs 0-3 take the lower half, 4-7 the upper: an XCD gathers from half of
                             // the embedding and of the normaliser table (6 MB instead of 12 MB at N = 1M against 4 MB of L2).
                             // The two partial sums of a row meet by atomic adds onto zero: 0 + a + b in either order.
};

// the half of the index range a workgroup of a two-half launch serves, and its row block
__device__ __forceinline__ void ne_half_of_block(const NeStepParams& S, int64_t& blk, int64_t& j_lo, int64_t& j_hi) {
    blk = blockIdx.x; j_lo = 0; j_hi = S.n_total;
    if (S.neg_halves == 2) {
        const int x = (int)(blockIdx.x & 7u);
        blk = (int64_t)(blockIdx.x >> 3) * 4 + (x & 3);
        const int64_t half = (S.n_total + 1) >> 1;
        j_lo = (x >> 2) ? half : 0;
        j_hi = (x >> 2) ? S.n_total : half;
    }
}

// InfoTSNE with the permutation sampler, pass 1: rowsum[i] = sum over row i's own draws of q = 1 / (1 + d)
template <int NC, int G, bool PAD = false>
__global__ __launch_bounds__(256) void ne_rowsum_kernel(const NeStepParams S, float* __restrict__ out) {
    const int nc = PAD ? S.nc : NC;
    const int gl = threadIdx.x % G;
    int64_t blk, j_lo, j_hi;
    ne_half_of_block(S, blk, j_lo, j_hi);
    const int64_t r = (blk * 256 + threadIdx.x) / G;
    if (r >= S.n_rows) return;
    const int64_t gi = S.row0 + r;
    const Vec<NC> zi = load_zp<NC, PAD>(S.Z, gi, nc);
    float s = 0.f;
    for (int col = gl; col < S.n_neg; col += G) {
        const PermKey K = perm_key(S.seed, S.iter, col, S.n_total);
        const uint32_t j = perm_succ(perm_inv((uint32_t)gi, K), K);     // never the row itself
        if ((int64_t)j < j_lo || (int64_t)j >= j_hi) continue;
        const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const float t = zi.v[c] - zj.v[c]; d += t * t; }
        s += 1.0f / (1.0f + d);
    }
    s = group_sum<G>(s);
    if (gl == 0) {
        if (S.neg_halves == 2) unsafeAtomicAdd(&out[gi], s);     // onto zero: the two visits' sums in either order
        else out[gi] = s;
    }
}

template <int NC, int G, bool PAD = false>
__global__ __launch_bounds__(256) void ne_grad_kernel(const NeStepParams S) {
    const int nc = PAD ? S.nc : NC;  // row width in memory (PAD: NC is the padded register width)
    const int gl = threadIdx.x % G;
    int64_t blk, j_lo, j_hi;
    ne_half_of_block(S, blk, j_lo, j_hi);
    const int64_t r = (blk * 256 + threadIdx.x) / G;
    if (r >= S.n_rows) return;
    const int64_t gi = S.row0 + r;
    const bool own_half = gi >= j_lo && gi < j_hi;      // this visit also takes the row's neighbour edges
    const Vec<NC> zi = load_zp<NC, PAD>(S.Z, gi, nc);
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    const float off = (S.kind == 0) ? 2.0f : 1.0f;
    const bool gauss = S.kind == 2;  // SNE: log Q = -d, the edge weight has no denominator
    const bool pull = S.t_rowptr != nullptr;
    // out-edges i -> j : +w (z_i - z_j) on i, and -w (z_i - z_j) on j (pushed with atomics unless j's own row
    // pulls it from the transposed graph)
    for (int p = gl; own_half && p < S.k; p += G) {
        const int64_t j = S.nn[(size_t)r * S.k + p];
        const float pij = S.P[(size_t)r * S.k + p];
        const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
        float df[NC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const float w = S.exag * 2.0f * pij * (gauss ? 1.0f : 1.0f / (off + d));
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float t = w * df[c];
            g[c] += t;
            if (!pull && c < nc) unsafeAtomicAdd(&S.grad[(size_t)j * nc + c], -t);
        }
    }
    if (pull && own_half) {
        // in-edges s -> i carry -w (z_s - z_i) = +w (z_i - z_s): the same expression as an out-edge
        const int64_t e1 = S.t_rowptr[r + 1];
        for (int64_t e = S.t_rowptr[r] + gl; e < e1; e += G) {
            const Vec<NC> zs = load_zp<NC, PAD>(S.Z, S.t_src[e], nc);
            float df[NC];
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zs.v[c]; d += df[c] * df[c]; }
            const float w = S.exag * 2.0f * S.t_val[e] * (gauss ? 1.0f : 1.0f / (off + d));
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] += w * df[c];
        }
    }
    if (S.perm_neg) {
        // permutation sampler: item 2c = this row's own draw of column c, item 2c + 1 = the row whose draw of column c hit
        // this row; both are  +w (z_i - z_other)  on THIS row -- nothing is sent to the other endpoint
        const float own_inv = (S.kind == 3) ? 1.0f / S.rowsum[gi] : 0.f;
        if (2 * S.n_neg <= G) {
            // few negatives per row (LargeVis: 5): one (column, side) ITEM per lane -- 10 of the 16 lanes each issue one gather, instead
            // of 5 lanes issuing two in a row (the launch is bound by the per-row dependency chain, profiles/r05_c3_pmc.json)
            if (gl < 2 * S.n_neg) {
                const int col = gl >> 1, side = gl & 1;
                const PermKey K = perm_key(S.seed, S.iter, col, S.n_total);
                const uint32_t a = perm_inv((uint32_t)gi, K);
                const uint32_t j = side ? perm_pred(a, K) : perm_succ(a, K);
                if ((int64_t)j >= j_lo && (int64_t)j < j_hi) {
                    const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
                    float df[NC];
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
                    float w;
                    if (S.kind == 3) {
                        const float q = 1.0f / (1.0f + d);
                        w = -S.rep_coef * q * q * (side ? 1.0f / S.rowsum[j] : own_inv);
                    } else {
                        w = -S.rep_coef / ((1.0f + d) * (2.0f + d));
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) g[c] += w * df[c];
                }
            }
        } else
        for (int col = gl; col < S.n_neg; col += G) {
            const PermKey K = perm_key(S.seed, S.iter, col, S.n_total);
            const uint32_t a = perm_inv((uint32_t)gi, K);       // this row's place in the column's cyclic order
            const uint32_t jj[2] = {perm_succ(a, K), perm_pred(a, K)};   // its own draw | the row that drew it
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const uint32_t j = jj[side];
                if ((int64_t)j < j_lo || (int64_t)j >= j_hi) continue;
                const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
                float df[NC];
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
                float w;
                if (S.kind == 3) {
                    const float q = 1.0f / (1.0f + d);
                    w = -S.rep_coef * q * q * (side ? 1.0f / S.rowsum[j] : own_inv);   // the DRAWING row's normaliser
                } else {
                    w = -S.rep_coef / ((1.0f + d) * (2.0f + d));
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) g[c] += w * df[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            g[c] = group_sum<G>(g[c]);
            if (gl == 0 && c < nc) unsafeAtomicAdd(&S.grad[(size_t)gi * nc + c], g[c]);
        }
        return;
    }
    const uint32_t rkey = neg_row_key(S.seed, S.iter, gi);
    // InfoTSNE (kind 3): the repulsion is the row's log-sum over its negatives of q = 1/(1+d); its
    // derivative weights each negative by q^2 / sum_n q -> one extra pass for the row normaliser
    float inv_rowsum = 0.f;
    if (S.kind == 3) {
        float s = 0.f;
        for (int col = gl; col < S.n_neg; col += G) {
            int64_t j;
            if (S.neg_inj) j = S.neg_inj[(size_t)r * S.n_neg + col];
            else j = sample_negative(rkey, gi, col, S.n_total);
            const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const float t = zi.v[c] - zj.v[c]; d += t * t; }
            s += 1.0f / (1.0f + d);
        }
        inv_rowsum = 1.0f / group_sum<G>(s);
    }
    for (int col = gl; col < S.n_neg; col += G) {
        int64_t j;
        if (S.neg_inj) j = S.neg_inj[(size_t)r * S.n_neg + col];
        else j = sample_negative(rkey, gi, col, S.n_total);
        const Vec<NC> zj = load_zp<NC, PAD>(S.Z, j, nc);
        float df[NC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        float w;
        if (S.kind == 3) {
            const float q = 1.0f / (1.0f + d);
            w = -S.rep_coef * q * q * inv_rowsum;
        } else {
            w = -S.rep_coef / ((1.0f + d) * (2.0f + d));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float t = w * df[c];
            g[c] += t;
            if (c < nc) unsafeAtomicAdd(&S.grad[(size_t)j * nc + c], -t);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum<G>(g[c]);
        if (gl == 0 && c < nc) unsafeAtomicAdd(&S.grad[(size_t)gi * nc + c], g[c]);
    }
}

// ---- LargeVis pull form, FOUR lanes per row (round 6) -----------------------------------------------------------------
// ne_grad_kernel<NC, 16> with the permutation sampler is bound by its dependency chains, not by bytes or instructions
// (profiles/r05_c3_pmc.json: every SIMD full of wavefronts that each live ~7 us, 96 G L2 requests/s of a 268 G/s ceiling, 89 vector
// instructions per row): a lane holds ONE edge, so a wavefront has 64 gathers in flight for 4 rows and walks row pointer -> source
// id -> gather one round trip at a time.  Here a row is worked by 4 lanes (16 rows per wavefront) and a lane issues everything it
// will need before it uses anything: 4 neighbour ids + weights, the in-edge range, then -- while those are in flight -- the keyed
// permutation arithmetic of its <= 4 negative items and their gathers, the in-edge source ids, the out-edge gathers, the in-edge
// gathers: up to 12 gathers per lane behind three index loads.  Same terms and formulas as ne_grad_kernel; a row's sum is taken
// lane by lane and then over the 4 lanes (another association of the same fp32 terms).  Kind 0 (LargeVis) with n_neg <= 8,
// 2 / 3 components, no halves; each row's gradient is STORED (every row is written exactly once: nothing is scattered).
template <int NC>
__global__ __launch_bounds__(256) void ne_pull4_kernel(const NeStepParams S) {
    constexpr int G = 4, U = 4, UN = 4;
    const int gl = threadIdx.x & (G - 1);
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= S.n_rows) return;
    const int64_t gi = S.row0 + r;
    const int k = S.k;
    // 1. index loads of the first batches
    int32_t jo[U];
    float po[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = gl + G * u;
        jo[u] = p < k ? S.nn[(size_t)r * k + p] : (int32_t)gi;
        po[u] = p < k ? S.P[(size_t)r * k + p] : 0.f;
    }
    const int64_t e0 = S.t_rowptr[r], e1 = S.t_rowptr[r + 1];
    const Vec<NC> zi = load_z<NC>(S.Z, gi);
    // 2. negative items of this lane: item = (column, side); side 0 = the row's own draw, 1 = the row whose draw hit it
    const int n_items = 2 * S.n_neg;
    uint32_t jn[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int it = gl + G * u;
        jn[u] = (uint32_t)gi;
        if (it < n_items) {
            const PermKey K = perm_key(S.seed, S.iter, it >> 1, S.n_total);
            const uint32_t a = perm_inv((uint32_t)gi, K);
            jn[u] = (it & 1) ? perm_pred(a, K) : perm_succ(a, K);
        }
    }
    Vec<NC> zn[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) zn[u] = load_z<NC>(S.Z, jn[u]);
    // 3. in-edge ids of the first batch, out-edge gathers, in-edge gathers
    int32_t js[U];
    float ps[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + gl + G * u;
        js[u] = e < e1 ? S.t_src[e] : (int32_t)gi;
        ps[u] = e < e1 ? S.t_val[e] : 0.f;
    }
    Vec<NC> zo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) zo[u] = load_z<NC>(S.Z, jo[u]);
    Vec<NC> zs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) zs[u] = load_z<NC>(S.Z, js[u]);
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    // negatives: -rep_coef / ((1 + d)(2 + d)) (largevis.py:181-201); an unused slot holds the row itself (zero difference)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        float df[NC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zn[u].v[c]; d += df[c] * df[c]; }
        const float w = -S.rep_coef / ((1.0f + d) * (2.0f + d));
#pragma unroll
        for (int c = 0; c < NC; ++c) g[c] += w * df[c];
    }
    // out-edges (weight 0 beyond the row's k), further batches for k > 16
    auto edge = [&](const Vec<NC>& zj, float pij) {
        float df[NC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const float w = S.exag * 2.0f * pij * (1.0f / (2.0f + d));
#pragma unroll
        for (int c = 0; c < NC; ++c) g[c] += w * df[c];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) edge(zo[u], po[u]);
    for (int p0 = G * U; p0 < k; p0 += G * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + gl + G * u;
            jo[u] = p < k ? S.nn[(size_t)r * k + p] : (int32_t)gi;
            po[u] = p < k ? S.P[(size_t)r * k + p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) zo[u] = load_z<NC>(S.Z, jo[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) edge(zo[u], po[u]);
    }
    // in-edges s -> i: the same expression (see ne_grad_kernel)
#pragma unroll
    for (int u = 0; u < U; ++u) edge(zs[u], ps[u]);
    for (int64_t eb = e0 + G * U; eb < e1; eb += G * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + gl + G * u;
            js[u] = e < e1 ? S.t_src[e] : (int32_t)gi;
            ps[u] = e < e1 ? S.t_val[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) zs[u] = load_z<NC>(S.Z, js[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) edge(zs[u], ps[u]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum<G>(g[c]);
        if (gl == 0) S.grad[(size_t)gi * NC + c] = g[c];
    }
}

// ---- LargeVis pull form with the negatives served from LDS: the RUN-PERMUTATION sampler (round 6) ----------------------------
// ne_pull4_kernel still issues 40 divergent 8-byte gathers per row (15 out-edges, ~15 in-edges, 2 x 5 negative items) and the L1
// serves those at ~0.44 lanes per clock and CU (tools/gather_bench.hip): 40 M lanes are >= 169 us at N = 1M whatever the kernel does
// with them -- 0.35 of the HBM roofline on SURVEY 8d's bytes is the CEILING of any formulation that gathers every negative.  The
// negatives are the part whose position nobody prescribes.  Here the keyed cyclic order of the permutation sampler runs over RUNS of
// 16 consecutive rows instead of rows: in column c of iteration t, run a draws run succ_c(a), row (a, o) draws row (succ_c(a),
// (o + delta) & 15) with delta hashed from (column key, a), and the row that drew (a, o) is (pred_c(a), (o - delta') & 15) with
// delta' the shift of pred_c(a).  A workgroup of 64 rows = 4 runs therefore needs, per column, 4 + 4 runs of 128 bytes (nc = 2):
// 40 coalesced line reads per 64 rows staged into LDS instead of 640 gathers, and a lane reads its negatives from LDS.
//   * per-row law: a row's draw is uniform over the rows OUTSIDE its own run (the run order is a keyed pseudo-random cyclic order,
//     the offset a hashed rotation): uniform over N - 16 of the reference's N - 1 candidates (base.py:628-636), never the row itself,
//     independent across columns and iterations;
//   * across rows: every row is the far endpoint of exactly n_neg pairs (as with the row permutation); the 16 rows of a run draw the
//     16 rows of ONE other run (a rotation of them), where the row permutation sends them to 16 unrelated rows;
//   * N not a multiple of 16: pairs with an endpoint in the padding of the last run do not exist (both of its shares are dropped:
//     up to 15 rows lose one pair per column and side).
// Both shares of every pair are still PULLED (nothing is scattered), the neighbour edges are ne_pull4_kernel's.
constexpr int RUNP_LEN = 16;
struct RunSlot { uint32_t run, shift; };
// side 0: the run that run `a` draws in column `col`; side 1: the run that draws run `a`; shift = the drawing run's offset rotation
__device__ __forceinline__ RunSlot runperm_slot(uint64_t seed, uint32_t iter, int col, uint32_t a, int side, uint32_t n_runs) {
    const PermKey K = perm_key(seed, iter, col, (int64_t)n_runs);
    const uint32_t pa = perm_inv(a, K);
    RunSlot sl;
    sl.run = side ? perm_pred(pa, K) : perm_succ(pa, K);
    const uint32_t drawer = side ? sl.run : a;
    sl.shift = mix32(K.a2 ^ (drawer * 0x9E3779B1u)) & (uint32_t)(RUNP_LEN - 1);
    return sl;
}
__device__ __forceinline__ uint32_t runperm_offset(uint32_t o, uint32_t shift, int side) {
    return (side ? o - shift : o + shift) & (uint32_t)(RUNP_LEN - 1);
}

__global__ __launch_bounds__(256) void runperm_debug_kernel(uint64_t seed, uint32_t iter, int64_t n_total, int n_neg,
                                                            int64_t* __restrict__ fwd, int64_t* __restrict__ inv) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_total * n_neg) return;
    const int64_t i = e / n_neg;
    const int c = (int)(e - i * n_neg);
    const uint32_t n_runs = (uint32_t)((n_total + RUNP_LEN - 1) / RUNP_LEN);
    const uint32_t a = (uint32_t)(i / RUNP_LEN), o = (uint32_t)(i % RUNP_LEN);
    const RunSlot s0 = runperm_slot(seed, iter, c, a, 0, n_runs), s1 = runperm_slot(seed, iter, c, a, 1, n_runs);
    const int64_t j0 = (int64_t)s0.run * RUNP_LEN + runperm_offset(o, s0.shift, 0);
    const int64_t j1 = (int64_t)s1.run * RUNP_LEN + runperm_offset(o, s1.shift, 1);
    fwd[e] = j0 < n_total ? j0 : -1;
    inv[e] = j1 < n_total ? j1 : -1;
}

template <int NC>
__global__ __launch_bounds__(256) void ne_pull4_runs_kernel(const NeStepParams S) {
    constexpr int G = 4, U = 4, UN = 4, RL = RUNP_LEN, RPW = 64 / RL, SLOTS = RPW * 16, PPR = RL * NC / 4;
    __shared__ uint32_t s_run[SLOTS], s_shift[SLOTS];
    __shared__ __attribute__((aligned(16))) float buf[SLOTS * RL * NC];
    const int tid = threadIdx.x, gl = tid & (G - 1), rl = tid >> 2;      // rl: row of the workgroup (0..63)
    // a workgroup owns a GLOBAL block of 64 rows = 4 runs (the same partition whatever the row sharding: a rank's launch covers
    // the blocks its chunk touches and leaves the rows outside the chunk idle)
    const int64_t gblk = S.row0 / 64 + (int64_t)blockIdx.x;
    const int64_t r = gblk * 64 + rl - S.row0;                           // local row of the launch
    const bool active = r >= 0 && r < S.n_rows;
    const int64_t gi = active ? S.row0 + r : S.row0;                     // idle lanes shadow row 0 of the launch and store nothing
    const int k = S.k;
    const int n_items = 2 * S.n_neg;
    const uint32_t n_runs = (uint32_t)((S.n_total + RL - 1) / RL);
    // 1. index loads of the first batches (as ne_pull4_kernel)
    int32_t jo[U];
    float po[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = gl + G * u;
        jo[u] = (active && p < k) ? S.nn[(size_t)r * k + p] : (int32_t)gi;
        po[u] = (active && p < k) ? S.P[(size_t)r * k + p] : 0.f;
    }
    const int64_t e0 = active ? S.t_rowptr[r] : 0, e1 = active ? S.t_rowptr[r + 1] : 0;
    const Vec<NC> zi = load_z<NC>(S.Z, gi);
    // 2. the runs this workgroup's 4 runs draw / are drawn by: one thread per (run, item)
    if (tid < SLOTS) {
        const uint32_t a = (uint32_t)gblk * RPW + (uint32_t)(tid >> 4);
        const int item = tid & 15;
        uint32_t run = 0xffffffffu, shift = 0u;
        if (item < n_items && a < n_runs) {
            const RunSlot sl = runperm_slot(S.seed, S.iter, item >> 1, a, item & 1, n_runs);
            run = sl.run; shift = sl.shift;
        }
        s_run[tid] = run; s_shift[tid] = shift;
    }
    int32_t js[U];
    float ps[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + gl + G * u;
        js[u] = e < e1 ? S.t_src[e] : (int32_t)gi;
        ps[u] = e < e1 ? S.t_val[e] : 0.f;
    }
    Vec<NC> zo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) zo[u] = load_z<NC>(S.Z, jo[u]);
    __syncthreads();
    // 3. stage the runs: 16-byte pieces, coalesced (a run is RL * NC * 4 bytes, 16-byte aligned)
    const int64_t z_floats = S.n_total * NC;
    for (int p = tid; p < SLOTS * PPR; p += 256) {
        const int slot = p / PPR, piece = p - slot * PPR;
        const uint32_t run = s_run[slot];
        if (run == 0xffffffffu) continue;
        const int64_t off = (int64_t)run * (RL * NC) + piece * 4;
        float4 v;
        if (off + 4 <= z_floats) v = *reinterpret_cast<const float4*>(S.Z + off);
        else {
            v.x = off < z_floats ? S.Z[off] : 0.f; v.y = off + 1 < z_floats ? S.Z[off + 1] : 0.f;
            v.z = off + 2 < z_floats ? S.Z[off + 2] : 0.f; v.w = 0.f;
        }
        *reinterpret_cast<float4*>(buf + (size_t)slot * (RL * NC) + piece * 4) = v;
    }
    Vec<NC> zs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) zs[u] = load_z<NC>(S.Z, js[u]);
    __syncthreads();
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    // 4. negatives from LDS: item = (column, side); a pair whose other endpoint is padding does not exist
    {
        const int a_local = rl >> 4;
        const uint32_t o = (uint32_t)(rl & 15);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int it = gl + G * u;
            if (it < n_items && active) {
                const int slot = a_local * 16 + it;
                const uint32_t run = s_run[slot];
                const uint32_t o2 = runperm_offset(o, s_shift[slot], it & 1);
                if ((int64_t)run * RL + o2 < S.n_total) {
                    const float* q = buf + (size_t)slot * (RL * NC) + o2 * NC;
                    float df[NC];
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - q[c]; d += df[c] * df[c]; }
                    const float w = -S.rep_coef / ((1.0f + d) * (2.0f + d));
#pragma unroll
                    for (int c = 0; c < NC; ++c) g[c] += w * df[c];
                }
            }
        }
    }
    auto edge = [&](const Vec<NC>& zj, float pij) {
        float df[NC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const float w = S.exag * 2.0f * pij * (1.0f / (2.0f + d));
#pragma unroll
        for (int c = 0; c < NC; ++c) g[c] += w * df[c];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) edge(zo[u], po[u]);
    for (int p0 = G * U; active && p0 < k; p0 += G * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + gl + G * u;
            jo[u] = p < k ? S.nn[(size_t)r * k + p] : (int32_t)gi;
            po[u] = p < k ? S.P[(size_t)r * k + p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) zo[u] = load_z<NC>(S.Z, jo[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) edge(zo[u], po[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) edge(zs[u], ps[u]);
    for (int64_t eb = e0 + G * U; eb < e1; eb += G * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + gl + G * u;
            js[u] = e < e1 ? S.t_src[e] : (int32_t)gi;
            ps[u] = e < e1 ? S.t_val[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) zs[u] = load_z<NC>(S.Z, js[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) edge(zs[u], ps[u]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum<G>(g[c]);
        if (gl == 0 && active) S.grad[(size_t)r * NC + c] = g[c];
    }
}

// ---- TSNE dense repulsion (tsne.py:172-180): S = sum_ij w_ij, F_i = sum_j (z_i - z_j) w_ij^2 ---------
// The 256 columns of a tile are staged pair-interleaved (x0 x1 | y0 y1 | ..) and two columns are evaluated at once with
// packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two results per lane and issue slot); the weights come from
// v_rcp_f32 (1 ulp) instead of an IEEE division (~10 instructions): ~34 issue cycles per pair instead of ~90.  Even and odd
// columns accumulate in the two halves of packed accumulators, added at the end (a fixed association: it depends on the
// column index only, so sharded and single-process runs still produce the same bits).
typedef float ne_f32x2 __attribute__((ext_vector_type(2)));

template <int NC, bool PAD = false>
__global__ __launch_bounds__(256) void tsne_repulsion_kernel(const float* __restrict__ Z, int64_t n_total, int64_t row0,
                                                             int64_t n_rows, float* __restrict__ F, double* __restrict__ S, int nc_,
                                                             int64_t cols_per_seg) {
    // blockIdx.y = column segment (tdr_tsne_repulsion_split_f32): this workgroup sums over columns [j_lo, j_hi) and writes
    // its rows' partial forces to plane blockIdx.y of F (n_seg planes of n_rows x nc; one plane = the result when unsplit)
    const int64_t j_lo = (int64_t)blockIdx.y * cols_per_seg;
    const int64_t j_hi = (j_lo + cols_per_seg < n_total) ? j_lo + cols_per_seg : n_total;
    F += (size_t)blockIdx.y * n_rows * (PAD ? nc_ : NC);
    const int nc = PAD ? nc_ : NC;
    __shared__ __attribute__((aligned(16))) float tile[256 * NC];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    ne_f32x2 zz[NC], f2[NC];
    float zs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        zs[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.f;
        zz[c] = ne_f32x2{zs[c], zs[c]};
        f2[c] = ne_f32x2{0.f, 0.f};
    }
    ne_f32x2 s2 = ne_f32x2{0.f, 0.f};
    for (int64_t j0 = j_lo; j0 < j_hi; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
        float* rec = tile + (threadIdx.x >> 1) * (2 * NC) + (threadIdx.x & 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) rec[2 * c] = (j < j_hi && c < nc) ? Z[(size_t)j * nc + c] : 0.f;
        __syncthreads();
        const int lim = (int)((j_hi - j0 < 256) ? (j_hi - j0) : 256);
        const int pairs = lim >> 1;
        for (int p = 0; p < pairs; ++p) {
            const ne_f32x2* q = reinterpret_cast<const ne_f32x2*>(tile + p * (2 * NC));
            ne_f32x2 df[NC];
            df[0] = zz[0] - q[0];
            ne_f32x2 d = df[0] * df[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) { df[c] = zz[c] - q[c]; d = __builtin_elementwise_fma(df[c], df[c], d); }
            d = d + 1.0f;
            const ne_f32x2 w = ne_f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            s2 = s2 + w;
            const ne_f32x2 w2 = w * w;
#pragma unroll
            for (int c = 0; c < NC; ++c) f2[c] = __builtin_elementwise_fma(w2, df[c], f2[c]);
        }
        if (lim & 1) {      // the last column of a ragged final tile
            const float* q = tile + pairs * (2 * NC);
            float df[NC];
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zs[c] - q[2 * c]; d = fmaf(df[c], df[c], d); }
            const float w = __builtin_amdgcn_rcpf(1.0f + d);
            s2.x += w;
            const float w2 = w * w;
#pragma unroll
            for (int c = 0; c < NC; ++c) f2[c].x = fmaf(w2, df[c], f2[c].x);
        }
    }
    float s = s2.x + s2.y;
    if (have) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < nc) F[(size_t)r * nc + c] = f2[c].x + f2[c].y;
    } else s = 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(S, (double)s);
}

// out[i] = planes[0][i] + planes[1][i] + ... in plane order
__global__ __launch_bounds__(256) void sum_planes_kernel(const float* __restrict__ planes, int n_planes, int64_t cnt,
                                                         float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    float a = planes[i];
    for (int p = 1; p < n_planes; ++p) a += planes[(size_t)p * cnt + i];
    out[i] = a;
}

// grad[row0 + r] += coef / S * F[r]
__global__ __launch_bounds__(256) void add_scaled_kernel(float* __restrict__ grad, const float* __restrict__ F,
                                                         const double* __restrict__ S, float coef, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float sc = coef / (float)(*S);
    grad[i] += sc * F[i];
}

// ---- SNE dense repulsion (sne.py:170-179): (1/N) sum_i log sum_j exp(-d_ij), diagonal included -------
// pass 1: R_i = sum_j exp(-d_ij) for the rows of this chunk
template <int NC, bool PAD = false>
__global__ __launch_bounds__(256) void sne_rowsum_kernel(const float* __restrict__ Z, int64_t n_total, int64_t row0,
                                                         int64_t n_rows, float* __restrict__ R, int nc_) {
    const int nc = PAD ? nc_ : NC;
    __shared__ __attribute__((aligned(16))) float tile[256 * NC];     // pair-interleaved, as in tsne_repulsion_kernel
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    ne_f32x2 zz[NC];
    float zs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        zs[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.f;
        zz[c] = ne_f32x2{zs[c], zs[c]};
    }
    ne_f32x2 s2 = ne_f32x2{0.f, 0.f};
    for (int64_t j0 = 0; j0 < n_total; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
        float* rec = tile + (threadIdx.x >> 1) * (2 * NC) + (threadIdx.x & 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) rec[2 * c] = (j < n_total && c < nc) ? Z[(size_t)j * nc + c] : 0.f;
        __syncthreads();
        const int lim = (int)((n_total - j0 < 256) ? (n_total - j0) : 256);
        const int pairs = lim >> 1;
        for (int p = 0; p < pairs; ++p) {
            const ne_f32x2* q = reinterpret_cast<const ne_f32x2*>(tile + p * (2 * NC));
            ne_f32x2 df = zz[0] - q[0];
            ne_f32x2 d = df * df;
#pragma unroll
            for (int c = 1; c < NC; ++c) { df = zz[c] - q[c]; d = __builtin_elementwise_fma(df, df, d); }
            s2 = s2 + ne_f32x2{__expf(-d.x), __expf(-d.y)};
        }
        if (lim & 1) {
            const float* q = tile + pairs * (2 * NC);
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const float u = zs[c] - q[2 * c]; d = fmaf(u, u, d); }
            s2.x += __expf(-d);
        }
    }
    if (have) R[r] = s2.x + s2.y;
}

// pass 2: grad_i += coef * sum_j exp(-d_ij) (1/R_i + 1/R_j) (z_i - z_j)   (R: all n_total rows)
template <int NC, bool PAD = false>
__global__ __launch_bounds__(256) void sne_repulsion_kernel(const float* __restrict__ Z, int64_t n_total, int64_t row0,
                                                            int64_t n_rows, const float* __restrict__ R, float coef,
                                                            float* __restrict__ grad, int nc_) {
    const int nc = PAD ? nc_ : NC;
    __shared__ __attribute__((aligned(16))) float tile[256 * (NC + 1)];   // pair-interleaved: coordinates, then 1 / R
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    ne_f32x2 zz[NC], f2[NC];
    float zs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        zs[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.f;
        zz[c] = ne_f32x2{zs[c], zs[c]};
        f2[c] = ne_f32x2{0.f, 0.f};
    }
    const float inv_ri = have ? 1.0f / R[row0 + r] : 0.f;
    const ne_f32x2 inv_ri2 = ne_f32x2{inv_ri, inv_ri};
    for (int64_t j0 = 0; j0 < n_total; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
        float* rec = tile + (threadIdx.x >> 1) * (2 * (NC + 1)) + (threadIdx.x & 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) rec[2 * c] = (j < n_total && c < nc) ? Z[(size_t)j * nc + c] : 0.f;
        rec[2 * NC] = (j < n_total) ? 1.0f / R[j] : 0.f;
        __syncthreads();
        const int lim = (int)((n_total - j0 < 256) ? (n_total - j0) : 256);
        const int pairs = lim >> 1;
        for (int p = 0; p < pairs; ++p) {
            const ne_f32x2* q = reinterpret_cast<const ne_f32x2*>(tile + p * (2 * (NC + 1)));
            ne_f32x2 df[NC];
            df[0] = zz[0] - q[0];
            ne_f32x2 d = df[0] * df[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) { df[c] = zz[c] - q[c]; d = __builtin_elementwise_fma(df[c], df[c], d); }
            const ne_f32x2 w = ne_f32x2{__expf(-d.x), __expf(-d.y)} * (inv_ri2 + q[NC]);
#pragma unroll
            for (int c = 0; c < NC; ++c) f2[c] = __builtin_elementwise_fma(w, df[c], f2[c]);
        }
        if (lim & 1) {
            const float* q = tile + pairs * (2 * (NC + 1));
            float df[NC];
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zs[c] - q[2 * c]; d = fmaf(df[c], df[c], d); }
            const float w = __expf(-d) * (inv_ri + q[2 * NC]);
#pragma unroll
            for (int c = 0; c < NC; ++c) f2[c].x = fmaf(w, df[c], f2[c].x);
        }
    }
    if (have) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < nc) grad[(size_t)(row0 + r) * nc + c] += coef * (f2[c].x + f2[c].y);
    }
}

// ---- PaCMAP mid-near sampling (pacmap.py:213-239): one 16-lane group per (row, slot) -----------------------------------------
// The lanes of a group split the feature dimension; each accumulates its share of the 6 candidate distances, the group adds
// them up (DPP row reductions) and every lane picks the second smallest.  The reference does this with six gathers of
// (n, 6, d) blocks, a distance kernel and a top-k per slot and iteration.
__global__ __launch_bounds__(256) void pacmap_mid_near_kernel(const float* __restrict__ X, int64_t ldx, int d, int64_t n, int n_mid,
                                                              int mode, uint64_t seed, uint32_t iter, int emit_index,
                                                              int64_t* __restrict__ out) {
    const int gl = threadIdx.x & 15;
    const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (item >= n * n_mid) return;
    const int64_t i = item / n_mid;
    const int slot = (int)(item - i * n_mid);
    const uint32_t key = neg_row_key(seed ^ 0x5bd1e9955bd1e995ull, iter, i) + 0x9E3779B9u * (uint32_t)(slot + 1);
    int64_t cand[6];
    const float* xi = X + (size_t)i * ldx;
    float acc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const uint32_t h = mix32(key + 0x85EBCA6Bu * (uint32_t)(c + 1));
        const int64_t r = 1 + (int64_t)(((uint64_t)h * (uint64_t)(n - 2)) >> 32);      // [1, n - 2]
        cand[c] = r + (r >= i ? 1 : 0);
        acc[c] = 0.f;
    }
    for (int f = gl; f < d; f += 16) {
        const float a = xi[f];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float b = X[(size_t)cand[c] * ldx + f];
            if (mode == 0) { const float t = a - b; acc[c] = fmaf(t, t, acc[c]); }
            else if (mode == 2) acc[c] += fabsf(a - b);
            else acc[c] = fmaf(-a, b, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = group_sum<16>(acc[c]);
    if (gl == 0) {
        int rank1 = -1;        // second of the ascending order; ties keep the earlier candidate first
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            int below = 0;
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) below += (acc[c2] < acc[c] || (acc[c2] == acc[c] && c2 < c)) ? 1 : 0;
            if (below == 1) rank1 = c;
        }
        int64_t pick = cand[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) pick = (rank1 == c) ? cand[c] : pick;
        // the reference stores `topk(...).indices[:, 1]`: the POSITION of the second nearest among the six, not its row
        // (pacmap.py:236-239) -- reproduced; emit_index is the test hook that shows which row that was
        out[item] = emit_index ? pick : (int64_t)rank1;
    }
}

// ---- PaCMAP pair losses (pacmap.py:213-265), closed-form gradient -------------------------------------------
// Three index tables per row: near pairs  w_nb * q/(10+q), mid-near pairs  w_mn * q/(1e4+q), further pairs
// w_fp / (1+q), with q = 1 + |z_i - z_j|^2.  d/dd of the three: 10 w_nb/(11+d)^2, 1e4 w_mn/(1e4+1+d)^2,
// -w_fp/(2+d)^2; both endpoints of a pair receive the force (autograd through the index gather).
struct PacmapParams {
    const float* Z;
    int64_t n;
    const int64_t* near; int m_near; float w_nb;
    const int64_t* mid;  int m_mid;  float w_mn;
    const int64_t* far_; int m_far;  float w_fp;
    float* grad;  // (n, nc), zero-initialised
    int nc;       // row width (PAD instances)
};

template <int NC, int G, bool PAD = false>
__global__ __launch_bounds__(256) void pacmap_grad_kernel(const PacmapParams P) {
    const int nc = PAD ? P.nc : NC;
    const int gl = threadIdx.x % G;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (i >= P.n) return;
    const Vec<NC> zi = load_zp<NC, PAD>(P.Z, i, nc);
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    const int total = P.m_near + P.m_mid + P.m_far;
    for (int p = gl; p < total; p += G) {
        int64_t j;
        float num, off, w;
        if (p < P.m_near) { j = P.near[(size_t)i * P.m_near + p]; num = 10.0f; off = 11.0f; w = P.w_nb; }
        else if (p < P.m_near + P.m_mid) { j = P.mid[(size_t)i * P.m_mid + (p - P.m_near)]; num = 1.0e4f; off = 10001.0f; w = P.w_mn; }
        else { j = P.far_[(size_t)i * P.m_far + (p - P.m_near - P.m_mid)]; num = -1.0f; off = 2.0f; w = P.w_fp; }
        if (w == 0.f) continue;
        const Vec<NC> zj = load_zp<NC, PAD>(P.Z, j, nc);
        float df[NC];
        const float d = sqdist<NC>(zi, zj, df);
        const float den = off + d;
        const float coef = 2.0f * w * num / (den * den);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float t = coef * df[c];
            g[c] += t;
            if (c < nc) unsafeAtomicAdd(&P.grad[(size_t)j * nc + c], -t);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum<G>(g[c]);
        if (gl == 0 && c < nc) unsafeAtomicAdd(&P.grad[(size_t)i * nc + c], g[c]);
    }
}

// ---- SGD(momentum) step, torch.optim.SGD semantics (no dampening / nesterov / weight decay) ----------
__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ Z, const float* __restrict__ grad,
                                                       float* __restrict__ buf, int64_t n, float lr, float momentum,
                                                       int first, int* __restrict__ nan_flag, int iter) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = grad[i];
    if (momentum != 0.f) {
        const float bprev = first ? 0.f : buf[i];
        g = first ? g : __fadd_rn(__fmul_rn(bprev, momentum), g);
        buf[i] = g;
    }
    const float z = fmaf(-lr, g, Z[i]);
    Z[i] = z;
    if (z != z) atomicCAS(nan_flag, 0, iter + 1);  // first iteration that produced a NaN (+1)
}

__global__ __launch_bounds__(256) void perm_debug_kernel(uint64_t seed, uint32_t iter, int64_t n_total, int n_neg,
                                                         int64_t* __restrict__ fwd, int64_t* __restrict__ inv) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_total * n_neg) return;
    const int64_t i = e / n_neg;
    const int c = (int)(e - i * n_neg);
    const PermKey K = perm_key(seed, iter, c, n_total);
    const uint32_t a = perm_inv((uint32_t)i, K);
    fwd[e] = perm_succ(a, K);
    inv[e] = perm_pred(a, K);
}

template <int G, typename Prm>
static int launch_group(void (*kern)(const Prm), const Prm& P, int64_t n_rows, hipStream_t st) {
    const int rpb = 256 / G;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_rows + rpb - 1) / rpb)), dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* umap.py:215-234 on the nnz CSR values: eps_per = A_max/(A+1e-3) (inf where A <= A_max/max_iter), next = copy.
 * scratch: >= 4 bytes of device memory. */
int tdr_umap_prepare_f32(const float* vals, int64_t nnz, int max_iter, float* eps_per, float* next, void* scratch,
                         void* stream) {
    if (!vals || !eps_per || !next || !scratch || nnz <= 0 || max_iter <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, vals, nnz, (unsigned*)scratch);
    hipLaunchKernelGGL(umap_prepare_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, vals, nnz,
                       (const unsigned*)scratch, (float)max_iter, eps_per, next);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

// Number of Z slices for the negative phase: 1 (single pass) while the embedding fits an XCD's L2, else 2 or 4 slices of
// <= 4 MiB -- every pass re-issues the row's whole negative loop, so more passes cost more than the L2 hits return.
static int umap_neg_slices(int64_t n_total, int nc) {
    const int64_t bytes = n_total * nc * (int64_t)sizeof(float);
    if (bytes <= (int64_t)3 << 20) return 1;
    const int64_t s = (bytes + ((int64_t)4 << 20) - 1) / ((int64_t)4 << 20);
    return s > 2 ? 4 : 2;  // 2 or 4: the dense passes split a row's negatives by exact binomial halving
}

/* Workspace of tdr_umap_grad_f32 (0 when the single-pass kernel is used). */
int64_t tdr_umap_grad_workspace_bytes(int64_t n_total, int64_t n_rows, int nc) {
    if (n_total < 2 || n_rows <= 0 || (nc != 2 && nc != 3)) return 0;
    if (umap_neg_slices(n_total, nc) <= 1) return 0;
    return 2 * n_rows * (int64_t)sizeof(int32_t) + n_rows * nc * (int64_t)sizeof(float);
}

/* One evaluation of UMAP's closed-form gradient for rows [row0, row0 + n_rows): grad (n_rows, nc).
 * neg_slices: 0 = automatic (L2-sliced negative phase for large N), 1 = single pass, > 1 = that many slices.
 * The sliced phase needs ws >= n_rows * (8 + 4 * nc) bytes (tdr_umap_grad_workspace_bytes for the automatic
 * choice); without it the single-pass kernel runs. */
int tdr_umap_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int64_t* rowptr,
                      const int32_t* cols, const float* eps_per, float* next, float a, float b, int n_iter,
                      int neg_rate, int n_negatives, const int64_t* neg_inj, uint64_t seed, float exag, float rep,
                      float eps, float* grad, int neg_slices, void* ws, int64_t ws_bytes, void* stream) {
    if (!Z || !rowptr || !cols || !eps_per || !next || !grad || n_rows <= 0 || n_total < 2) return TDR_ERR_BAD_ARG;
    if (nc != 2 && nc != 3) return TDR_ERR_UNSUPPORTED;
    UmapStepParams P;
    P.Z = Z; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.rowptr = rowptr; P.cols = cols;
    P.eps_per = eps_per; P.next = next; P.a = a; P.b = b; P.t1 = (float)(n_iter + 1); P.neg_rate = neg_rate;
    P.n_negatives = n_negatives; P.neg_inj = neg_inj; P.seed = seed; P.iter = (uint32_t)n_iter; P.exag = exag;
    P.rep = rep; P.eps = eps; P.grad = grad;
    P.nuse = nullptr; P.gr_acc = nullptr; P.j_lo = 0; P.j_hi = n_total; P.first = 1; P.last = 1; P.slice = 0; P.n_slices = 1;
    hipStream_t st = (hipStream_t)stream;
    int slices = neg_slices > 0 ? neg_slices : umap_neg_slices(n_total, nc);
    if (slices > n_total) slices = (int)n_total;
    const int64_t need = 2 * n_rows * (int64_t)sizeof(int32_t) + n_rows * nc * (int64_t)sizeof(float);
    // the row header packs n_use and the slice counts into 8-bit fields: larger counts take the single-pass kernel
    if (slices > 1 && ws && ws_bytes >= need && n_negatives > 0 && n_negatives <= 255 && neg_rate > 0) {
        P.nuse = (int32_t*)ws;
        P.n_slices = slices;
        P.gr_acc = (float*)((char*)ws + 2 * n_rows * sizeof(int32_t));
        int rc = (nc == 2) ? launch_group<16>(umap_grad_kernel<2, 16, 4, true>, P, n_rows, st)
                           : launch_group<16>(umap_grad_kernel<3, 16, 4, true>, P, n_rows, st);
        if (rc != TDR_OK) return rc;
        const int64_t step = (n_total + slices - 1) / slices;
        for (int sidx = 0; sidx < slices; ++sidx) {
            P.j_lo = sidx * step;
            P.j_hi = (sidx + 1) * step < n_total ? (sidx + 1) * step : n_total;
            P.first = sidx == 0; P.last = sidx == slices - 1; P.slice = sidx; P.n_slices = slices;
            if (!neg_inj && (slices == 2 || slices == 4) && n_total < 0x7fffffffLL) {
                // 8 lanes per row x 2 columns per lane and round (16 x 2, 8 x 3, 8 x 4, 4 x 6, 32 x 1 measured within 2 %
                // of each other, 32 x 1 10 % slower: the pass is bound by L2 line requests, not by lane utilisation)
                rc = (nc == 2) ? launch_group<8>(umap_neg_dense_kernel<2, 8, 2>, P, n_rows, st)
                               : launch_group<8>(umap_neg_dense_kernel<3, 8, 2>, P, n_rows, st);
            } else {
                rc = (nc == 2) ? launch_group<16>(umap_neg_slice_kernel<2, 16, 4>, P, n_rows, st)
                               : launch_group<16>(umap_neg_slice_kernel<3, 16, 4>, P, n_rows, st);
            }
            if (rc != TDR_OK) return rc;
        }
        return TDR_OK;
    }
    // 16 lanes per row x 4-deep unrolled gathers (32 x 2 and 8 x 8 measured within a few % of it)
    if (nc == 2) return launch_group<16>(umap_grad_kernel<2, 16, 4>, P, n_rows, st);
    return launch_group<16>(umap_grad_kernel<3, 16, 4>, P, n_rows, st);
}

static int g_ne_pull4 = 1;         // 0: tdr_ne_grad_perm_f32 keeps the 16-lanes-per-row kernel (tdr_ne_grad_perm_lanes: measurements, equality test)
static int g_ne_halves_mode = 0;   // 0 = by size, 1 = never (tdr_ne_grad_perm_halves: measurements and the equality test)

static int ne_grad_launch(NeStepParams& S, float* rowsum, hipStream_t st) {
    const int nc = S.nc;
    const int64_t n_rows = S.n_rows;
    const int64_t blocks1 = (n_rows + 15) / 16;
    const int64_t grid_rows = S.neg_halves == 2 ? ((blocks1 + 3) / 4) * 8 : blocks1;     // two visits per (padded) row block
    if (rowsum) {   // InfoTSNE with the permutation sampler, pass 1: every row's normaliser over its own draws
        if (S.neg_halves == 2) {
            hipError_t e0 = hipMemsetAsync(rowsum, 0, (size_t)S.n_total * sizeof(float), st);
            if (e0 != hipSuccess) return (int)e0;
        }
        const dim3 grid((unsigned)grid_rows);
        if (nc == 2) hipLaunchKernelGGL((ne_rowsum_kernel<2, 16>), grid, dim3(256), 0, st, S, rowsum);
        else if (nc == 3) hipLaunchKernelGGL((ne_rowsum_kernel<3, 16>), grid, dim3(256), 0, st, S, rowsum);
        else if (nc <= 4) hipLaunchKernelGGL((ne_rowsum_kernel<4, 16, true>), grid, dim3(256), 0, st, S, rowsum);
        else if (nc <= 8) hipLaunchKernelGGL((ne_rowsum_kernel<8, 16, true>), grid, dim3(256), 0, st, S, rowsum);
        else if (nc <= 16) hipLaunchKernelGGL((ne_rowsum_kernel<16, 16, true>), grid, dim3(256), 0, st, S, rowsum);
        else hipLaunchKernelGGL((ne_rowsum_kernel<32, 16, true>), grid, dim3(256), 0, st, S, rowsum);
        TDR_CHECK_LAUNCH();
    }
    if (S.neg_halves == 2) {
#define TDR_NE2(K) { hipLaunchKernelGGL(K, dim3((unsigned)grid_rows), dim3(256), 0, st, S); hipError_t e = hipGetLastError(); return e == hipSuccess ? TDR_OK : (int)e; }
        if (nc == 2) TDR_NE2((ne_grad_kernel<2, 16>))
        if (nc == 3) TDR_NE2((ne_grad_kernel<3, 16>))
        if (nc <= 4) TDR_NE2((ne_grad_kernel<4, 16, true>))
        if (nc <= 8) TDR_NE2((ne_grad_kernel<8, 16, true>))
        if (nc <= 16) TDR_NE2((ne_grad_kernel<16, 16, true>))
        TDR_NE2((ne_grad_kernel<32, 16, true>))
#undef TDR_NE2
    }
    if (g_ne_pull4 && S.perm_neg && S.kind == 0 && S.t_rowptr && !S.neg_inj && S.n_neg <= 8 && S.neg_halves != 2 && (nc == 2 || nc == 3)) {
        const dim3 grid((unsigned)((n_rows + 63) / 64));
        if (nc == 2) hipLaunchKernelGGL(ne_pull4_kernel<2>, grid, dim3(256), 0, st, S);
        else hipLaunchKernelGGL(ne_pull4_kernel<3>, grid, dim3(256), 0, st, S);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? TDR_OK : (int)e;
    }
    // exact instances for 2 and 3 components, zero-padded register instances for any other width up to 32
    if (nc == 2) return launch_group<16>(ne_grad_kernel<2, 16>, S, n_rows, st);
    if (nc == 3) return launch_group<16>(ne_grad_kernel<3, 16>, S, n_rows, st);
    if (nc <= 4) return launch_group<16>(ne_grad_kernel<4, 16, true>, S, n_rows, st);
    if (nc <= 8) return launch_group<16>(ne_grad_kernel<8, 16, true>, S, n_rows, st);
    if (nc <= 16) return launch_group<16>(ne_grad_kernel<16, 16, true>, S, n_rows, st);
    return launch_group<16>(ne_grad_kernel<32, 16, true>, S, n_rows, st);
}

/* Sparse attraction (+ LargeVis negative-sample repulsion) gradient; grad (N, nc) must be zeroed by the
 * caller.  Both endpoints of every edge receive their share: with the transposed graph (t_rowptr / t_src /
 * t_val = in-edges of rows [row0, row0+n_rows)) each row pulls its in-edges itself and only the negative
 * samples use fp32 atomics; without it (NULL) every neighbour edge pushes to its far endpoint atomically. */
int tdr_ne_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                    const float* P_, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, int kind,
                    float exag, float rep_coef, int n_neg, const int64_t* neg_inj, uint64_t seed, int n_iter,
                    float* grad, void* stream) {
    if (!Z || !nn || !P_ || !grad || n_rows <= 0 || k <= 0 || n_total < 2) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32) return TDR_ERR_UNSUPPORTED;
    if (kind < 0 || kind > 3) return TDR_ERR_BAD_ARG;
    NeStepParams S;
    S.nc = nc;
    S.perm_neg = 0; S.rowsum = nullptr; S.neg_halves = 1;
    S.Z = Z; S.n_total = n_total; S.row0 = row0; S.n_rows = n_rows; S.nn = nn; S.P = P_; S.k = k; S.kind = kind;
    S.exag = exag; S.rep_coef = rep_coef; S.n_neg = n_neg; S.neg_inj = neg_inj; S.seed = seed;
    S.iter = (uint32_t)n_iter; S.grad = grad;
    if (t_rowptr && (!t_src || !t_val)) return TDR_ERR_BAD_ARG;
    S.t_rowptr = t_rowptr; S.t_src = t_src; S.t_val = t_val;
    return ne_grad_launch(S, nullptr, (hipStream_t)stream);
}

/* The same gradient with the negatives drawn as keyed PERMUTATIONS of the rows (kinds 0 = LargeVis, 3 = InfoTSNE): column c
 * of iteration t is j = P_{t,c}(i), so the row that drew j is P^{-1}(j) and a row pulls both its own draws and the draws that
 * hit it -- no atomics for the far endpoints (tdr_embed_common.h, "permutation sampler").  Each row's draws are uniform over
 * the rows and independent across columns / iterations, as neighbor_embedding/base.py:628-636 draws them; across the rows of
 * one column they are distinct.  Needs the transposed graph (the neighbour edges are pulled too).  rowsum_ws: kind 3 only,
 * n_total floats of scratch (the row normalisers of every row); all rows in one call (row0 = 0, n_rows = n_total). */
int tdr_ne_grad_perm_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                         const float* P_, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, int kind,
                         float exag, float rep_coef, int n_neg, uint64_t seed, int n_iter, float* rowsum_ws, float* grad,
                         void* stream) {
    if (!Z || !nn || !P_ || !grad || !t_rowptr || !t_src || !t_val || n_rows <= 0 || k <= 0 || n_total < 2) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32 || n_total > 0x7fffffffLL) return TDR_ERR_UNSUPPORTED;
    if ((kind != 0 && kind != 3) || n_neg <= 0) return TDR_ERR_BAD_ARG;
    if (kind == 3 && (!rowsum_ws || row0 != 0 || n_rows != n_total)) return TDR_ERR_BAD_ARG;
    NeStepParams S;
    S.nc = nc;
    S.perm_neg = 1; S.rowsum = kind == 3 ? rowsum_ws : nullptr;
    // two halves where the negatives dominate the row (>= 64 items) and the gathered tables do not fit an XCD's 4 MB of L2:
    // InfoTSNE's 2 x 300 items, 30 iterations at N = 500k / 700k / 1M: 122 / 229 / 413 -> 110 / 202 / 245 ms; at N = 300k
    // (3.6 MB) 64 -> 102 ms, hence the threshold.  LargeVis (2 x 5 items against ~30 edge items): 0.236 -> 0.288 ms at N = 1M.
    S.neg_halves = (g_ne_halves_mode != 1 && row0 == 0 && n_rows == n_total && 2 * n_neg >= 64 &&
                    (int64_t)n_total * (nc + (kind == 3 ? 1 : 0)) * (int64_t)sizeof(float) > (11ll << 19)) ? 2 : 1;   // 5.5 MB
    S.Z = Z; S.n_total = n_total; S.row0 = row0; S.n_rows = n_rows; S.nn = nn; S.P = P_; S.k = k; S.kind = kind;
    S.exag = exag; S.rep_coef = rep_coef; S.n_neg = n_neg; S.neg_inj = nullptr; S.seed = seed;
    S.iter = (uint32_t)n_iter; S.grad = grad;
    S.t_rowptr = t_rowptr; S.t_src = t_src; S.t_val = t_val;
    return ne_grad_launch(S, kind == 3 ? rowsum_ws : nullptr, (hipStream_t)stream);
}

/* 1 when tdr_ne_grad_runs_f32 serves the shape: 2 / 3 components, 1..8 negatives, at least two runs of 16 rows. */
int tdr_ne_grad_runs_supported(int nc, int64_t n_total, int n_neg) {
    return (nc == 2 || nc == 3) && n_neg >= 1 && n_neg <= 8 && n_total > 2 * RUNP_LEN && n_total <= 0x7fffffffLL;
}

/* LargeVis gradient (kind 0 of tdr_ne_grad_perm_f32) of rows [row0, row0 + n_rows) with the negatives drawn by the RUN-permutation
 * sampler and served from LDS (ne_pull4_runs_kernel above: law, padding rule).  nn / P (n_rows, k), t_* = in-edges of those rows,
 * Z (n_total, nc) the whole embedding, 16-byte aligned; grad (n_rows, nc) is WRITTEN (every row once, complete: nothing is sent to
 * other rows, so a rank of a row-sharded fit steps its own rows from it).  The sampler is keyed by global rows and runs: a row
 * chunk gives the bits of the same rows of the full launch. */
int tdr_ne_grad_runs_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn, const float* P_, int k,
                         const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, float exag, float rep_coef, int n_neg,
                         uint64_t seed, int n_iter, float* grad, void* stream) {
    if (!Z || !nn || !P_ || !grad || !t_rowptr || !t_src || !t_val || k <= 0 || ((uintptr_t)Z & 15u)) return TDR_ERR_BAD_ARG;
    if (row0 < 0 || n_rows <= 0 || row0 + n_rows > n_total) return TDR_ERR_BAD_ARG;
    if (!tdr_ne_grad_runs_supported(nc, n_total, n_neg)) return TDR_ERR_UNSUPPORTED;
    NeStepParams S;
    S.nc = nc; S.perm_neg = 1; S.rowsum = nullptr; S.neg_halves = 1;
    S.Z = Z; S.n_total = n_total; S.row0 = row0; S.n_rows = n_rows; S.nn = nn; S.P = P_; S.k = k; S.kind = 0;
    S.exag = exag; S.rep_coef = rep_coef; S.n_neg = n_neg; S.neg_inj = nullptr; S.seed = seed; S.iter = (uint32_t)n_iter; S.grad = grad;
    S.t_rowptr = t_rowptr; S.t_src = t_src; S.t_val = t_val;
    const dim3 grid((unsigned)((row0 + n_rows - 1) / 64 - row0 / 64 + 1));
    if (nc == 2) hipLaunchKernelGGL(ne_pull4_runs_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, S);
    else hipLaunchKernelGGL(ne_pull4_runs_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, S);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* test hook: fwd[i, c] = the row that row i draws in column c, inv[i, c] = the row whose draw of column c hits row i (-1: the pair
 * does not exist -- its other endpoint is padding of the last run) */
int tdr_runs_negatives_debug(uint64_t seed, int n_iter, int64_t n_total, int n_neg, int64_t* fwd, int64_t* inv, void* stream) {
    if (!fwd || !inv || n_total <= 2 * RUNP_LEN || n_total > 0x7fffffffLL || n_neg <= 0) return TDR_ERR_BAD_ARG;
    const int64_t total = n_total * n_neg;
    hipLaunchKernelGGL(runperm_debug_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       (uint32_t)n_iter, n_total, n_neg, fwd, inv);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Measurement / test switch: lanes per row of tdr_ne_grad_perm_f32's LargeVis launch -- 4 (default: ne_pull4_kernel) or 16
 * (ne_grad_kernel, the form of rounds 3-5).  Returns the previous value. */
int tdr_ne_grad_perm_lanes(int lanes) {
    const int old = g_ne_pull4 ? 4 : 16;
    if (lanes == 4 || lanes == 16) g_ne_pull4 = lanes == 4;
    return old;
}

/* Measurement / test switch: 1 = tdr_ne_grad_perm_f32 never splits its launch into the two halves of the index range, 0 (default)
 * = by size.  Returns the previous value. */
int tdr_ne_grad_perm_halves(int mode) {
    const int old = g_ne_halves_mode;
    g_ne_halves_mode = mode == 1 ? 1 : 0;
    return old;
}

/* Test hook: the permutation sampler's draws -- fwd[i][c] = P_{t,c}(i) and inv[i][c] = P_{t,c}^{-1}(i), (n_total, n_neg) int64. */
int tdr_perm_negatives_debug(uint64_t seed, int n_iter, int64_t n_total, int n_neg, int64_t* fwd, int64_t* inv, void* stream) {
    if (!fwd || !inv || n_total < 2 || n_total > 0x7fffffffLL || n_neg <= 0) return TDR_ERR_BAD_ARG;
    const int64_t total = n_total * n_neg;
    hipLaunchKernelGGL(perm_debug_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       (uint32_t)n_iter, n_total, n_neg, fwd, inv);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

// column segments of the dense TSNE repulsion: >= ~4096 workgroups, segments of >= 1024 columns (multiples of the 256-column tile)
// (a function of n_total alone: a rank that holds a row chunk cuts the columns where the single-process launch does, so the
// sums keep their association whatever the sharding)
static inline int tsne_rep_segments(int64_t n_total) {
    const int64_t row_blocks = (n_total + 255) / 256;
    int64_t s = (4096 + row_blocks - 1) / row_blocks;
    if (s > n_total / 1024) s = n_total / 1024;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

static int tsne_repulsion_launch(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* F, double* S,
                                 int n_seg, hipStream_t st) {
    const int64_t cols = n_seg > 1 ? (((n_total + n_seg - 1) / n_seg + 255) / 256) * 256 : n_total;
    const dim3 grid((unsigned)((n_rows + 255) / 256), (unsigned)(n_seg > 1 ? (n_total + cols - 1) / cols : 1));
    if (nc == 2) hipLaunchKernelGGL(tsne_repulsion_kernel<2>, grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else if (nc == 3) hipLaunchKernelGGL(tsne_repulsion_kernel<3>, grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else if (nc >= 1 && nc <= 4) hipLaunchKernelGGL((tsne_repulsion_kernel<4, true>), grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else if (nc <= 8 && nc >= 1) hipLaunchKernelGGL((tsne_repulsion_kernel<8, true>), grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else if (nc <= 16 && nc >= 1) hipLaunchKernelGGL((tsne_repulsion_kernel<16, true>), grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else if (nc <= 32 && nc >= 1) hipLaunchKernelGGL((tsne_repulsion_kernel<32, true>), grid, dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc, cols);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return (int)grid.y;     // > 0: the number of planes written
}

/* TSNE dense repulsion for rows [row0, row0+n_rows): F (n_rows, nc) = sum_j (z_i - z_j)/(1+d)^2 and
 * *S (double, device, caller-zeroed) += sum_ij 1/(1+d_ij). */
int tdr_tsne_repulsion_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* F, double* S,
                           void* stream) {
    if (!Z || !F || !S || n_rows <= 0) return TDR_ERR_BAD_ARG;
    const int rc = tsne_repulsion_launch(Z, nc, n_total, row0, n_rows, F, S, 1, (hipStream_t)stream);
    return rc > 0 ? TDR_OK : rc;
}

/* Bytes of the workspace with which tdr_tsne_repulsion_split_f32 spreads the columns over several workgroups per row block
 * (0 = the launch of this size is not split).  A row block of 256 rows against all columns is one workgroup: N = 50k is 196
 * workgroups for 256 CUs, N = 100k 1.5 per CU. */
int64_t tdr_tsne_repulsion_workspace_bytes(int64_t n_total, int64_t n_rows, int nc) {
    if (n_total <= 0 || n_rows <= 0 || nc < 1) return 0;
    const int n_seg = tsne_rep_segments(n_total);
    return n_seg > 1 ? (int64_t)n_seg * n_rows * nc * (int64_t)sizeof(float) : 0;
}

/* The same result as tdr_tsne_repulsion_f32 with the columns cut into segments (blockIdx.y), each writing its partial forces to
 * a plane of `ws`; the planes are added in segment order (a fixed association).  ws NULL / too small: the unsplit launch. */
int tdr_tsne_repulsion_split_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* F, double* S,
                                 void* ws, int64_t ws_bytes, void* stream) {
    if (!Z || !F || !S || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_seg = tsne_rep_segments(n_total);
    if (n_seg <= 1 || !ws || ws_bytes < (int64_t)n_seg * n_rows * nc * (int64_t)sizeof(float))
        return tdr_tsne_repulsion_f32(Z, nc, n_total, row0, n_rows, F, S, stream);
    const int planes = tsne_repulsion_launch(Z, nc, n_total, row0, n_rows, (float*)ws, S, n_seg, st);
    if (planes <= 0) return planes;
    const int64_t cnt = n_rows * nc;
    hipLaunchKernelGGL(sum_planes_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (const float*)ws, planes, cnt, F);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* SNE dense repulsion, pass 1: R[r] = sum_j exp(-|z_{row0+r} - z_j|^2) for r < n_rows (sne.py:170-179). */
int tdr_sne_rowsum_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* R, void* stream) {
    if (!Z || !R || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    if (nc == 2) hipLaunchKernelGGL(sne_rowsum_kernel<2>, dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc == 3) hipLaunchKernelGGL(sne_rowsum_kernel<3>, dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc >= 1 && nc <= 4) hipLaunchKernelGGL((sne_rowsum_kernel<4, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc <= 8 && nc >= 1) hipLaunchKernelGGL((sne_rowsum_kernel<8, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc <= 16 && nc >= 1) hipLaunchKernelGGL((sne_rowsum_kernel<16, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc <= 32 && nc >= 1) hipLaunchKernelGGL((sne_rowsum_kernel<32, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* SNE dense repulsion, pass 2: grad (n_total, nc) rows [row0, row0+n_rows) += coef * sum_j exp(-d_ij)
 * (1/R_i + 1/R_j) (z_i - z_j); R holds the row sums of ALL n_total rows. */
int tdr_sne_repulsion_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const float* R,
                          float coef, float* grad, void* stream) {
    if (!Z || !R || !grad || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    if (nc == 2) hipLaunchKernelGGL(sne_repulsion_kernel<2>, dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc == 3) hipLaunchKernelGGL(sne_repulsion_kernel<3>, dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc >= 1 && nc <= 4) hipLaunchKernelGGL((sne_repulsion_kernel<4, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc <= 8 && nc >= 1) hipLaunchKernelGGL((sne_repulsion_kernel<8, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc <= 16 && nc >= 1) hipLaunchKernelGGL((sne_repulsion_kernel<16, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc <= 32 && nc >= 1) hipLaunchKernelGGL((sne_repulsion_kernel<32, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* grad[i] += (coef / *S) * F[i] for i < n (flat). */
int tdr_add_scaled_f32(float* grad, const float* F, const double* S, float coef, int64_t n, void* stream) {
    if (!grad || !F || !S || n <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_scaled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, F, S, coef, n);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* In-place SGD step on n flat elements. buf may be NULL when momentum == 0. nan_flag: device int, set to
 * (iteration + 1) by the first step that produces a NaN (check_NaNs, affinity_matcher.py:315). */
int tdr_sgd_step_f32(float* Z, const float* grad, float* buf, int64_t n, float lr, float momentum, int first,
                     int* nan_flag, int n_iter, void* stream) {
    if (!Z || !grad || !nan_flag || n <= 0) return TDR_ERR_BAD_ARG;
    if (momentum != 0.f && !buf) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, grad, buf, n, lr, momentum, first, nan_flag, n_iter);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* PaCMAP mid-near pairs (neighbor_embedding/pacmap.py:213-239), one launch for all rows and slots: for row i and slot s draw
 * 6 candidates j = r + (r >= i), r uniform in [1, n - 2] (the reference's `randint(1, n - 1)` + shift past the row itself),
 * rank them by their INPUT-space distance to x_i and keep the second nearest.  mode 0: squared / plain Euclidean (same
 * ranking), 2: manhattan, 3: angular (-dot).  Candidates come from the counter hash keyed by (seed, iteration, row, slot,
 * candidate); ties between candidates keep the earlier one, as a stable ascending sort would.  out (n, n_mid) int64:
 * emit_index = 0: the POSITION (0..5) of that candidate among the six -- what the reference's `topk(...).indices[:, 1]` stores
 * and then uses as a row index (pacmap.py:236-239; reproduced for result parity); 1: the candidate's row (test hook). */
int tdr_pacmap_mid_near_f32(const float* X, int64_t ldx, int d, int64_t n, int n_mid, int mode, uint64_t seed, int n_iter,
                            int emit_index, int64_t* out, void* stream) {
    if (!X || !out || n < 8 || n >= 0x7fffffffLL || n_mid <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    if (mode != 0 && mode != 2 && mode != 3) return TDR_ERR_UNSUPPORTED;
    const int64_t items = n * n_mid;
    hipLaunchKernelGGL(pacmap_mid_near_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, (hipStream_t)stream, X, ldx, d, n,
                       n_mid, mode, seed, (uint32_t)n_iter, emit_index, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Gradient of PaCMAP's three pair losses (neighbor_embedding/pacmap.py:213-265) on the (n, m_*) int64 index tables
 * near / mid / far (mid or far may be NULL with m = 0); grad (n, nc) must be zeroed by the caller. */
int tdr_pacmap_grad_f32(const float* Z, int nc, int64_t n, const int64_t* near_idx, int m_near, float w_nb,
                        const int64_t* mid_idx, int m_mid, float w_mn, const int64_t* far_idx, int m_far, float w_fp,
                        float* grad, void* stream) {
    if (!Z || !grad || n <= 0 || m_near < 0 || m_mid < 0 || m_far < 0) return TDR_ERR_BAD_ARG;
    if ((m_near > 0 && !near_idx) || (m_mid > 0 && !mid_idx) || (m_far > 0 && !far_idx)) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32) return TDR_ERR_UNSUPPORTED;
    PacmapParams P;
    P.nc = nc;
    P.Z = Z; P.n = n; P.near = near_idx; P.m_near = m_near; P.w_nb = w_nb; P.mid = mid_idx; P.m_mid = m_mid; P.w_mn = w_mn;
    P.far_ = far_idx; P.m_far = m_far; P.w_fp = w_fp; P.grad = grad;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return launch_group<16>(pacmap_grad_kernel<2, 16>, P, n, st);
    if (nc == 3) return launch_group<16>(pacmap_grad_kernel<3, 16>, P, n, st);
    if (nc <= 4) return launch_group<16>(pacmap_grad_kernel<4, 16, true>, P, n, st);
    if (nc <= 8) return launch_group<16>(pacmap_grad_kernel<8, 16, true>, P, n, st);
    if (nc <= 16) return launch_group<16>(pacmap_grad_kernel<16, 16, true>, P, n, st);
    return launch_group<16>(pacmap_grad_kernel<32, 16, true>, P, n, st);
}

/* Test hook: negatives drawn by the dense slice passes of tdr_umap_grad_f32 (n_slices = 2 or 4) for rows
 * [row0, row0 + n_rows) with nuse[r] negatives each -> out (n_rows, width) int64, -1 padded. */
int tdr_umap_debug_negatives(uint64_t seed, int n_iter, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nuse,
                             int n_slices, int width, int64_t* out, void* stream) {
    if (!nuse || !out || n_rows <= 0 || n_total < 2 || width <= 0 || (n_slices != 1 && n_slices != 2 && n_slices != 4 && n_slices != 8)) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(umap_debug_negatives_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       seed, (uint32_t)n_iter, n_total, row0, n_rows, nuse, n_slices, width, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
