// Device helpers shared by the embedding-loop kernels (tdr_embed.hip, tdr_umap_sched.hip): Z row loads, the counter-hash
// negative sampler (neighbor_embedding/base.py:617-649 replaced by a stateless generator), hardware pow / rcp.
#pragma once
#include "tdr_common.h"

namespace tdr {

template <int NC>
struct Vec {
    float v[NC];
};

template <int NC>
__device__ __forceinline__ Vec<NC> load_z(const float* __restrict__ Z, int64_t i) {
    Vec<NC> r;
    if (NC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(Z + (size_t)i * 2);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) r.v[c] = Z[(size_t)i * NC + c];
    }
    return r;
}

// row i of a block whose rows are `nc` floats wide into NC >= nc registers (zeros beyond nc contribute nothing to
// distances or forces): the PAD instances of the gradient kernels serve any embedding width up to NC
template <int NC, bool PAD>
__device__ __forceinline__ Vec<NC> load_zp(const float* __restrict__ Z, int64_t i, int nc) {
    if (!PAD) return load_z<NC>(Z, i);
    Vec<NC> r;
#pragma unroll
    for (int c = 0; c < NC; ++c) r.v[c] = c < nc ? Z[(size_t)i * nc + c] : 0.f;
    return r;
}

// Counter-based hash generator for the negatives: three chained rounds of the "triple32" integer mixer
// (xorshift-multiply, bias-tested avalanche) over (seed, row) -> (+iteration) -> (+column).  The first two
// rounds are per row / per iteration and hoisted out of the column loop, so one negative costs ~10 VALU ops
// (Philox4x32-10 costs ~90 and made the kernel instruction-bound).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 17; x *= 0xed5ad4bbu;
    x ^= x >> 11; x *= 0xac4c1b51u;
    x ^= x >> 15; x *= 0x31848babu;
    x ^= x >> 14;
    return x;
}
// Two-multiply mixer ("lowbias32" constants) for the per-negative hash of the sliced samplers: its input is the row key
// -- already three rounds of mix32 over (seed, row, iteration) -- plus a Weyl step per column, so one more strong round
// is not needed, and a 32-bit integer multiply is a quarter-rate instruction (16 SIMD cycles per wavefront): the scheduled
// gradient pass spends 48 of them per wavefront on the sampler.
__device__ __forceinline__ uint32_t mix32_item(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t neg_row_key(uint64_t seed, uint32_t iter, int64_t grow) {
    uint32_t h = mix32((uint32_t)grow ^ (uint32_t)seed);
    h = mix32(h + (uint32_t)((uint64_t)grow >> 32) * 0x9E3779B9u + (uint32_t)(seed >> 32));
    return mix32(h ^ (iter * 0x85EBCA6Bu + 0xC2B2AE35u));
}
__device__ __forceinline__ int64_t sample_negative(uint32_t row_key, int64_t grow, int col, int64_t n_total) {
    // neighbor_embedding/base.py:628-636 : r ~ U{0..N-2}, then +1 where r >= own index.
    // 32 random bits -> [0, N-1) by multiply-shift range reduction (bias < N / 2^32).
    const uint32_t x = mix32(row_key + (uint32_t)col * 0x9E3779B9u);
    int64_t r = (int64_t)(((uint64_t)x * (uint64_t)(n_total - 1)) >> 32);
    if (r >= grow) r += 1;
    return r;
}

// ---- permutation sampler (LargeVis / InfoTSNE negatives, pull form) ------------------------------------------------------
// The reference draws, per row i and column c, a uniform j != i (neighbor_embedding/base.py:628-636) and autograd sends the
// pair's force to BOTH endpoints.  With a hash sampler the far endpoint's share has to be scattered with atomics (10 M
// device-scope fp32 atomics per LargeVis iteration at N = 1M: 0.5 of its 0.54 ms).  Here column c of iteration t is a keyed
// pseudo-random PERMUTATION of the rows, j = P_{t,c}(i) (round 4: a fixed-point-free one, see perm_succ below): every row's
// draws are uniform over the OTHER rows and independent across columns and iterations (the per-row law of the reference's
// sampler), and the row that drew j is P^{-1}(j) -- so a row PULLS both its own draws and the draws that hit
// it, and nothing is scattered.  What differs from independent draws is the joint law across rows of one column (no two
// rows draw the same j): every row is the far endpoint of exactly n_negatives pairs instead of Poisson(n_negatives).
// P = three rounds of (odd multiply + keyed add mod 2^b, xorshift by ceil(b/2)) on b = ceil(log2 N) bits with cycle walking
// into [0, N); each step is invertible (the xorshift is an involution at that shift).
struct PermKey {
    uint32_t a1, a2, a3, mask, n;
    int s;
};
constexpr uint32_t PERM_M1 = 0x9E3779B1u, PERM_M2 = 0x85EBCA6Bu;
constexpr uint32_t inv_odd_u32(uint32_t a) {
    uint32_t x = a;                       // a * a = 1 mod 8: 3 correct bits, doubled by every Newton step
    x *= 2u - a * x; x *= 2u - a * x; x *= 2u - a * x; x *= 2u - a * x;
    return x;
}
constexpr uint32_t PERM_I1 = inv_odd_u32(PERM_M1), PERM_I2 = inv_odd_u32(PERM_M2);
__device__ __forceinline__ PermKey perm_key(uint64_t seed, uint32_t iter, int col, int64_t n_total) {
    PermKey K;
    uint32_t h = mix32((uint32_t)seed ^ (0x9E3779B9u * (uint32_t)(col + 1)));
    h = mix32(h + (uint32_t)(seed >> 32) + iter * 0x85EBCA6Bu);
    K.a1 = h; K.a2 = mix32(h ^ 0xC2B2AE35u); K.a3 = mix32(h + 0x27D4EB2Fu);
    int b = 32 - __clz((uint32_t)(n_total - 1));
    if (b < 2) b = 2;
    K.mask = b >= 32 ? 0xffffffffu : ((1u << b) - 1u);
    K.s = (b + 1) >> 1;
    K.n = (uint32_t)n_total;
    return K;
}
__device__ __forceinline__ uint32_t perm_fwd(uint32_t x, const PermKey& K) {
    do {
        x = (x * PERM_M1 + K.a1) & K.mask; x ^= x >> K.s;
        x = (x * PERM_M2 + K.a2) & K.mask; x ^= x >> K.s;
        x = (x * PERM_M1 + K.a3) & K.mask; x ^= x >> K.s;
    } while (x >= K.n);
    return x;
}
__device__ __forceinline__ uint32_t perm_inv(uint32_t x, const PermKey& K) {
    do {
        x ^= x >> K.s; x = ((x - K.a3) * PERM_I1) & K.mask;
        x ^= x >> K.s; x = ((x - K.a2) * PERM_I2) & K.mask;
        x ^= x >> K.s; x = ((x - K.a1) * PERM_I1) & K.mask;
    } while (x >= K.n);
    return x;
}

// Round 4: a row never draws ITSELF (the reference shifts self out: r ~ U{0..N-2}, +1 where r >= own index, base.py:634-636).
// A keyed permutation has ~1 fixed point per column; the sampler therefore uses the permutation as a keyed CYCLIC ORDER of
// the rows and lets every row draw its SUCCESSOR in it:  j = pi(pi^-1(i) + 1 mod N).  The map i -> j is one N-cycle -- a
// bijection without fixed points -- and j is uniform over the N - 1 other rows (the reference's per-row law); the row that
// drew i is its predecessor pi(pi^-1(i) - 1 mod N).  a = pi^-1(i) serves both: three permutation evaluations per (row,
// column) for the pair (own draw, drawer).
__device__ __forceinline__ uint32_t perm_succ(uint32_t a, const PermKey& K) { return perm_fwd(a + 1u == K.n ? 0u : a + 1u, K); }
__device__ __forceinline__ uint32_t perm_pred(uint32_t a, const PermKey& K) { return perm_fwd(a == 0u ? K.n - 1u : a - 1u, K); }

// d^b through the hardware log2 / exp2 (relative error ~ |b log2 d| * 2^-23, i.e. <= ~3e-6 for the
// distances an embedding produces) and reciprocals through v_rcp_f32 (1 ulp): the force coefficients stay
// well inside the 1e-5 parity budget while the kernel drops from ~300 to ~80 VALU ops per edge.
__device__ __forceinline__ float fast_pow(float d, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(d)); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int NC>
__device__ __forceinline__ float sqdist(const Vec<NC>& a, const Vec<NC>& b, float (&df)[NC]) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { df[c] = a.v[c] - b.v[c]; d = __fadd_rn(d, __fmul_rn(df[c], df[c])); }
    return d;
}

// ---- split of a row's negative count over S equal slices of the index range (S in {1, 2, 4, 8}) ------------------
// The number of a row's n negatives that fall into the lower of two equal halves of the range is Binomial(n, 1/2) =
// the population count of n fair hash bits; applied level by level (1 / 2 / 3 levels for 2 / 4 / 8 slices) this is an
// exact multinomial split, and "split the count, then draw uniformly inside the part" has the distribution of n
// i.i.d. uniform draws (neighbor_embedding/base.py:628-636).
__device__ __forceinline__ int binomial_half(uint32_t key, int n) {
    int m = 0;
    for (int t = 0; t * 32 < n; ++t) {
        uint32_t w = mix32(key + 0x7F4A7C15u * (uint32_t)(t + 1));
        const int rem = n - t * 32;
        if (rem < 32) w &= (1u << rem) - 1u;
        m += __popc(w);
    }
    return m;
}
// the same count with the words spread over the G lanes of a row group (every lane returns the sum)
template <int G>
__device__ __forceinline__ int binomial_half_group(uint32_t key, int n, int gl) {
    int m = 0;
    for (int t = gl; t * 32 < n; t += G) {
        uint32_t w = mix32(key + 0x7F4A7C15u * (uint32_t)(t + 1));
        const int rem = n - t * 32;
        if (rem < 32) w &= (1u << rem) - 1u;
        m += __popc(w);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    return m;
}
__device__ __forceinline__ uint32_t split_key(uint32_t rkey, int level, int group) {
    if (level == 0) return rkey ^ 0x9E3779B9u;
    if (level == 1) return rkey ^ (0x85EBCA6Bu + 0x27D4EB2Fu * (uint32_t)group);
    return rkey ^ (0xC2B2AE35u + 0x165667B1u * (uint32_t)group);
}
__device__ __forceinline__ int slice_count(uint32_t rkey, int n_use, int slice, int n_slices) {
    int mine = n_use, level = 0;
    for (int span = n_slices; span > 1; span >>= 1, ++level) {
        const int lo = binomial_half(split_key(rkey, level, slice / span), mine);
        mine = ((slice / (span >> 1)) & 1) ? mine - lo : lo;
    }
    return mine;
}
template <int G>
__device__ __forceinline__ int slice_count_group(uint32_t rkey, int n_use, int slice, int n_slices, int gl) {
    int mine = n_use, level = 0;
    for (int span = n_slices; span > 1; span >>= 1, ++level) {
        const int lo = binomial_half_group<G>(split_key(rkey, level, slice / span), mine, gl);
        mine = ((slice / (span >> 1)) & 1) ? mine - lo : lo;
    }
    return mine;
}
// the c-th negative a row draws inside `slice` of the reduced index range [0, N-1): uniform in the slice, then the
// reference's "+1 where >= own index" (base.py:634-636)
__device__ __forceinline__ uint32_t slice_negative(uint32_t rkey, uint32_t gi, int col, int slice, int n_slices, uint32_t nred) {
    const uint32_t step = (nred + (uint32_t)n_slices - 1u) / (uint32_t)n_slices;
    const uint32_t r_lo = (uint32_t)slice * step;
    const uint32_t r_len = (r_lo < nred) ? ((nred - r_lo < step) ? nred - r_lo : step) : 0u;
    const uint32_t x = mix32_item(rkey + 0x632BE5ABu * (uint32_t)(slice + 1) + (uint32_t)col * 0x9E3779B9u);
    const uint32_t rr = r_lo + __umulhi(x, r_len);
    return rr + (rr >= gi ? 1u : 0u);
}

}  // namespace tdr
