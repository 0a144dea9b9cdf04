// K7 / K8 -- matrix-free dense affinities for TSNEkhorn (symmetric entropic affinity in, Sinkhorn out).
//
// Replaces (citations under /root/reference/torchdr):
//   affinity/entropic.py:37-42, 518-565   _log_Pse + row entropy + row logsumexp of the dense N x N
//                                         log-affinity inside the dual-ascent loop of SymmetricEntropicAffinity
//   affinity/entropic.py:45-48, 733-748   symmetric log-domain Sinkhorn iterations (student base kernel)
//   neighbor_embedding/tsnekhorn.py:210-230  loss  CE(P, log Q) + sum(Q)  whose gradient w.r.t. the embedding is
//                                         4 * sum_j (P_ij - Q_ij) / (1 + d_ij) * (z_i - z_j)   (duals detached)
//
// The reference materialises N x N matrices (160 GB each at N = 200k).  Here nothing of size N^2 exists:
// the pairwise squared distances are recomputed tile by tile on the fp32 MFMA pipe from the packed point
// images (same operand layout, same k-ordered FMA chain as the kNN scan in tdr_knn.hip), and each pass
// reduces the tile on the fly -- a lane owns one row i of the affinity matrix and folds its 16 columns
// per tile into running sums held in registers (flash-attention style streaming reductions).
#include "tdr_common.h"

namespace tdr {

struct PairScanParams {
    const float* qp;       // packed rows (queries)
    const float* yp;       // packed columns (database) -- same object for the symmetric passes
    int64_t nq, q_offset, n_db;
    int n_db_tiles;
    const float* side;     // (n_db, SIDE) per-column scalars, row-major
    const float* qside;    // (nq, SIDE) per-row scalars
    float c0, c1;          // epilogue constants
    float diag_add;        // added to C[i][i] when exclude_diag (distance/torch.py:111-116)
    int exclude_diag;
    float* out0;           // per-row outputs
    float* out1;
    float* out2;           // optional third output (SeaStats: sum_j P_ij C_ij), may be NULL
    // database split (launch_pair_scan): workgroup b scans segment b / n_qblocks of the tiles for query block b % n_qblocks and
    // saves its running statistics to part[(seg * nq + row) * NSTATE ..]; pair_scan_merge_kernel folds the segments in order
    int n_seg, tiles_per_seg;
    int64_t n_qblocks;
    float* part;
};

// ---- epilogues ---------------------------------------------------------------------------------------
// SEA row statistics: lp_ij = (mu_i + mu_j - 2 C_ij) / (e_i + e_j); outputs P_sum_i = sum_j exp(lp_ij) and
// H_i = -sum_j exp(lp_ij) (lp_ij - 1)   (entropic.py:522-525; P is NOT normalised inside the entropy).
struct SeaStats {
    static constexpr int SIDE = 2;  // mu, e (= eps^2 or eps)
    // Round 6: the three running sums are COMPENSATED (Kahan): a row of C5 adds 200 000 terms, and plain fp32 accumulation left the
    // row sums / entropies 1.6e-5 / 1.05e-5 from a float64 evaluation (north_star asks 1e-5).  The four pairs a lane meets per
    // quarter tile are summed on their own first (one rescale to the running maximum per quarter instead of a test per pair) and
    // enter the running sums with one compensated addition each: ~3 vector instructions per pair more than the plain form, less
    // the per-pair maximum test.
    static constexpr bool BLOCK4 = true;
    float mu_i, e_i, m, s, t, u;   // u = sum_j exp(lp_ij - m) C_ij (the energy term of the dual objective, entropic.py:487)
    float cs, ct, cu;               // compensation terms of s, t, u
    __device__ __forceinline__ void init(const float* qs) {
        mu_i = qs[0]; e_i = qs[1]; m = -__builtin_inff(); s = 0.f; t = 0.f; u = 0.f; cs = 0.f; ct = 0.f; cu = 0.f;
    }
    static __device__ __forceinline__ void kahan(float& sum, float& comp, float x) {
        const float y = __fsub_rn(x, comp);
        const float tt = __fadd_rn(sum, y);
        comp = __fsub_rn(__fsub_rn(tt, sum), y);
        sum = tt;
    }
    __device__ __forceinline__ void rescale_to(float mm) {
        const float sc = __expf(m - mm);      // m = -inf: 0 (the sums are 0 then)
        s *= sc; t *= sc; u *= sc; cs *= sc; ct *= sc; cu *= sc; m = mm;
    }
    __device__ __forceinline__ void add(float c, const float* sj, const PairScanParams&) {
        const float lp = (mu_i + sj[0] - 2.0f * c) * __builtin_amdgcn_rcpf(e_i + sj[1]);
        if (lp > m) rescale_to(lp);
        const float p = __expf(lp - m);
        kahan(s, cs, p);
        kahan(t, ct, p * lp);
        if (p > 0.f) kahan(u, cu, p * c);   // the excluded diagonal carries c = 1e12 and p = 0
    }
    // four pairs of an interior tile (no diagonal, no padding rows)
    __device__ __forceinline__ void add4(const float (&c)[4], const float (&sj)[4][SIDE], const PairScanParams&) {
        float lp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lp[e] = (mu_i + sj[e][0] - 2.0f * c[e]) * __builtin_amdgcn_rcpf(e_i + sj[e][1]);
        const float mx = fmaxf(fmaxf(lp[0], lp[1]), fmaxf(lp[2], lp[3]));
        if (mx > m) rescale_to(mx);
        float ls = 0.f, lt = 0.f, lu = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = __expf(lp[e] - m);
            ls += p;
            lt = fmaf(p, lp[e], lt);
            lu = fmaf(p, c[e], lu);
        }
        kahan(s, cs, ls);
        kahan(t, ct, lt);
        kahan(u, cu, lu);
    }
    __device__ __forceinline__ void merge(const SeaStats& o) {
        const float mm = fmaxf(m, o.m);
        const float a = (m == -__builtin_inff()) ? 0.f : __expf(m - mm);
        const float b = (o.m == -__builtin_inff()) ? 0.f : __expf(o.m - mm);
        s = s * a + o.s * b; t = t * a + o.t * b; u = u * a + o.u * b; m = mm;
        cs = 0.f; ct = 0.f; cu = 0.f;
    }
    __device__ __forceinline__ void shfl_from(const SeaStats& x, int src) {
        m = __shfl(x.m, src, 64); s = __shfl(x.s, src, 64); t = __shfl(x.t, src, 64); u = __shfl(x.u, src, 64); mu_i = x.mu_i; e_i = x.e_i;
        cs = 0.f; ct = 0.f; cu = 0.f;
    }
    static constexpr int NSTATE = 4;
    __device__ __forceinline__ void save(float* p) const { p[0] = m; p[1] = s; p[2] = t; p[3] = u; }
    __device__ __forceinline__ void load(const float* p) { m = p[0]; s = p[1]; t = p[2]; u = p[3]; cs = 0.f; ct = 0.f; cu = 0.f; }
    __device__ __forceinline__ void store(int64_t row, const PairScanParams& P) const {
        const float em = expf(m);
        const float S = em * s, T = em * t;
        P.out0[row] = S;
        P.out1[row] = -(T - S);
        if (P.out2) P.out2[row] = em * u;
    }
};

// Symmetric Sinkhorn on a Gaussian base kernel (entropic.py:728-734): row log-sum-exp of log K_ij + f_j with
// log K = -C / eps (student base: -log(1 + C) / eps); out0[i] = LSE_j(...).  side = (f).
struct SinkLse {
    static constexpr int SIDE = 1;
    float m, s;
    __device__ __forceinline__ void init(const float*) { m = -__builtin_inff(); s = 0.f; }
    __device__ __forceinline__ void add(float c, const float* sj, const PairScanParams& P) {
        const float base = P.c1 != 0.f ? __logf(1.0f + c) : c;   // c1 != 0: student base kernel
        const float lp = sj[0] - base * P.c0;                     // c0 = 1 / eps
        if (lp > m) { s *= __expf(m - lp); m = lp; }
        s += __expf(lp - m);
    }
    __device__ __forceinline__ void merge(const SinkLse& o) {
        const float mm = fmaxf(m, o.m);
        const float a = (m == -__builtin_inff()) ? 0.f : __expf(m - mm);
        const float b = (o.m == -__builtin_inff()) ? 0.f : __expf(o.m - mm);
        s = s * a + o.s * b; m = mm;
    }
    __device__ __forceinline__ void shfl_from(const SinkLse& x, int src) { m = __shfl(x.m, src, 64); s = __shfl(x.s, src, 64); }
    static constexpr int NSTATE = 2;
    __device__ __forceinline__ void save(float* p) const { p[0] = m; p[1] = s; }
    __device__ __forceinline__ void load(const float* p) { m = p[0]; s = p[1]; }
    __device__ __forceinline__ void store(int64_t row, const PairScanParams& P) const { P.out0[row] = m + logf(s); }
};

// TSNEkhorn force: g_i = 4 * sum_j (P_ij - Q_ij) w_ij (z_i - z_j), w = 1/(1+|z_i-z_j|^2),
//   P_ij = exp((mu_i+mu_j-2C_ij)/(e_i+e_j) - log N),  Q_ij = E_i E_j w_ij / N  with E = exp(dual)
//   (the diagonal term vanishes with z_i - z_i).  side = (mu, e, z[NC], E); out0 = grad (n, NC).
//   Instances for NC = 2, 3 and the zero-padded widths 4 / 8 / 16 / 32 (the caller pads z; padded components add nothing).
template <int NC>
struct KhornForce {
    static constexpr int SIDE = 3 + NC;
    float mu_i, e_i, z[NC], E_i, g[NC];
    __device__ __forceinline__ void init(const float* qs) {
        mu_i = qs[0]; e_i = qs[1]; E_i = qs[2 + NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = qs[2 + c]; g[c] = 0.f; }
    }
    __device__ __forceinline__ void add(float c, const float* sj, const PairScanParams& P) {
        const float lp = (mu_i + sj[0] - 2.0f * c) * __builtin_amdgcn_rcpf(e_i + sj[1]) - P.c0;  // c0 = log N
        const float p = __expf(lp);
        float df[NC], d2 = 1.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k) df[k] = z[k] - sj[2 + k];
        if (NC == 2) d2 = 1.0f + df[0] * df[0] + df[1] * df[1];
        else {
#pragma unroll
            for (int k = 0; k < NC; ++k) d2 = fmaf(df[k], df[k], d2);
        }
        const float w = __builtin_amdgcn_rcpf(d2);
        const float q = E_i * sj[2 + NC] * w * P.c1;  // c1 = 1/N
        const float coef = (p - q) * w;
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = fmaf(coef, df[k], g[k]);
    }
    __device__ __forceinline__ void merge(const KhornForce& o) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] += o.g[k];
    }
    __device__ __forceinline__ void shfl_from(const KhornForce& x, int src) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = __shfl(x.g[k], src, 64);
    }
    static constexpr int NSTATE = NC;
    __device__ __forceinline__ void save(float* p) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) p[k] = g[k];
    }
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = p[k];
    }
    __device__ __forceinline__ void store(int64_t row, const PairScanParams& P) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) P.out0[row * NC + k] = 4.0f * g[k];
    }
};

// TSNEkhorn force with `unrolling=True` (tsnekhorn.py:134, 224-227: loss = CE(P, log Q) alone, autograd THROUGH the K <= 5
// symmetric Sinkhorn updates of entropic.py:733-736 that produce the dual from the detached warm start).  In closed form:
// with w = 1/(1+d), Ef^k = exp(f^{k-1} - max f^{k-1}), s^k_i = sum_j Ef^k_j w_ij (what update k reduces),
// S^k_ij = w_ij Ef^k_j / s^k_i its softmax, and the adjoints  g^K = -(rowsum P + colsum P),  g^{k-1} = (g^k - S^k^T g^k) / 2,
//   grad_i = sum_j [ 4 P_ij + w_ij sum_k (g^k_i S^k_ij + g^k_j S^k_ji) / w_ij ] w_ij (z_i - z_j)
//          = 4 sum_j [ P_ij + w_ij sum_k (a^k_i b^k_j + a^k_j b^k_i) ] w_ij (z_i - z_j),   a^k = g^k / (4 s^k),  b^k = Ef^k
// -- the Q term of KhornForce replaced by a rank-2K bilinear form of per-point vectors (the caller runs the K forward
// updates and the K adjoint mat-vecs, tdr_student_matvec_f32, and pads a / b with zeros when the updates stopped early).
// side = (mu, e, z[NC], a[5], b[5]).
template <int NC>
struct KhornForceUnrolled {
    static constexpr int KU = 5;
    static constexpr int SIDE = 2 + NC + 2 * KU;
    float mu_i, e_i, z[NC], a[KU], b[KU], g[NC];
    __device__ __forceinline__ void init(const float* qs) {
        mu_i = qs[0]; e_i = qs[1];
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = qs[2 + c]; g[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < KU; ++k) { a[k] = qs[2 + NC + k]; b[k] = qs[2 + NC + KU + k]; }
    }
    __device__ __forceinline__ void add(float c, const float* sj, const PairScanParams& P) {
        const float lp = (mu_i + sj[0] - 2.0f * c) * __builtin_amdgcn_rcpf(e_i + sj[1]) - P.c0;  // c0 = log N
        const float p = __expf(lp);
        float df[NC], d2 = 1.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k) { df[k] = z[k] - sj[2 + k]; d2 = fmaf(df[k], df[k], d2); }
        const float w = __builtin_amdgcn_rcpf(d2);
        float bil = 0.f;
#pragma unroll
        for (int k = 0; k < KU; ++k) bil = fmaf(a[k], sj[2 + NC + KU + k], fmaf(sj[2 + NC + k], b[k], bil));
        const float coef = fmaf(w, bil, p) * w;
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = fmaf(coef, df[k], g[k]);
    }
    __device__ __forceinline__ void merge(const KhornForceUnrolled& o) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] += o.g[k];
    }
    __device__ __forceinline__ void shfl_from(const KhornForceUnrolled& x, int src) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = __shfl(x.g[k], src, 64);
    }
    static constexpr int NSTATE = NC;
    __device__ __forceinline__ void save(float* p) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) p[k] = g[k];
    }
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int k = 0; k < NC; ++k) g[k] = p[k];
    }
    __device__ __forceinline__ void store(int64_t row, const PairScanParams& P) const {
#pragma unroll
        for (int k = 0; k < NC; ++k) P.out0[row * NC + k] = 4.0f * g[k];
    }
};

typedef __attribute__((address_space(1))) const void* dgptr_t;
typedef __attribute__((address_space(3))) void* dlptr_t;

// an epilogue that takes the four pairs of a quarter tile at once (SeaStats) says so with `static constexpr bool BLOCK4 = true`
template <class E, class = void> struct epi_block4 { static constexpr bool value = false; };
template <class E> struct epi_block4<E, decltype((void)E::BLOCK4)> { static constexpr bool value = E::BLOCK4; };

// One tile step, software pipelined like the kNN scan (tdr_knn.hip): the MFMA chain of tile T runs with the
// row reduction of tile T-1 (held in `accp`) placed BETWEEN its MFMAs, so the exp-heavy epilogue executes in
// the shadow of the 64-cycle matrix instructions.  Quarter `g` of the previous tile = its rows 8g+4h .. +3.
// `edge` (uniform over the wavefront): the tile holds the diagonal of one of the wavefront's rows or database rows beyond
// n_db -- every other tile takes the lean loop (no 64-bit index compare, no diagonal select, no bounds mask per pair; the
// epilogue's vector instructions share the issue port with the MFMAs and bound the scan together with them).
template <class Epi>
__device__ __forceinline__ void reduce_part(Epi& epi, const PairScanParams& P, const f32x16& accp, const float* ynp,
                                            const float* sd, int g, int h, float xn, int64_t row_base, int64_t gq, bool edge) {
    constexpr int SIDE = Epi::SIDE;
    const f32x4 y4 = *reinterpret_cast<const f32x4*>(ynp + 8 * g);
    if (!edge) {
        if constexpr (epi_block4<Epi>::value) {
            float c4[4], sj4[4][SIDE];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = e + 8 * g + 4 * h;  // database row inside the tile
                c4[e] = __builtin_fmaf(-2.0f, accp[4 * g + e], __fadd_rn(xn, y4[e]));
#pragma unroll
                for (int s_ = 0; s_ < SIDE; ++s_) sj4[e][s_] = sd[i * SIDE + s_];
            }
            epi.add4(c4, sj4, P);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = e + 8 * g + 4 * h;  // database row inside the tile
            const float c = __builtin_fmaf(-2.0f, accp[4 * g + e], __fadd_rn(xn, y4[e]));
            float sj[SIDE];
#pragma unroll
            for (int s_ = 0; s_ < SIDE; ++s_) sj[s_] = sd[i * SIDE + s_];
            epi.add(c, sj, P);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int i = e + 8 * g + 4 * h;  // database row inside the tile
        const int64_t j = row_base + e + 8 * g;
        float c = __builtin_fmaf(-2.0f, accp[r], __fadd_rn(xn, y4[e]));
        if (P.exclude_diag && j == gq + P.q_offset) c = __fadd_rn(c, P.diag_add);
        if (j < P.n_db) {
            float sj[SIDE];
#pragma unroll
            for (int s_ = 0; s_ < SIDE; ++s_) sj[s_] = sd[i * SIDE + s_];
            epi.add(c, sj, P);
        }
    }
}

template <int KQ, class Epi, bool HAVE_PREV>
__device__ __forceinline__ void pair_tile_step(Epi& epi, const PairScanParams& P, const float* __restrict__ img,
                                               const float (&b)[4 * KQ], f32x16& acc, const f32x16& accp,
                                               const float* ynp_prev, const float* sd_prev, int lane, int h, float xn,
                                               int64_t row_base_prev, int64_t gq, bool edge_prev) {
    constexpr int GQ = (KQ >= 4) ? 4 : KQ, NG = KQ / GQ;
    constexpr int PARTS_PER_GROUP = (4 + NG - 1) / NG;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = img + lane * 4;
    f32x4 a0[GQ], a1[GQ];
#pragma unroll
    for (int u = 0; u < GQ; ++u) a0[u] = *reinterpret_cast<const f32x4*>(ap + u * 256);
    int part = 0;
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) {
#pragma unroll
            for (int u = 0; u < GQ; ++u) a1[u] = *reinterpret_cast<const f32x4*>(ap + ((g + 1) * GQ + u) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (HAVE_PREV) {
#pragma unroll
            for (int pp = 0; pp < PARTS_PER_GROUP; ++pp)
                if (part + pp < 4) reduce_part<Epi>(epi, P, accp, ynp_prev, sd_prev, part + pp, h, xn, row_base_prev, gq, edge_prev);
        }
        part += PARTS_PER_GROUP;
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int t = g * GQ + u;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][0], b[4 * t + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][1], b[4 * t + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][2], b[4 * t + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][3], b[4 * t + 3], acc, 0, 0, 0);
        }
        if (g + 1 < NG) {
            if (g + 2 < NG) {
#pragma unroll
                for (int u = 0; u < GQ; ++u) a0[u] = *reinterpret_cast<const f32x4*>(ap + ((g + 2) * GQ + u) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HAVE_PREV) {
#pragma unroll
                for (int pp = 0; pp < PARTS_PER_GROUP; ++pp)
                    if (part + pp < 4) reduce_part<Epi>(epi, P, accp, ynp_prev, sd_prev, part + pp, h, xn, row_base_prev, gq, edge_prev);
            }
            part += PARTS_PER_GROUP;
#pragma unroll
            for (int u = 0; u < GQ; ++u) {
                const int t = (g + 1) * GQ + u;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][0], b[4 * t + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][1], b[4 * t + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][2], b[4 * t + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][3], b[4 * t + 3], acc, 0, 0, 0);
            }
        }
    }
}

// Four workgroups per CU (<= 128 VGPRs) for the statistics scans at D <= 64 -- measured at N = 200k, D = 64 (SeaStats):
// 60.4 ms per launch at three wavefronts per SIMD, 58.0 ms at four (two spilled registers); the force functors carry the
// embedding coordinates and would spill heavily, they stay at two.
#ifndef TDR_PAIR_MINBLK
#define TDR_PAIR_MINBLK 4
#endif
template <int KQ, class Epi>
__global__ __launch_bounds__(256, (KQ <= 8 && Epi::SIDE <= 2) ? TDR_PAIR_MINBLK : 2) void pair_scan_kernel(const PairScanParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TILE_F = KQ * 256 + 64;
    constexpr int IMG_F = KQ * 256;
    constexpr int SIDE = Epi::SIDE;
    constexpr int SIDE_SLOT = 32 * SIDE;
    float* tile0 = reinterpret_cast<float*>(smem_raw);
    float* tile1 = tile0 + IMG_F;
    float* nring = tile1 + IMG_F;       // [4 slots][64]  squared norms of the tile's rows
    float* sring = nring + 4 * 64;      // [4 slots][32][SIDE] per-column scalars
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 31, h = lane >> 5;
    const int64_t n_qtiles = (P.nq + 31) / 32;
    const int seg = (int)((int64_t)blockIdx.x / P.n_qblocks);        // segment-major: concurrent workgroups stream the same tiles
    const int64_t qblock = (int64_t)blockIdx.x - (int64_t)seg * P.n_qblocks;
    const int64_t qt = qblock * 4 + wave;
    const bool wave_active = qt < n_qtiles;
    const int64_t gq = qt * 32 + q;
    const bool lane_valid = wave_active && gq < P.nq;

    float b[4 * KQ];
    float xn = 0.f;
    Epi epi;
    {
        float qs[SIDE];
#pragma unroll
        for (int c = 0; c < SIDE; ++c) qs[c] = lane_valid ? P.qside[(size_t)gq * SIDE + c] : 1.0f;
        epi.init(qs);
    }
    if (wave_active) {
        const float* qimg = P.qp + (size_t)qt * TILE_F;
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qimg + t * 256 + lane * 4);
            b[4 * t + 0] = v[0]; b[4 * t + 1] = v[1]; b[4 * t + 2] = v[2]; b[4 * t + 3] = v[3];
        }
        xn = qimg[KQ * 256 + q];
    } else {
#pragma unroll
        for (int t = 0; t < 4 * KQ; ++t) b[t] = 0.f;
    }

    const int T0 = seg * P.tiles_per_seg;
    const int n_tiles = (T0 + P.tiles_per_seg < P.n_db_tiles) ? T0 + P.tiles_per_seg : P.n_db_tiles;   // this segment: [T0, n_tiles)
    // stage(T): image -> tile[T & 1] and norms -> nring[T & 3] by LDS-DMA; side scalars -> sring[T & 3] (the
    // caller stores `sreg` right before the barrier, when the DMA has long landed)
    constexpr int SREG = (SIDE_SLOT + 255) / 256;   // side scalars of a tile held per thread between the load and the store
    float sreg[SREG];
    auto stage = [&](int T) {
        const float* src = P.yp + (size_t)T * TILE_F;
        float* dst = (T & 1) ? tile1 : tile0;
#pragma unroll
        for (int t = 0; t < KQ; t += 4) {
            const int blk = t + wave;
            if (blk < KQ)
                __builtin_amdgcn_global_load_lds((dgptr_t)(src + blk * 256 + lane * 4), (dlptr_t)(dst + blk * 256), 16, 0, 0);
        }
        if (wave == 3)
            __builtin_amdgcn_global_load_lds((dgptr_t)(src + KQ * 256 + lane), (dlptr_t)(nring + (T & 3) * 64), 4, 0, 0);
#pragma unroll
        for (int u = 0; u < SREG; ++u) {
            const int o = u * 256 + tid;
            if (o < SIDE_SLOT) {
                const int64_t idx = (int64_t)T * SIDE_SLOT + o;
                sreg[u] = (idx < P.n_db * SIDE) ? P.side[idx] : 0.f;
            }
        }
    };
    auto stage_side_store = [&](int T) {
#pragma unroll
        for (int u = 0; u < SREG; ++u) {
            const int o = u * 256 + tid;
            if (o < SIDE_SLOT) sring[(T & 3) * SIDE_SLOT + o] = sreg[u];
        }
    };
    if (n_tiles > T0) { stage(T0); stage_side_store(T0); }
    __syncthreads();

    // tiles that need the careful epilogue: the last one when n_db is ragged, and the one(s) holding the wavefront's diagonal
    const int64_t dq0 = qt * 32 + P.q_offset;
    auto is_edge = [&](int Tp) -> bool {
        const int64_t r0 = (int64_t)Tp * 32;
        return (r0 + 32 > P.n_db) || (P.exclude_diag && r0 < dq0 + 32 && r0 + 32 > dq0);
    };
    f32x16 accA, accB;
    int T = T0;
    if (T < n_tiles) {
        const bool nx = T + 1 < n_tiles;
        if (nx) stage(T + 1);
        if (wave_active)
            pair_tile_step<KQ, Epi, false>(epi, P, (T & 1) ? tile1 : tile0, b, accA, accA, nring, sring, lane, h, xn, 0, gq, false);
        if (nx) stage_side_store(T + 1);
        __syncthreads();
        ++T;
    }
    while (T < n_tiles) {
        {
            const bool nx = T + 1 < n_tiles;
            if (nx) stage(T + 1);
            if (wave_active)
                pair_tile_step<KQ, Epi, true>(epi, P, (T & 1) ? tile1 : tile0, b, accB, accA,
                                              nring + ((T - 1) & 3) * 64 + 4 * h, sring + ((T - 1) & 3) * SIDE_SLOT, lane, h,
                                              xn, (int64_t)(T - 1) * 32 + 4 * h, gq, is_edge(T - 1));
            if (nx) stage_side_store(T + 1);
            __syncthreads();
            ++T;
        }
        if (T < n_tiles) {
            const bool nx = T + 1 < n_tiles;
            if (nx) stage(T + 1);
            if (wave_active)
                pair_tile_step<KQ, Epi, true>(epi, P, (T & 1) ? tile1 : tile0, b, accA, accB,
                                              nring + ((T - 1) & 3) * 64 + 4 * h, sring + ((T - 1) & 3) * SIDE_SLOT, lane, h,
                                              xn, (int64_t)(T - 1) * 32 + 4 * h, gq, is_edge(T - 1));
            if (nx) stage_side_store(T + 1);
            __syncthreads();
            ++T;
        } else {
            accA = accB;
        }
    }
    if (wave_active && n_tiles > T0) {
        const int Tl = n_tiles - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            reduce_part<Epi>(epi, P, accA, nring + (Tl & 3) * 64 + 4 * h, sring + (Tl & 3) * SIDE_SLOT, g, h, xn,
                             (int64_t)Tl * 32 + 4 * h, gq, is_edge(Tl));
        // combine the two lanes (h = 0, 1) that share a row
        Epi other;
        other.shfl_from(epi, lane ^ 32);
        epi.merge(other);
        if (h == 0 && lane_valid) {
            if (P.n_seg == 1) epi.store(gq, P);
            else epi.save(P.part + ((size_t)seg * P.nq + gq) * Epi::NSTATE);
        }
    }
}

// folds the per-segment statistics of a row in segment order (a fixed association: the result does not depend on scheduling)
template <class Epi>
__global__ __launch_bounds__(256) void pair_scan_merge_kernel(const PairScanParams P) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= P.nq) return;
    Epi epi, other;
    epi.load(P.part + (size_t)row * Epi::NSTATE);
    for (int sgm = 1; sgm < P.n_seg; ++sgm) {
        other.load(P.part + ((size_t)sgm * P.nq + row) * Epi::NSTATE);
        epi.merge(other);
    }
    epi.store(row, P);
}

// ---- Sinkhorn pass on the 2-D / 3-D embedding (student kernel), entropic.py:733-740 --------------------
//   red_j = -LSE_i(log K_ij + f_i),  log K_ij = -log(1 + d_ij) / eps,  d_ii += 1e12 when zero_diag
// With eps == 1:  red_j = -( fmax + log sum_i exp(f_i - fmax) / (1 + d_ij) )  -- no per-pair transcendental.
// out[j] = 0.5 * (f_j + red_j)  (the averaged update), resid2 += (out[j] - red_j)^2  (convergence test :738).
// s_j = sum_i v_i / (1 + |z_j - z_i|^2) for the thread's row j (all BS threads of the workgroup take part in the staging).
// BS = 256, or 64 when 256-row workgroups would be too few to load the CUs evenly (N = 200k: 782 workgroups on 256 CUs is
// 3.05 per CU -- a quarter of the launch runs with most CUs idle; 3125 single-wavefront workgroups leave a 6 % tail).
// The tile is staged pair-interleaved -- columns (2p, 2p+1) as x0 x1 | y0 y1 | .. | v0 v1 -- so that the distance of two
// columns is formed by packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two results per lane and issue slot); the
// two reciprocals and the two ordered accumulations stay scalar (the sum over i keeps its order: same bits as the one-column
// loop).  Only the tile that holds the workgroup's own rows looks for the diagonal.  Per pair: 30 issue cycles instead of 48.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NC, int BS>
__device__ __forceinline__ float student_weighted_sum(const float* __restrict__ Z, const float* __restrict__ v, int64_t n,
                                                      int64_t j, const float (&zj)[NC], int zero_diag, float diag_add,
                                                      float* tile, int64_t i_lo, int64_t i_hi) {
    constexpr int REC = 2 * (NC + 1);      // floats per column pair
    float s = 0.f;
    f32x2 zz[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) zz[c] = f32x2{zj[c], zj[c]};
    const int64_t own0 = (int64_t)blockIdx.x * BS;
    for (int64_t i0 = i_lo; i0 < i_hi; i0 += BS) {   // i_lo: a multiple of BS
        __syncthreads();
        const int64_t i = i0 + threadIdx.x;
        float* rec = tile + (threadIdx.x >> 1) * REC + (threadIdx.x & 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) rec[2 * c] = (i < i_hi) ? Z[(size_t)i * NC + c] : 0.f;
        rec[2 * NC] = (i < i_hi) ? v[i] : 0.f;
        __syncthreads();
        const int lim = (int)((i_hi - i0 < BS) ? (i_hi - i0) : BS);
        if (zero_diag && i0 == own0) {     // the tile with this workgroup's own rows: one column at a time, diagonal weighted
            for (int t = 0; t < lim; ++t) {
                const float* q = tile + (t >> 1) * REC + (t & 1);
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) { const float df = zj[c] - q[2 * c]; d = fmaf(df, df, d); }
                if ((i0 + t) == j) d += diag_add;
                s = fmaf(q[2 * NC], __builtin_amdgcn_rcpf(1.0f + d), s);
            }
            continue;
        }
        const int lim2 = (lim + 1) >> 1;   // a ragged last pair carries v = 0 in its second column
        for (int p = 0; p < lim2; ++p) {
            const f32x2* q = reinterpret_cast<const f32x2*>(tile + p * REC);
            f32x2 df = zz[0] - q[0];
            f32x2 d = df * df;             // fma(df, df, 0) of the one-column loop
#pragma unroll
            for (int c = 1; c < NC; ++c) { df = zz[c] - q[c]; d = __builtin_elementwise_fma(df, df, d); }
            d = d + 1.0f;
            const f32x2 w = q[NC];
            s = fmaf(w.x, __builtin_amdgcn_rcpf(d.x), s);
            s = fmaf(w.y, __builtin_amdgcn_rcpf(d.y), s);
        }
    }
    return s;
}

template <int NC, int BS>
__global__ __launch_bounds__(BS) void sinkhorn_pass_kernel(const float* __restrict__ Z, const float* __restrict__ f,
                                                           const float* __restrict__ Ef, float fmax, int64_t n,
                                                           int zero_diag, float diag_add, float* __restrict__ f_new,
                                                           float* __restrict__ resid2) {
    __shared__ __attribute__((aligned(16))) float tile[BS * (NC + 1)];
    const int64_t j = (int64_t)blockIdx.x * BS + threadIdx.x;
    const bool have = j < n;
    float zj[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) zj[c] = have ? Z[(size_t)j * NC + c] : 0.f;
    const float s = student_weighted_sum<NC, BS>(Z, Ef, n, j, zj, zero_diag, diag_add, tile, 0, n);
    float r2 = 0.f;
    if (have) {
        const float red = -(fmax + logf(s));
        const float fn = 0.5f * (f[j] + red);
        f_new[j] = fn;
        const float df = fn - red;
        r2 = df * df;
    }
    r2 = wave_sum(r2);
    if ((threadIdx.x & 63) == 0 && r2 != 0.f) atomicAdd(resid2, r2);
}

// out_j = sum_i v_i / (1 + d_ij): the Student-kernel mat-vec of the adjoint Sinkhorn updates (v may be negative).
// blockIdx.y = column segment [y * cols_per_seg, ..): the partial sum goes to plane y of `out` (one plane when unsplit).
template <int NC, int BS>
__global__ __launch_bounds__(BS) void student_matvec_kernel(const float* __restrict__ Z, const float* __restrict__ v, int64_t n,
                                                            int zero_diag, float diag_add, float* __restrict__ out,
                                                            int64_t cols_per_seg) {
    __shared__ __attribute__((aligned(16))) float tile[BS * (NC + 1)];
    const int64_t j = (int64_t)blockIdx.x * BS + threadIdx.x;
    const bool have = j < n;
    float zj[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) zj[c] = have ? Z[(size_t)j * NC + c] : 0.f;
    const int64_t i_lo = (int64_t)blockIdx.y * cols_per_seg;
    const int64_t i_hi = (i_lo + cols_per_seg < n) ? i_lo + cols_per_seg : n;
    const float s = student_weighted_sum<NC, BS>(Z, v, n, j, zj, zero_diag, diag_add, tile, i_lo, i_hi);
    if (have) out[(size_t)blockIdx.y * n + j] = s;
}

// the Sinkhorn update from per-segment partial sums (added in segment order)
__global__ __launch_bounds__(256) void sinkhorn_finish_kernel(const float* __restrict__ planes, int n_planes,
                                                              const float* __restrict__ f, float fmax, int64_t n,
                                                              float* __restrict__ f_new, float* __restrict__ resid2) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float r2 = 0.f;
    if (j < n) {
        float s = planes[j];
        for (int p = 1; p < n_planes; ++p) s += planes[(size_t)p * n + j];
        const float red = -(fmax + logf(s));
        const float fn = 0.5f * (f[j] + red);
        f_new[j] = fn;
        const float df = fn - red;
        r2 = df * df;
    }
    r2 = wave_sum(r2);
    if ((threadIdx.x & 63) == 0 && r2 != 0.f) atomicAdd(resid2, r2);
}

__global__ __launch_bounds__(256) void sum_planes_f32_kernel(const float* __restrict__ planes, int n_planes, int64_t cnt,
                                                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    float a = planes[i];
    for (int p = 1; p < n_planes; ++p) a += planes[(size_t)p * cnt + i];
    out[i] = a;
}

// single-wavefront workgroups until 256-row ones number >= 16 per CU (unsplit launches)
static inline bool student_small_blocks(int64_t n) { return (n + 255) / 256 < 16 * 256; }

// column segments of the all-pairs passes on the embedding: >= ~4096 workgroups of 256 rows, segments of >= 1024 columns
static inline int student_segments(int64_t n) {
    const int64_t row_blocks = (n + 255) / 256;
    int64_t s = (4096 + row_blocks - 1) / row_blocks;
    if (s > n / 1024) s = n / 1024;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

// launches the (possibly segmented) mat-vec into `out` (n_seg planes of n floats); returns the number of planes or an error < 0
static int student_matvec_launch(const float* Z, int nc, const float* v, int64_t n, int zero_diag, float diag_add, float* out,
                                 int n_seg, hipStream_t st) {
    if (n_seg <= 1) {
        const bool small = student_small_blocks(n);
        const unsigned grid = (unsigned)(small ? (n + 63) / 64 : (n + 255) / 256);
#define TDR_MV(NCV)                                                                                                         \
    {                                                                                                                       \
        if (small) hipLaunchKernelGGL((student_matvec_kernel<NCV, 64>), dim3(grid), dim3(64), 0, st, Z, v, n, zero_diag, diag_add, out, n); \
        else hipLaunchKernelGGL((student_matvec_kernel<NCV, 256>), dim3(grid), dim3(256), 0, st, Z, v, n, zero_diag, diag_add, out, n);    \
    }
        switch (nc) {
            case 2: TDR_MV(2); break;
            case 3: TDR_MV(3); break;
            case 4: TDR_MV(4); break;
            case 8: TDR_MV(8); break;
            case 16: TDR_MV(16); break;
            case 32: TDR_MV(32); break;
            default: return TDR_ERR_UNSUPPORTED;
        }
#undef TDR_MV
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 1 : -(int)e - 1000;
    }
    const int64_t cols = (((n + n_seg - 1) / n_seg + 255) / 256) * 256;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)((n + cols - 1) / cols));
#define TDR_MV(NCV) hipLaunchKernelGGL((student_matvec_kernel<NCV, 256>), grid, dim3(256), 0, st, Z, v, n, zero_diag, diag_add, out, cols)
    switch (nc) {
        case 2: TDR_MV(2); break;
        case 3: TDR_MV(3); break;
        case 4: TDR_MV(4); break;
        case 8: TDR_MV(8); break;
        case 16: TDR_MV(16); break;
        case 32: TDR_MV(32); break;
        default: return TDR_ERR_UNSUPPORTED;
    }
#undef TDR_MV
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? (int)grid.y : -(int)e - 1000;
}

static inline int dense_pick_kq(int d) {
    if (d <= 32) return 4;
    if (d <= 64) return 8;
    if (d <= 128) return 16;
    if (d <= 256) return 32;
    return 0;
}

// Database split of a pair scan.  A workgroup serves 128 rows against the tiles it is given; unsplit, N = 200k is 1563
// workgroups for 1024 resident ones (4 per CU): 1.5 rounds, a quarter of the launch with the chip mostly idle -- and below
// N = 32k the grid does not cover the CUs at all.  Split into segments of >= 64 tiles so that there are >= ~16 k workgroups.
static inline int pair_scan_segments(int64_t nq, int n_tiles) {
    const int64_t n_qb = (nq + 127) / 128;
    int64_t s = (16384 + n_qb - 1) / n_qb;
    if (s > n_tiles / 64) s = n_tiles / 64;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
}

template <class Epi>
static int launch_pair_scan(PairScanParams P, int d, hipStream_t st, void* ws, int64_t ws_bytes) {
    const int kq = dense_pick_kq(d);
    if (kq == 0) return TDR_ERR_UNSUPPORTED;
    P.n_qblocks = (P.nq + 127) / 128;
    P.n_seg = 1; P.tiles_per_seg = P.n_db_tiles; P.part = nullptr;
    {
        const int want = pair_scan_segments(P.nq, P.n_db_tiles);
        if (want > 1 && ws && ws_bytes >= (int64_t)want * P.nq * Epi::NSTATE * (int64_t)sizeof(float)) {
            P.tiles_per_seg = (P.n_db_tiles + want - 1) / want;
            P.n_seg = (P.n_db_tiles + P.tiles_per_seg - 1) / P.tiles_per_seg;
            P.part = (float*)ws;
        }
    }
    const size_t lds = (size_t)2 * (kq * 256) * sizeof(float) + (size_t)4 * 64 * sizeof(float) +
                       (size_t)4 * 32 * Epi::SIDE * sizeof(float);
    const unsigned grid = (unsigned)(P.n_qblocks * P.n_seg);
#define TDR_LAUNCH(KQV)                                                                                          \
    {                                                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_scan_kernel<KQV, Epi>),            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        if (e != hipSuccess) return (int)e;                                                                      \
        hipLaunchKernelGGL((pair_scan_kernel<KQV, Epi>), dim3(grid), dim3(256), lds, st, P);                     \
    }
    switch (kq) {
        case 4: TDR_LAUNCH(4) break;
        case 8: TDR_LAUNCH(8) break;
        case 16: TDR_LAUNCH(16) break;
        default: TDR_LAUNCH(32) break;
    }
#undef TDR_LAUNCH
    if (P.n_seg > 1)
        hipLaunchKernelGGL((pair_scan_merge_kernel<Epi>), dim3((unsigned)((P.nq + 255) / 256)), dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* Bytes of the optional workspace of the pair scans below for n points and `n_state` floats of running statistics per row
 * (row statistics 4, log-sum-exp 2, forces: the instance width nc): with it the database is split into segments so that
 * the launch fills the chip evenly (pair_scan_segments); 0 = the scan of this size is not split.  NULL / a smaller buffer
 * runs the unsplit scan. */
int64_t tdr_pair_scan_workspace_bytes(int64_t n, int n_state) {
    if (n <= 0 || n_state <= 0) return 0;
    const int want = pair_scan_segments(n, (int)((n + 31) / 32));
    return want > 1 ? (int64_t)want * n * n_state * (int64_t)sizeof(float) : 0;
}

/* SEA row statistics over the implicit N x N matrix lp_ij = (mu_i + mu_j - 2 C_ij)/(e_i + e_j):
 *   psum[i] = sum_j exp(lp_ij),  ent[i] = -sum_j exp(lp_ij)(lp_ij - 1).
 * packed: tile images of X (tdr_pack_rows_f32); side: (n, 2) row-major (mu, e) with e = eps^2 or eps. */
int tdr_sea_rowstats_f32(const float* packed, int64_t n, int d, const float* side, int exclude_diag, float diag_add,
                         float* psum, float* ent, void* ws, int64_t ws_bytes, void* stream) {
    if (!packed || !side || !psum || !ent || n <= 0) return TDR_ERR_BAD_ARG;
    PairScanParams P;
    P.qp = packed; P.yp = packed; P.nq = n; P.q_offset = 0; P.n_db = n; P.n_db_tiles = (int)((n + 31) / 32);
    P.side = side; P.qside = side; P.c0 = 0.f; P.c1 = 0.f; P.diag_add = diag_add; P.exclude_diag = exclude_diag;
    P.out0 = psum; P.out1 = ent; P.out2 = nullptr;
    return launch_pair_scan<SeaStats>(P, d, (hipStream_t)stream, ws, ws_bytes);
}

/* The same with the third row statistic energy[i] = sum_j exp(lp_ij) C_ij: the dual objective of entropic.py:483-491
 * (the LBFGS path) is -sum(energy) - <e, target - ent> + <mu, psum - 1>. */
int tdr_sea_rowstats3_f32(const float* packed, int64_t n, int d, const float* side, int exclude_diag, float diag_add,
                          float* psum, float* ent, float* energy, void* ws, int64_t ws_bytes, void* stream) {
    if (!packed || !side || !psum || !ent || !energy || n <= 0) return TDR_ERR_BAD_ARG;
    PairScanParams P;
    P.qp = packed; P.yp = packed; P.nq = n; P.q_offset = 0; P.n_db = n; P.n_db_tiles = (int)((n + 31) / 32);
    P.side = side; P.qside = side; P.c0 = 0.f; P.c1 = 0.f; P.diag_add = diag_add; P.exclude_diag = exclude_diag;
    P.out0 = psum; P.out1 = ent; P.out2 = energy;
    return launch_pair_scan<SeaStats>(P, d, (hipStream_t)stream, ws, ws_bytes);
}

/* One reduction of the symmetric Sinkhorn fixed point on the INPUT points (entropic.py:728-734, matrix-free):
 *   lse[i] = LSE_j(log K_ij + f_j),  log K = -C / eps (student != 0: -log(1 + C) / eps), C = squared distances of the packed
 *   points with diag_add on the diagonal when exclude_diag.  The caller forms f <- 0.5 (f - lse). */
int tdr_sinkhorn_lse_f32(const float* packed, int64_t n, int d, const float* f, float inv_eps, int student, int exclude_diag,
                         float diag_add, float* lse, void* ws, int64_t ws_bytes, void* stream) {
    if (!packed || !f || !lse || n <= 0 || !(inv_eps > 0.f)) return TDR_ERR_BAD_ARG;
    PairScanParams P;
    P.qp = packed; P.yp = packed; P.nq = n; P.q_offset = 0; P.n_db = n; P.n_db_tiles = (int)((n + 31) / 32);
    P.side = f; P.qside = f; P.c0 = inv_eps; P.c1 = student ? 1.0f : 0.f; P.diag_add = diag_add; P.exclude_diag = exclude_diag;
    P.out0 = lse; P.out1 = nullptr; P.out2 = nullptr;
    return launch_pair_scan<SinkLse>(P, d, (hipStream_t)stream, ws, ws_bytes);
}

/* TSNEkhorn embedding gradient (n, nc): 4 sum_j (P_ij - Q_ij)/(1+d_ij) (z_i - z_j).  nc in {2, 3, 4, 8, 16, 32} (wider
 * instances serve any width below them: the caller pads z with zeros and drops the padded gradient columns).
 * side: (n, 3 + nc) row-major (mu, e, z_0 .. z_{nc-1}, exp(dual)); log_n = log(n). */
int tdr_khorn_grad_nc_f32(const float* packed, int64_t n, int d, const float* side, int nc, float log_n, float* grad,
                          void* ws, int64_t ws_bytes, void* stream) {
    if (!packed || !side || !grad || n <= 0) return TDR_ERR_BAD_ARG;
    PairScanParams P;
    P.qp = packed; P.yp = packed; P.nq = n; P.q_offset = 0; P.n_db = n; P.n_db_tiles = (int)((n + 31) / 32);
    P.side = side; P.qside = side; P.c0 = log_n; P.c1 = 1.0f / (float)n; P.diag_add = 0.f; P.exclude_diag = 0;
    P.out0 = grad; P.out1 = nullptr; P.out2 = nullptr;
    switch (nc) {
        case 2: return launch_pair_scan<KhornForce<2>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 3: return launch_pair_scan<KhornForce<3>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 4: return launch_pair_scan<KhornForce<4>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 8: return launch_pair_scan<KhornForce<8>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 16: return launch_pair_scan<KhornForce<16>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 32: return launch_pair_scan<KhornForce<32>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        default: return TDR_ERR_UNSUPPORTED;
    }
}

/* The same for TSNEkhorn(unrolling=True) (tsnekhorn.py:224-227; KhornForceUnrolled above): 4 sum_j [P_ij + w_ij sum_k
 * (a^k_i b^k_j + a^k_j b^k_i)] w_ij (z_i - z_j).  side: (n, 2 + nc + 10) row-major (mu, e, z_0 .. z_{nc-1}, a^1..a^5, b^1..b^5). */
int tdr_khorn_grad_unrolled_f32(const float* packed, int64_t n, int d, const float* side, int nc, float log_n, float* grad,
                                void* ws, int64_t ws_bytes, void* stream) {
    if (!packed || !side || !grad || n <= 0) return TDR_ERR_BAD_ARG;
    PairScanParams P;
    P.qp = packed; P.yp = packed; P.nq = n; P.q_offset = 0; P.n_db = n; P.n_db_tiles = (int)((n + 31) / 32);
    P.side = side; P.qside = side; P.c0 = log_n; P.c1 = 0.f; P.diag_add = 0.f; P.exclude_diag = 0;
    P.out0 = grad; P.out1 = nullptr; P.out2 = nullptr;
    switch (nc) {
        case 2: return launch_pair_scan<KhornForceUnrolled<2>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 3: return launch_pair_scan<KhornForceUnrolled<3>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 4: return launch_pair_scan<KhornForceUnrolled<4>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 8: return launch_pair_scan<KhornForceUnrolled<8>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 16: return launch_pair_scan<KhornForceUnrolled<16>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        case 32: return launch_pair_scan<KhornForceUnrolled<32>>(P, d, (hipStream_t)stream, ws, ws_bytes);
        default: return TDR_ERR_UNSUPPORTED;
    }
}

/* The two-component form (side: (n, 5)). */
int tdr_khorn_grad_f32(const float* packed, int64_t n, int d, const float* side, float log_n, float* grad, void* stream) {
    return tdr_khorn_grad_nc_f32(packed, n, d, side, 2, log_n, grad, nullptr, 0, stream);
}

/* Bytes of the optional workspace of the two all-pairs passes on the embedding below: with it the columns are spread over
 * several workgroups per block of 256 rows (one workgroup per row block is too coarse below ~1M rows: N = 200k is 3 per CU);
 * 0 = a pass of this size is not split. */
int64_t tdr_student_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const int n_seg = student_segments(n);
    return n_seg > 1 ? (int64_t)n_seg * n * (int64_t)sizeof(float) : 0;
}

/* One symmetric Sinkhorn update on the embedding Z (n, nc), student kernel, eps = 1:
 *   f_new = 0.5 (f + red), red_j = -LSE_i(-log(1 + d_ij) + f_i);  *resid2 (device, caller-zeroed) += |f_new - red|^2.
 * Ef = exp(f - fmax) precomputed by the caller (fmax = max f).  ws / ws_bytes: tdr_student_workspace_bytes (NULL: unsplit). */
int tdr_sinkhorn_pass_f32(const float* Z, int nc, const float* f, const float* Ef, float fmax, int64_t n, int zero_diag,
                          float diag_add, float* f_new, float* resid2, void* ws, int64_t ws_bytes, void* stream) {
    if (!Z || !f || !Ef || !f_new || !resid2 || n <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_seg = student_segments(n);
    if (n_seg > 1 && ws && ws_bytes >= (int64_t)n_seg * n * (int64_t)sizeof(float)) {
        const int planes = student_matvec_launch(Z, nc, Ef, n, zero_diag, diag_add, (float*)ws, n_seg, st);
        if (planes <= 0) return planes == TDR_ERR_UNSUPPORTED ? planes : -planes - 1000;
        hipLaunchKernelGGL(sinkhorn_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, planes, f,
                           fmax, n, f_new, resid2);
        TDR_CHECK_LAUNCH();
        return TDR_OK;
    }
    const bool small = student_small_blocks(n);
    const unsigned grid = (unsigned)(small ? (n + 63) / 64 : (n + 255) / 256);
#define TDR_SK(NCV)                                                                                                         \
    {                                                                                                                       \
        if (small) hipLaunchKernelGGL((sinkhorn_pass_kernel<NCV, 64>), dim3(grid), dim3(64), 0, st, Z, f, Ef, fmax, n, zero_diag, diag_add, f_new, resid2); \
        else hipLaunchKernelGGL((sinkhorn_pass_kernel<NCV, 256>), dim3(grid), dim3(256), 0, st, Z, f, Ef, fmax, n, zero_diag, diag_add, f_new, resid2);    \
    }
    switch (nc) {
        case 2: TDR_SK(2); break;
        case 3: TDR_SK(3); break;
        case 4: TDR_SK(4); break;
        case 8: TDR_SK(8); break;
        case 16: TDR_SK(16); break;
        case 32: TDR_SK(32); break;
        default: return TDR_ERR_UNSUPPORTED;
    }
#undef TDR_SK
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* out_j = sum_i v_i / (1 + |z_i - z_j|^2) over the embedding Z (n, nc), the diagonal term weighted 1 / (1 + diag_add) when
 * zero_diag: the Student-kernel mat-vec of the adjoint Sinkhorn updates (entropic.py:733-736 differentiated; v signed).
 * ws / ws_bytes: tdr_student_workspace_bytes (NULL: unsplit). */
int tdr_student_matvec_f32(const float* Z, int nc, const float* v, int64_t n, int zero_diag, float diag_add, float* out,
                           void* ws, int64_t ws_bytes, void* stream) {
    if (!Z || !v || !out || n <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_seg = student_segments(n);
    if (n_seg > 1 && ws && ws_bytes >= (int64_t)n_seg * n * (int64_t)sizeof(float)) {
        const int planes = student_matvec_launch(Z, nc, v, n, zero_diag, diag_add, (float*)ws, n_seg, st);
        if (planes <= 0) return planes == TDR_ERR_UNSUPPORTED ? planes : -planes - 1000;
        hipLaunchKernelGGL(sum_planes_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, planes, n, out);
        TDR_CHECK_LAUNCH();
        return TDR_OK;
    }
    const int planes = student_matvec_launch(Z, nc, v, n, zero_diag, diag_add, out, 1, st);
    if (planes <= 0) return planes == TDR_ERR_UNSUPPORTED ? planes : -planes - 1000;
    return TDR_OK;
}

}  // extern "C"
