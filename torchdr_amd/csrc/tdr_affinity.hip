// K2 / K3 -- per-row root searches for the entropic and UMAP affinities, plus the gathered
// (indexed) squared distances used by the embedding loop.
//
// Replaces (citations under /root/reference/torchdr):
//   utils/root_search.py:17-77    binary_search   (tol 1e-6 on |f(m)|, masked in-place updates)
//   utils/root_search.py:147-198  init_bounds     (halve b while f(b) > 0, double e while f(e) < 0)
//   affinity/knn_normalized.py:445-465  UMAP:     f(eps) = exp(LSE_j(-(C_ij - rho_i)/eps)) - log2(k)
//   affinity/entropic.py:272-310        entropic: f(eps) = H(log_softmax(-C_i/eps)) - (log(perp) + 1)
//   affinity/entropic.py:96-113         Vladymyrov bounds (per-row part; the scalar p1 root is host-side)
//   distance/base.py:384-385            indexed squared distances by direct difference
//
// One row group of G lanes (G = 32 for k <= 32, else 64) runs the WHOLE bracketing + bisection for its
// row in registers: the reference's global masked loop (one host sync per iteration, root_search.py:62)
// becomes a per-row early exit, which is equivalent because inactive rows stop moving (:61-75).
#include "tdr_common.h"

namespace tdr {

constexpr int MAX_ITEMS = 4;  // k <= 64 * 4

// group reductions of the row searches: groups of up to 16 lanes stay on the DPP path (one instruction per step, no LDS round trip);
// max is exact whatever the pairing, the sum's pairing differs from the shuffle butterfly's in rounding only
template <int G>
__device__ __forceinline__ float search_group_max(float v) {
    if (G > 16) return group_max<G>(v);
    if (G >= 2) v = fmaxf(v, dpp_mov_f<0xB1>(v));
    if (G >= 4) v = fmaxf(v, dpp_mov_f<0x4E>(v));
    if (G >= 8) v = fmaxf(v, dpp_mov_f<0x141>(v));
    if (G >= 16) v = fmaxf(v, dpp_mov_f<0x140>(v));
    return v;
}
template <int G>
__device__ __forceinline__ float search_group_sum(float v) { return G > 16 ? group_sum<G>(v) : group_sum_dpp<G>(v); }

struct UmapF {
    float rho, target;
    template <int G, int ITEMS>
    __device__ __forceinline__ float eval(const float (&c)[ITEMS], const bool (&valid)[ITEMS], float eps) const {
        float lp[ITEMS];
        float m = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
            lp[t] = valid[t] ? (-(c[t] - rho)) / eps : -__builtin_inff();
            m = fmaxf(m, lp[t]);
        }
        m = search_group_max<G>(m);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) s += valid[t] ? expf(lp[t] - m) : 0.f;
        s = search_group_sum<G>(s);
        const float lse = m + logf(s);
        return expf(lse) - target;
    }
};

struct EntropicF {
    float target;
    template <int G, int ITEMS>
    __device__ __forceinline__ float eval(const float (&c)[ITEMS], const bool (&valid)[ITEMS], float eps) const {
        float lp[ITEMS];
        float m = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
            lp[t] = valid[t] ? (-c[t]) / eps : -__builtin_inff();
            m = fmaxf(m, lp[t]);
        }
        m = group_max<G>(m);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) s += valid[t] ? expf(lp[t] - m) : 0.f;
        s = group_sum<G>(s);
        const float lse = m + logf(s);
        float h = 0.f;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
            const float l = lp[t] - lse;
            h += valid[t] ? expf(l) * (l - 1.0f) : 0.f;
        }
        h = group_sum<G>(h);
        return (-h) - target;
    }
};

// root_search.py:17-77 + :147-198 for one row; every lane of the group carries the same scalars.
template <int G, int ITEMS, typename F>
__device__ __forceinline__ float row_binary_search(const F& f, const float (&c)[ITEMS], const bool (&valid)[ITEMS],
                                                   float b, float e, int max_iter, float tol) {
    for (int it = 0; it < max_iter; ++it) {
        if (!(f.template eval<G, ITEMS>(c, valid, b) > 0.f)) break;
        e = fminf(e, b);
        b = b * 0.5f;
    }
    for (int it = 0; it < max_iter; ++it) {
        if (!(f.template eval<G, ITEMS>(c, valid, e) < 0.f)) break;
        b = fmaxf(b, e);
        e = e * 2.0f;
    }
    float f_b = f.template eval<G, ITEMS>(c, valid, b);
    float m = (b + e) * 0.5f;
    float f_m = f.template eval<G, ITEMS>(c, valid, m);
    for (int it = 0; it < max_iter; ++it) {
        if (!(fabsf(f_m) >= tol)) break;
        if (f_m * f_b > 0.f) { b = m; f_b = f_m; }
        else e = m;
        m = (b + e) * 0.5f;
        f_m = f.template eval<G, ITEMS>(c, valid, m);
    }
    return m;
}

template <int G, int ITEMS>
__global__ __launch_bounds__(256) void umap_search_kernel(const float* __restrict__ C, int64_t n, int k, float target,
                                                          int max_iter, float tol, float* __restrict__ rho_out,
                                                          float* __restrict__ eps_out, float* __restrict__ P_out) {
    const int gl = threadIdx.x % G;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (row >= n) return;
    float c[ITEMS];
    bool valid[ITEMS];
    float mn = __builtin_inff();
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int j = gl + t * G;
        valid[t] = j < k;
        c[t] = valid[t] ? C[(size_t)row * k + j] : 0.f;
        if (valid[t]) mn = fminf(mn, c[t]);
    }
    UmapF f;
    f.rho = group_min<G>(mn);
    f.target = target;
    const float eps = row_binary_search<G, ITEMS>(f, c, valid, 1.0f, 1.0f, max_iter, tol);
#pragma unroll
    for (int t = 0; t < ITEMS; ++t)
        if (valid[t]) P_out[(size_t)row * k + gl + t * G] = expf((-(c[t] - f.rho)) / eps);
    if (gl == 0) { rho_out[row] = f.rho; eps_out[row] = eps; }
}

struct EntropicScalars {
    float target;      // log(perp) + 1
    int use_bounds;    // entropic.py:280-287
    float tN_logratio; // tN * log(tN / perp)
    float tN_m1;       // tN - 1
    float log_ratio;   // log(tN / perp)
    float beta_u_num;  // log((tN - 1) * p1 / (1 - p1))
    float log_n;       // log(n_total)
};

template <int G, int ITEMS>
__global__ __launch_bounds__(256) void entropic_search_kernel(const float* __restrict__ C, int64_t n, int k,
                                                              EntropicScalars S, int max_iter, float tol,
                                                              float* __restrict__ eps_out,
                                                              float* __restrict__ lognorm_out,
                                                              float* __restrict__ logP_out) {
    const int gl = threadIdx.x % G;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (row >= n) return;
    float c[ITEMS];
    bool valid[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int j = gl + t * G;
        valid[t] = j < k;
        c[t] = valid[t] ? C[(size_t)row * k + j] : 0.f;
    }
    float b = 1.0f, e = 1.0f;
    if (S.use_bounds) {
        // d1 <= d2 = two smallest, dN = largest of the row (entropic.py:96-100)
        float mx = -__builtin_inff(), m1 = __builtin_inff();
#pragma unroll
        for (int t = 0; t < ITEMS; ++t)
            if (valid[t]) { mx = fmaxf(mx, c[t]); m1 = fminf(m1, c[t]); }
        const float dN = group_max<G>(mx);
        const float d1 = group_min<G>(m1);
        // second smallest: smallest among all but ONE occurrence of d1
        unsigned long long eqmask_any = 0;
        float m2 = __builtin_inff();
        int first_eq = 1 << 30;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t)
            if (valid[t] && c[t] == d1) first_eq = min(first_eq, gl + t * G);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) first_eq = min(first_eq, __shfl_xor(first_eq, o, 64));
#pragma unroll
        for (int t = 0; t < ITEMS; ++t)
            if (valid[t] && (gl + t * G) != first_eq) m2 = fminf(m2, c[t]);
        const float d2 = group_min<G>(m2);
        (void)eqmask_any;
        const float Delta_N = dN - d1;
        const float Delta_2 = d2 - d1;
        const float bl1 = S.tN_logratio / (S.tN_m1 * Delta_N);
        const float bl2 = sqrtf(S.log_ratio / (dN * dN - d1 * d1));
        const float beta_L = fmaxf(bl1, bl2);
        const float beta_U = S.beta_u_num / Delta_2;
        b = 1.0f / beta_U + 1e-6f;
        e = 1.0f / beta_L;
    }
    EntropicF f;
    f.target = S.target;
    const float eps = row_binary_search<G, ITEMS>(f, c, valid, b, e, max_iter, tol);
    // log P = -C/eps - LSE - log N  (entropic.py:299-310)
    float lp[ITEMS];
    float m = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        lp[t] = valid[t] ? (-c[t]) / eps : -__builtin_inff();
        m = fmaxf(m, lp[t]);
    }
    m = group_max<G>(m);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) s += valid[t] ? expf(lp[t] - m) : 0.f;
    s = group_sum<G>(s);
    const float lse = m + logf(s);
#pragma unroll
    for (int t = 0; t < ITEMS; ++t)
        if (valid[t]) logP_out[(size_t)row * k + gl + t * G] = (lp[t] - lse) - S.log_n;
    if (gl == 0) { eps_out[row] = eps; lognorm_out[row] = lse; }
}

// ---- streaming variants (any row length; dense N x N affinities, sparsity=False) ----------------------
// One wavefront per row; the row is re-read from global memory (L2-resident) at every evaluation.
struct StreamRow {
    const float* c;
    int k, lane;
};
__device__ __forceinline__ float stream_umap_f(const StreamRow& R, float rho, float target, float eps) {
    float m = -__builtin_inff();
    for (int j = R.lane; j < R.k; j += 64) m = fmaxf(m, (-(R.c[j] - rho)) / eps);
    m = wave_max(m);
    float s = 0.f;
    for (int j = R.lane; j < R.k; j += 64) s += expf((-(R.c[j] - rho)) / eps - m);
    s = wave_sum(s);
    return expf(m + logf(s)) - target;
}
__device__ __forceinline__ float stream_entropic_f(const StreamRow& R, float target, float eps, float* lse_out) {
    float m = -__builtin_inff();
    for (int j = R.lane; j < R.k; j += 64) m = fmaxf(m, (-R.c[j]) / eps);
    m = wave_max(m);
    float s = 0.f;
    for (int j = R.lane; j < R.k; j += 64) s += expf((-R.c[j]) / eps - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    float h = 0.f;
    for (int j = R.lane; j < R.k; j += 64) {
        const float l = (-R.c[j]) / eps - lse;
        h += expf(l) * (l - 1.0f);
    }
    h = wave_sum(h);
    if (lse_out) *lse_out = lse;
    return (-h) - target;
}

template <typename F>
__device__ __forceinline__ float stream_binary_search(F f, float b, float e, int max_iter, float tol) {
    for (int it = 0; it < max_iter; ++it) {
        if (!(f(b) > 0.f)) break;
        e = fminf(e, b);
        b = b * 0.5f;
    }
    for (int it = 0; it < max_iter; ++it) {
        if (!(f(e) < 0.f)) break;
        b = fmaxf(b, e);
        e = e * 2.0f;
    }
    float f_b = f(b);
    float m = (b + e) * 0.5f;
    float f_m = f(m);
    for (int it = 0; it < max_iter; ++it) {
        if (!(fabsf(f_m) >= tol)) break;
        if (f_m * f_b > 0.f) { b = m; f_b = f_m; }
        else e = m;
        m = (b + e) * 0.5f;
        f_m = f(m);
    }
    return m;
}

__global__ __launch_bounds__(256) void umap_search_stream_kernel(const float* __restrict__ C, int64_t n, int k, float target,
                                                                 int max_iter, float tol, float* __restrict__ rho_out,
                                                                 float* __restrict__ eps_out, float* __restrict__ P_out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    StreamRow R;
    R.c = C + (size_t)row * k; R.k = k; R.lane = threadIdx.x & 63;
    float mn = __builtin_inff();
    for (int j = R.lane; j < k; j += 64) mn = fminf(mn, R.c[j]);
    const float rho = -wave_max(-mn);
    const float eps = stream_binary_search([&](float x) { return stream_umap_f(R, rho, target, x); }, 1.0f, 1.0f, max_iter, tol);
    for (int j = R.lane; j < k; j += 64) P_out[(size_t)row * k + j] = expf((-(R.c[j] - rho)) / eps);
    if (R.lane == 0) { rho_out[row] = rho; eps_out[row] = eps; }
}

__global__ __launch_bounds__(256) void entropic_search_stream_kernel(const float* __restrict__ C, int64_t n, int k,
                                                                     EntropicScalars S, int max_iter, float tol,
                                                                     float* __restrict__ eps_out, float* __restrict__ lognorm_out,
                                                                     float* __restrict__ logP_out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    StreamRow R;
    R.c = C + (size_t)row * k; R.k = k; R.lane = threadIdx.x & 63;
    float b = 1.0f, e = 1.0f;
    if (S.use_bounds) {
        float mx = -__builtin_inff(), m1 = __builtin_inff();
        for (int j = R.lane; j < k; j += 64) { mx = fmaxf(mx, R.c[j]); m1 = fminf(m1, R.c[j]); }
        const float dN = wave_max(mx);
        const float d1 = -wave_max(-m1);
        int first_eq = 1 << 30;
        for (int j = R.lane; j < k; j += 64) if (R.c[j] == d1) first_eq = min(first_eq, j);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) first_eq = min(first_eq, __shfl_xor(first_eq, o, 64));
        float m2 = __builtin_inff();
        for (int j = R.lane; j < k; j += 64) if (j != first_eq) m2 = fminf(m2, R.c[j]);
        const float d2 = -wave_max(-m2);
        const float beta_L = fmaxf(S.tN_logratio / (S.tN_m1 * (dN - d1)), sqrtf(S.log_ratio / (dN * dN - d1 * d1)));
        const float beta_U = S.beta_u_num / (d2 - d1);
        b = 1.0f / beta_U + 1e-6f;
        e = 1.0f / beta_L;
    }
    const float eps = stream_binary_search([&](float x) { return stream_entropic_f(R, S.target, x, nullptr); }, b, e, max_iter, tol);
    float lse;
    stream_entropic_f(R, S.target, eps, &lse);
    for (int j = R.lane; j < k; j += 64) logP_out[(size_t)row * k + j] = ((-R.c[j]) / eps - lse) - S.log_n;
    if (R.lane == 0) { eps_out[row] = eps; lognorm_out[row] = lse; }
}

// distance/base.py:384-385 -- out[i][c] = sum_d (X[q_i][d] - Y[key[i][c]][d])^2 ; negative keys wrap.
__global__ __launch_bounds__(256) void indexed_sqdist_kernel(const float* __restrict__ X, int64_t nx, int d,
                                                             const float* __restrict__ Y, int64_t ny,
                                                             const int64_t* __restrict__ q, int64_t nq, int nk,
                                                             int take_sqrt, const int64_t* __restrict__ keys,
                                                             float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nq * nk) return;
    const int64_t i = idx / nk;
    int64_t qi = q[i];
    if (qi < 0) qi += nx;
    int64_t kj = keys[idx];
    if (kj < 0) kj += ny;
    const float* x = X + (size_t)qi * d;
    const float* y = Y + (size_t)kj * d;
    float acc = 0.f;
    if (take_sqrt == 2) {  // manhattan (distance/base.py:388)
        for (int t = 0; t < d; ++t) acc = __fadd_rn(acc, fabsf(x[t] - y[t]));
        out[idx] = acc;
        return;
    }
    if (take_sqrt == 4) {  // sqhyperbolic by direct difference (distance/base.py:392-398)
        float xn = 0.f, yn = 0.f;
        for (int t = 0; t < d; ++t) {
            const float df = x[t] - y[t];
            acc = __fadd_rn(acc, __fmul_rn(df, df));
            xn = __fadd_rn(xn, __fmul_rn(x[t], x[t]));
            yn = __fadd_rn(yn, __fmul_rn(y[t], y[t]));
        }
        const float den = __fmul_rn(__fsub_rn(1.0f, xn), __fsub_rn(1.0f, yn));
        const float w = __fadd_rn(__fadd_rn(1.0f, __fmul_rn(2.0f, __fdiv_rn(fmaxf(acc, 0.f), den))), 1e-8f);
        const float u = acoshf(w);
        out[idx] = __fmul_rn(u, u);
        return;
    }
    if (take_sqrt == 3) {  // angular: -<x, y> (distance/base.py:390-391)
        for (int t = 0; t < d; ++t) acc = __fadd_rn(acc, __fmul_rn(x[t], y[t]));
        out[idx] = -acc;
        return;
    }
    for (int t = 0; t < d; ++t) {
        const float df = x[t] - y[t];
        acc = __fadd_rn(acc, __fmul_rn(df, df));
    }
    out[idx] = take_sqrt ? sqrtf(acc) : acc;
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

template <typename K, typename... A>
static int launch_rows(K kern, int G, int64_t n, hipStream_t st, A... args) {
    const int rows_per_block = 256 / G;
    const unsigned grid = (unsigned)((n + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, args...);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* UMAP sigma search: C (n,k) -> rho (n), eps (n), P (n,k).  target = log2(n_neighbors). */
int tdr_umap_search_f32(const float* C, int64_t n, int k, float target, int max_iter, float tol, float* rho,
                        float* eps, float* P, void* stream) {
    if (!C || !rho || !eps || !P || n <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (k > 64 * MAX_ITEMS) {
        hipLaunchKernelGGL(umap_search_stream_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, C, n, k, target, max_iter,
                           tol, rho, eps, P);
        TDR_CHECK_LAUNCH();
        return TDR_OK;
    }
    // 8 lanes x 4 entries for the usual widths (eight rows per wavefront, three DPP steps per reduction): 1.46 -> 0.94 (16 x 2) -> see
    // DESIGN section 6 at N = 1M, k = 30
    if (k <= 32) return launch_rows(umap_search_kernel<8, 4>, 8, n, st, C, n, k, target, max_iter, tol, rho, eps, P);
    if (k <= 64) return launch_rows(umap_search_kernel<64, 1>, 64, n, st, C, n, k, target, max_iter, tol, rho, eps, P);
    if (k <= 128) return launch_rows(umap_search_kernel<64, 2>, 64, n, st, C, n, k, target, max_iter, tol, rho, eps, P);
    return launch_rows(umap_search_kernel<64, 4>, 64, n, st, C, n, k, target, max_iter, tol, rho, eps, P);
}

/*
 * Entropic (perplexity) search: C (n,k) -> eps (n), log_norm (n), log_P (n,k).
 *   target = log(perp) + 1; log_n = log(n_total);
 *   use_bounds != 0: per-row Vladymyrov bounds from the host-computed scalars
 *     tN_logratio = tN*log(tN/perp), tN_m1 = tN-1, log_ratio = log(tN/perp),
 *     beta_u_num = log((tN-1)*p1/(1-p1))   (entropic.py:96-113, tN = number of rows of C).
 */
int tdr_entropic_search_f32(const float* C, int64_t n, int k, float target, float log_n, int max_iter, float tol,
                            int use_bounds, float tN_logratio, float tN_m1, float log_ratio, float beta_u_num,
                            float* eps, float* log_norm, float* log_P, void* stream) {
    if (!C || !eps || !log_norm || !log_P || n <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    if (use_bounds && k < 2) return TDR_ERR_BAD_ARG;
    EntropicScalars S;
    S.target = target; S.use_bounds = use_bounds; S.tN_logratio = tN_logratio; S.tN_m1 = tN_m1;
    S.log_ratio = log_ratio; S.beta_u_num = beta_u_num; S.log_n = log_n;
    hipStream_t st = (hipStream_t)stream;
    if (k > 64 * MAX_ITEMS) {
        hipLaunchKernelGGL(entropic_search_stream_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, C, n, k, S, max_iter,
                           tol, eps, log_norm, log_P);
        TDR_CHECK_LAUNCH();
        return TDR_OK;
    }
    if (k <= 32) return launch_rows(entropic_search_kernel<32, 1>, 32, n, st, C, n, k, S, max_iter, tol, eps, log_norm, log_P);
    if (k <= 64) return launch_rows(entropic_search_kernel<64, 1>, 64, n, st, C, n, k, S, max_iter, tol, eps, log_norm, log_P);
    if (k <= 128) return launch_rows(entropic_search_kernel<64, 2>, 64, n, st, C, n, k, S, max_iter, tol, eps, log_norm, log_P);
    return launch_rows(entropic_search_kernel<64, 4>, 64, n, st, C, n, k, S, max_iter, tol, eps, log_norm, log_P);
}

/* Gathered distances: out (nq, nk); take_sqrt 0 = squared Euclidean, 1 = Euclidean, 2 = manhattan, 3 = angular, 4 = sqhyperbolic. q/keys are int64;
 * negative keys wrap. */
int tdr_indexed_sqdist_f32(const float* X, int64_t nx, int d, const float* Y, int64_t ny, const int64_t* q,
                           int64_t nq, int nk, int take_sqrt, const int64_t* keys, float* out, void* stream) {
    if (!X || !Y || !q || !keys || !out || nq < 0 || nk < 0 || d <= 0) return TDR_ERR_BAD_ARG;
    if (nq == 0 || nk == 0) return TDR_OK;
    const int64_t total = nq * nk;
    hipLaunchKernelGGL(indexed_sqdist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       X, nx, d, Y, ny, q, nq, nk, take_sqrt, keys, out);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

int tdr_fill_f32(float* p, int64_t n, float v, void* stream) {
    if (!p || n < 0) return TDR_ERR_BAD_ARG;
    if (n == 0) return TDR_OK;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n, v);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
