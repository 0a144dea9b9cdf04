// float64 twins of the embedding-loop kernels (tdr_embed.hip): the reference computes in the dtype of its input and its
// own tests run every neighbour-embedding method in float32 AND float64 (tests/test_neighbor_embedding.py:34,55-74), so a
// float64 block is embedded in float64 end to end -- kNN / affinity (tdr_f64.hip), UMAP's epoch counters, the embedding,
// the forces and the optimizer state.
//
//   neighbor_embedding/umap.py:215-234   epochs_per_sample / epoch_of_next_sample   -> tdr_umap_prepare_f64
//   neighbor_embedding/umap.py:236-292   closed-form gradients, per-step form       -> tdr_umap_grad_f64
//   neighbor_embedding/largevis.py:181-201, tsne.py:162-170, sne.py, infotsne.py    -> tdr_ne_grad_f64 (kinds 0-3)
//   neighbor_embedding/tsne.py:172-180   dense repulsion                            -> tdr_tsne_repulsion_f64 / tdr_add_scaled_f64
//   affinity_matcher.py:427-429          torch.optim.SGD(momentum) step             -> tdr_sgd_step_f64
//
// These are the plain per-step forms (one row group of 16 lanes walks a row's edges and negatives; no firing lists, no L2
// slicing): float64 runs are parity runs, the arithmetic is 1/2-rate at best and d^b goes through the double-precision
// pow of the device library.  Same counter-hash negative sampler as the float32 kernels (tdr_embed_common.h), so a
// float64 fit draws the negatives a float32 fit of the same data and seed draws.
#include "tdr_embed_common.h"

namespace tdr {

template <int NC>
struct VecD {
    double v[NC];
};

template <int NC, bool PAD>
__device__ __forceinline__ VecD<NC> load_zd(const double* __restrict__ Z, int64_t i, int nc) {
    VecD<NC> r;
    if (!PAD && NC == 2) {
        const double2 t = *reinterpret_cast<const double2*>(Z + (size_t)i * 2);
        r.v[0] = t.x; r.v[1] = t.y;
        return r;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) r.v[c] = (!PAD || c < nc) ? Z[(size_t)i * (PAD ? nc : NC) + c] : 0.0;
    return r;
}

template <int G>
__device__ __forceinline__ double group_sum_d(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- umap.py:215-234 in float64 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_f64_kernel(const double* __restrict__ v, int64_t n, unsigned long long* __restrict__ amax_bits) {
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmax(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, (unsigned long long)__double_as_longlong(m));  // values are >= 0
}

__global__ __launch_bounds__(256) void umap_prepare_f64_kernel(const double* __restrict__ v, int64_t n,
                                                               const unsigned long long* __restrict__ amax_bits, double max_iter,
                                                               double* __restrict__ eps_per, double* __restrict__ next) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double amax = __longlong_as_double((long long)*amax_bits);
    const double thr = amax / max_iter;
    const double a = v[i];
    double e = __dmul_rn(1.0 / __dadd_rn(a, 1e-3), amax);   // A_max * (1 / (A + 1e-3)): the reference's op order
    if (a <= thr) e = __builtin_inf();
    eps_per[i] = e;
    next[i] = e;
}

struct UmapStepParamsD {
    const double* Z;
    int64_t n_total, row0, n_rows;
    const int64_t* rowptr;
    const int32_t* cols;
    const double* eps_per;
    double* next;
    double a, b, t1;
    int neg_rate, n_negatives;
    const int64_t* neg_inj;
    uint64_t seed;
    uint32_t iter;
    double exag, rep, eps;
    double* grad;
    int nc;
};

template <int NC, int G, bool PAD>
__global__ __launch_bounds__(256) void umap_grad_f64_kernel(const UmapStepParamsD P) {
    const int nc = PAD ? P.nc : NC;
    const int gl = threadIdx.x % G;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= P.n_rows) return;
    const int64_t gi = P.row0 + r;
    const int64_t e0 = P.rowptr[r], e1 = P.rowptr[r + 1];
    const VecD<NC> zi = load_zd<NC, PAD>(P.Z, gi, nc);
    const double two_ab = 2.0 * P.a * P.b;
    double ga[NC], gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ga[c] = 0.0; gr[c] = 0.0; }
    int act = 0;
    for (int64_t e = e0 + gl; e < e1; e += G) {
        const double nx = P.next[e];
        if (!(nx <= P.t1)) continue;
        P.next[e] = nx + P.eps_per[e];
        act++;
        const VecD<NC> zj = load_zd<NC, PAD>(P.Z, P.cols[e], nc);
        double df[NC], d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        if (d > 0.0) {
            const double pb = pow(d, P.b);
            const double coef = (two_ab * pb / d) / (1.0 + P.a * pb);   // 2ab d^(b-1) / (1 + a d^b)
#pragma unroll
            for (int c = 0; c < NC; ++c) ga[c] += coef * df[c];
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) ga[c] = group_sum_d<G>(ga[c]);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) act += __shfl_xor(act, o, 64);
    int n_use = act * P.neg_rate;
    if (n_use > P.n_negatives) n_use = P.n_negatives;
    const uint32_t rkey = neg_row_key(P.seed, P.iter, gi);
    const double m2b = -2.0 * P.b;
    for (int col = gl; col < n_use; col += G) {
        const int64_t j = P.neg_inj ? P.neg_inj[(size_t)r * P.n_negatives + col] : sample_negative(rkey, gi, col, P.n_total);
        const VecD<NC> zj = load_zd<NC, PAD>(P.Z, j, nc);
        double df[NC], d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const double den = 1.0 + P.a * (d > 0.0 ? pow(d, P.b) : 0.0);
        const double coef = m2b / ((d + P.eps) * den);
#pragma unroll
        for (int c = 0; c < NC; ++c) gr[c] += coef * df[c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = group_sum_d<G>(gr[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c >= nc) break;
            const double a_ = fmin(fmax(ga[c], -4.0), 4.0);
            const double r_ = fmin(fmax(gr[c], -4.0), 4.0);
            P.grad[(size_t)r * nc + c] = P.exag * a_ + P.rep * r_;
        }
    }
}

// ---- LargeVis / TSNE / SNE / InfoTSNE sparse terms (the float32 kernel's structure, tdr_embed.hip:ne_grad_kernel) -------
struct NeStepParamsD {
    const double* Z;
    int64_t n_total, row0, n_rows;
    const int32_t* nn;
    const double* P;
    int k, kind;
    double exag, rep_coef;
    int n_neg;
    const int64_t* neg_inj;
    uint64_t seed;
    uint32_t iter;
    double* grad;
    const int64_t* t_rowptr;
    const int32_t* t_src;
    const double* t_val;
    int nc;
};

template <int NC, int G, bool PAD>
__global__ __launch_bounds__(256) void ne_grad_f64_kernel(const NeStepParamsD S) {
    const int nc = PAD ? S.nc : NC;
    const int gl = threadIdx.x % G;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (r >= S.n_rows) return;
    const int64_t gi = S.row0 + r;
    const VecD<NC> zi = load_zd<NC, PAD>(S.Z, gi, nc);
    double g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.0;
    const double off = (S.kind == 0) ? 2.0 : 1.0;
    const bool gauss = S.kind == 2;
    const bool pull = S.t_rowptr != nullptr;
    for (int p = gl; p < S.k; p += G) {
        const int64_t j = S.nn[(size_t)r * S.k + p];
        const double pij = S.P[(size_t)r * S.k + p];
        const VecD<NC> zj = load_zd<NC, PAD>(S.Z, j, nc);
        double df[NC], d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const double w = S.exag * 2.0 * pij * (gauss ? 1.0 : 1.0 / (off + d));
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double t = w * df[c];
            g[c] += t;
            if (!pull && c < nc) unsafeAtomicAdd(&S.grad[(size_t)j * nc + c], -t);
        }
    }
    if (pull) {
        const int64_t e1 = S.t_rowptr[r + 1];
        for (int64_t e = S.t_rowptr[r] + gl; e < e1; e += G) {
            const VecD<NC> zs = load_zd<NC, PAD>(S.Z, S.t_src[e], nc);
            double df[NC], d = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zs.v[c]; d += df[c] * df[c]; }
            const double w = S.exag * 2.0 * S.t_val[e] * (gauss ? 1.0 : 1.0 / (off + d));
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] += w * df[c];
        }
    }
    const uint32_t rkey = neg_row_key(S.seed, S.iter, gi);
    double inv_rowsum = 0.0;
    if (S.kind == 3) {
        double s = 0.0;
        for (int col = gl; col < S.n_neg; col += G) {
            const int64_t j = S.neg_inj ? S.neg_inj[(size_t)r * S.n_neg + col] : sample_negative(rkey, gi, col, S.n_total);
            const VecD<NC> zj = load_zd<NC, PAD>(S.Z, j, nc);
            double d = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const double t = zi.v[c] - zj.v[c]; d += t * t; }
            s += 1.0 / (1.0 + d);
        }
        inv_rowsum = 1.0 / group_sum_d<G>(s);
    }
    for (int col = gl; col < S.n_neg; col += G) {
        const int64_t j = S.neg_inj ? S.neg_inj[(size_t)r * S.n_neg + col] : sample_negative(rkey, gi, col, S.n_total);
        const VecD<NC> zj = load_zd<NC, PAD>(S.Z, j, nc);
        double df[NC], d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        double w;
        if (S.kind == 3) {
            const double q = 1.0 / (1.0 + d);
            w = -S.rep_coef * q * q * inv_rowsum;
        } else {
            w = -S.rep_coef / ((1.0 + d) * (2.0 + d));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double t = w * df[c];
            g[c] += t;
            if (c < nc) unsafeAtomicAdd(&S.grad[(size_t)j * nc + c], -t);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum_d<G>(g[c]);
        if (gl == 0 && c < nc) unsafeAtomicAdd(&S.grad[(size_t)gi * nc + c], g[c]);
    }
}

// ---- TSNE dense repulsion (tsne.py:172-180) -----------------------------------------------------------------------------
template <int NC, bool PAD>
__global__ __launch_bounds__(256) void tsne_repulsion_f64_kernel(const double* __restrict__ Z, int64_t n_total, int64_t row0,
                                                                 int64_t n_rows, double* __restrict__ F, double* __restrict__ S, int nc_) {
    const int nc = PAD ? nc_ : NC;
    __shared__ double tile[256 * NC];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    VecD<NC> zi;
#pragma unroll
    for (int c = 0; c < NC; ++c) zi.v[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.0;
    double f[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) f[c] = 0.0;
    double s = 0.0;
    for (int64_t j0 = 0; j0 < n_total; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < NC; ++c) tile[threadIdx.x * NC + c] = (j < n_total && c < nc) ? Z[(size_t)j * nc + c] : 0.0;
        __syncthreads();
        const int lim = (int)((n_total - j0 < 256) ? (n_total - j0) : 256);
        for (int t = 0; t < lim; ++t) {
            double df[NC], d = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - tile[t * NC + c]; d += df[c] * df[c]; }
            const double w = 1.0 / (1.0 + d);
            s += w;
            const double w2 = w * w;
#pragma unroll
            for (int c = 0; c < NC; ++c) f[c] += w2 * df[c];
        }
    }
    if (have) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < nc) F[(size_t)r * nc + c] = f[c];
    } else s = 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(S, s);
}

// ---- SNE dense repulsion (sne.py:172-179) in float64: R_i = sum_j e^{-d_ij} (diagonal included), then
// g_i += coef * sum_j e^{-d_ij} (1/R_i + 1/R_j) (z_i - z_j).  Plain LDS-tiled all-pairs loops, one thread per row.
template <int NC, bool PAD>
__global__ __launch_bounds__(256) void sne_rowsum_f64_kernel(const double* __restrict__ Z, int64_t n_total, int64_t row0, int64_t n_rows,
                                                             double* __restrict__ R, int nc_) {
    const int nc = PAD ? nc_ : NC;
    __shared__ double tile[256 * NC];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    double zi[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) zi[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.0;
    double s = 0.0;
    for (int64_t j0 = 0; j0 < n_total; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < NC; ++c) tile[threadIdx.x * NC + c] = (j < n_total && c < nc) ? Z[(size_t)j * nc + c] : 0.0;
        __syncthreads();
        const int lim = (int)((n_total - j0 < 256) ? (n_total - j0) : 256);
        for (int t = 0; t < lim; ++t) {
            double d = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const double u = zi[c] - tile[t * NC + c]; d += u * u; }
            s += exp(-d);
        }
    }
    if (have) R[r] = s;
}

template <int NC, bool PAD>
__global__ __launch_bounds__(256) void sne_repulsion_f64_kernel(const double* __restrict__ Z, int64_t n_total, int64_t row0, int64_t n_rows,
                                                                const double* __restrict__ R, double coef, double* __restrict__ grad, int nc_) {
    const int nc = PAD ? nc_ : NC;
    __shared__ double tile[256 * (NC + 1)];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = r < n_rows;
    double zi[NC], f[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { zi[c] = (have && c < nc) ? Z[(size_t)(row0 + r) * nc + c] : 0.0; f[c] = 0.0; }
    const double inv_ri = have ? 1.0 / R[row0 + r] : 0.0;
    for (int64_t j0 = 0; j0 < n_total; j0 += 256) {
        __syncthreads();
        const int64_t j = j0 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < NC; ++c) tile[threadIdx.x * (NC + 1) + c] = (j < n_total && c < nc) ? Z[(size_t)j * nc + c] : 0.0;
        tile[threadIdx.x * (NC + 1) + NC] = (j < n_total) ? 1.0 / R[j] : 0.0;
        __syncthreads();
        const int lim = (int)((n_total - j0 < 256) ? (n_total - j0) : 256);
        for (int t = 0; t < lim; ++t) {
            double df[NC], d = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) { df[c] = zi[c] - tile[t * (NC + 1) + c]; d += df[c] * df[c]; }
            const double w = exp(-d) * (inv_ri + tile[t * (NC + 1) + NC]);
#pragma unroll
            for (int c = 0; c < NC; ++c) f[c] += w * df[c];
        }
    }
    if (have) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < nc) grad[(size_t)(row0 + r) * nc + c] += coef * f[c];
    }
}

// ---- PaCMAP pair losses (pacmap.py:213-265) in float64: the closed-form gradient of tdr_pacmap_grad_f32 ------------------
struct PacmapParamsD {
    const double* Z;
    int64_t n;
    const int64_t* near; int m_near; double w_nb;
    const int64_t* mid;  int m_mid;  double w_mn;
    const int64_t* far_; int m_far;  double w_fp;
    double* grad;
    int nc;
};

template <int NC, int G, bool PAD>
__global__ __launch_bounds__(256) void pacmap_grad_f64_kernel(const PacmapParamsD P) {
    const int nc = PAD ? P.nc : NC;
    const int gl = threadIdx.x % G;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (i >= P.n) return;
    const VecD<NC> zi = load_zd<NC, PAD>(P.Z, i, nc);
    double g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.0;
    const int total = P.m_near + P.m_mid + P.m_far;
    for (int p = gl; p < total; p += G) {
        int64_t j;
        double num, off, w;
        if (p < P.m_near) { j = P.near[(size_t)i * P.m_near + p]; num = 10.0; off = 11.0; w = P.w_nb; }
        else if (p < P.m_near + P.m_mid) { j = P.mid[(size_t)i * P.m_mid + (p - P.m_near)]; num = 1.0e4; off = 10001.0; w = P.w_mn; }
        else { j = P.far_[(size_t)i * P.m_far + (p - P.m_near - P.m_mid)]; num = -1.0; off = 2.0; w = P.w_fp; }
        if (w == 0.0) continue;
        const VecD<NC> zj = load_zd<NC, PAD>(P.Z, j, nc);
        double df[NC], d = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { df[c] = zi.v[c] - zj.v[c]; d += df[c] * df[c]; }
        const double den = off + d;
        const double coef = 2.0 * w * num / (den * den);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double t = coef * df[c];
            g[c] += t;
            if (c < nc) unsafeAtomicAdd(&P.grad[(size_t)j * nc + c], -t);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = group_sum_d<G>(g[c]);
        if (gl == 0 && c < nc) unsafeAtomicAdd(&P.grad[(size_t)i * nc + c], g[c]);
    }
}

__global__ __launch_bounds__(256) void add_scaled_f64_kernel(double* __restrict__ grad, const double* __restrict__ F,
                                                             const double* __restrict__ S, double coef, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    grad[i] += (coef / *S) * F[i];
}

__global__ __launch_bounds__(256) void sgd_step_f64_kernel(double* __restrict__ Z, const double* __restrict__ grad,
                                                           double* __restrict__ buf, int64_t n, double lr, double momentum,
                                                           int first, int* __restrict__ nan_flag, int iter) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double g = grad[i];
    if (momentum != 0.0) {
        const double bprev = first ? 0.0 : buf[i];
        g = first ? g : __dadd_rn(__dmul_rn(bprev, momentum), g);
        buf[i] = g;
    }
    const double z = __dadd_rn(Z[i], -__dmul_rn(lr, g));   // p.add_(grad, alpha=-lr): one rounding for lr*g, one for the sum
    Z[i] = z;
    if (z != z) atomicCAS(nan_flag, 0, iter + 1);
}

template <int G, typename Prm>
static int launch_group_d(void (*kern)(const Prm), const Prm& P, int64_t n_rows, hipStream_t st) {
    const int rpb = 256 / G;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_rows + rpb - 1) / rpb)), dim3(256), 0, st, P);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDR_OK : (int)e;
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* umap.py:215-234 on float64 CSR values.  scratch: >= 8 bytes of device memory. */
int tdr_umap_prepare_f64(const double* vals, int64_t nnz, int max_iter, double* eps_per, double* next, void* scratch, void* stream) {
    if (!vals || !eps_per || !next || !scratch || nnz <= 0 || max_iter <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, 8, st);
    if (e != hipSuccess) return (int)e;
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(amax_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, vals, nnz, (unsigned long long*)scratch);
    hipLaunchKernelGGL(umap_prepare_f64_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, vals, nnz,
                       (const unsigned long long*)scratch, (double)max_iter, eps_per, next);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* One evaluation of UMAP's closed-form gradient in float64 for rows [row0, row0 + n_rows): grad (n_rows, nc); `next` is
 * advanced in place (umap.py:243-247).  nc in 1..32. */
int tdr_umap_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int64_t* rowptr,
                      const int32_t* cols, const double* eps_per, double* next, double a, double b, int n_iter, int neg_rate,
                      int n_negatives, const int64_t* neg_inj, uint64_t seed, double exag, double rep, double eps, double* grad,
                      void* stream) {
    if (!Z || !rowptr || !cols || !eps_per || !next || !grad || n_rows <= 0 || n_total < 2) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32) return TDR_ERR_UNSUPPORTED;
    UmapStepParamsD P;
    P.Z = Z; P.n_total = n_total; P.row0 = row0; P.n_rows = n_rows; P.rowptr = rowptr; P.cols = cols; P.eps_per = eps_per;
    P.next = next; P.a = a; P.b = b; P.t1 = (double)(n_iter + 1); P.neg_rate = neg_rate; P.n_negatives = n_negatives;
    P.neg_inj = neg_inj; P.seed = seed; P.iter = (uint32_t)n_iter; P.exag = exag; P.rep = rep; P.eps = eps; P.grad = grad; P.nc = nc;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return launch_group_d<16>(umap_grad_f64_kernel<2, 16, false>, P, n_rows, st);
    if (nc == 3) return launch_group_d<16>(umap_grad_f64_kernel<3, 16, false>, P, n_rows, st);
    if (nc <= 4) return launch_group_d<16>(umap_grad_f64_kernel<4, 16, true>, P, n_rows, st);
    if (nc <= 8) return launch_group_d<16>(umap_grad_f64_kernel<8, 16, true>, P, n_rows, st);
    if (nc <= 16) return launch_group_d<16>(umap_grad_f64_kernel<16, 16, true>, P, n_rows, st);
    return launch_group_d<16>(umap_grad_f64_kernel<32, 16, true>, P, n_rows, st);
}

/* float64 twin of tdr_ne_grad_f32 (same arguments; grad (N, nc) zeroed by the caller). */
int tdr_ne_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn, const double* P_,
                    int k, const int64_t* t_rowptr, const int32_t* t_src, const double* t_val, int kind, double exag,
                    double rep_coef, int n_neg, const int64_t* neg_inj, uint64_t seed, int n_iter, double* grad, void* stream) {
    if (!Z || !nn || !P_ || !grad || n_rows <= 0 || k <= 0 || n_total < 2) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32) return TDR_ERR_UNSUPPORTED;
    if (kind < 0 || kind > 3) return TDR_ERR_BAD_ARG;
    if (t_rowptr && (!t_src || !t_val)) return TDR_ERR_BAD_ARG;
    NeStepParamsD S;
    S.Z = Z; S.n_total = n_total; S.row0 = row0; S.n_rows = n_rows; S.nn = nn; S.P = P_; S.k = k; S.kind = kind; S.exag = exag;
    S.rep_coef = rep_coef; S.n_neg = n_neg; S.neg_inj = neg_inj; S.seed = seed; S.iter = (uint32_t)n_iter; S.grad = grad;
    S.t_rowptr = t_rowptr; S.t_src = t_src; S.t_val = t_val; S.nc = nc;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return launch_group_d<16>(ne_grad_f64_kernel<2, 16, false>, S, n_rows, st);
    if (nc == 3) return launch_group_d<16>(ne_grad_f64_kernel<3, 16, false>, S, n_rows, st);
    if (nc <= 4) return launch_group_d<16>(ne_grad_f64_kernel<4, 16, true>, S, n_rows, st);
    if (nc <= 8) return launch_group_d<16>(ne_grad_f64_kernel<8, 16, true>, S, n_rows, st);
    if (nc <= 16) return launch_group_d<16>(ne_grad_f64_kernel<16, 16, true>, S, n_rows, st);
    return launch_group_d<16>(ne_grad_f64_kernel<32, 16, true>, S, n_rows, st);
}

/* float64 twin of tdr_tsne_repulsion_f32: F (n_rows, nc), *S (device double, caller-zeroed) += sum_ij 1/(1+d_ij). */
int tdr_tsne_repulsion_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double* F, double* S, void* stream) {
    if (!Z || !F || !S || n_rows <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    if (nc == 2) hipLaunchKernelGGL((tsne_repulsion_f64_kernel<2, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc);
    else if (nc == 3) hipLaunchKernelGGL((tsne_repulsion_f64_kernel<3, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc);
    else if (nc >= 1 && nc <= 4) hipLaunchKernelGGL((tsne_repulsion_f64_kernel<4, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc);
    else if (nc >= 1 && nc <= 8) hipLaunchKernelGGL((tsne_repulsion_f64_kernel<8, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc);
    else if (nc >= 1 && nc <= 16) hipLaunchKernelGGL((tsne_repulsion_f64_kernel<16, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, F, S, nc);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* grad[i] += (coef / *S) * F[i] for i < n (flat), float64. */
int tdr_add_scaled_f64(double* grad, const double* F, const double* S, double coef, int64_t n, void* stream) {
    if (!grad || !F || !S || n <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_scaled_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, F, S, coef, n);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* torch.optim.SGD(momentum) step on n flat float64 elements (tdr_sgd_step_f32's contract). */
int tdr_sgd_step_f64(double* Z, const double* grad, double* buf, int64_t n, double lr, double momentum, int first, int* nan_flag,
                     int n_iter, void* stream) {
    if (!Z || !grad || !nan_flag || n <= 0) return TDR_ERR_BAD_ARG;
    if (momentum != 0.0 && !buf) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sgd_step_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, grad, buf, n, lr,
                       momentum, first, nan_flag, n_iter);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* float64 twins of tdr_sne_rowsum_f32 / tdr_sne_repulsion_f32 (sne.py:172-179): R (n_rows) = sum_j exp(-d_ij) over ALL n_total
 * points (the all-gathered R of every point is what the second pass reads); grad rows [row0, row0 + n_rows) +=
 * coef * sum_j exp(-d_ij) (1/R_i + 1/R_j) (z_i - z_j).  nc <= 16. */
int tdr_sne_rowsum_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double* R, void* stream) {
    if (!Z || !R || n_rows <= 0 || n_total <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    if (nc == 2) hipLaunchKernelGGL((sne_rowsum_f64_kernel<2, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc == 3) hipLaunchKernelGGL((sne_rowsum_f64_kernel<3, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc >= 1 && nc <= 8) hipLaunchKernelGGL((sne_rowsum_f64_kernel<8, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else if (nc >= 1 && nc <= 16) hipLaunchKernelGGL((sne_rowsum_f64_kernel<16, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, nc);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

int tdr_sne_repulsion_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const double* R, double coef,
                          double* grad, void* stream) {
    if (!Z || !R || !grad || n_rows <= 0 || n_total <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    if (nc == 2) hipLaunchKernelGGL((sne_repulsion_f64_kernel<2, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc == 3) hipLaunchKernelGGL((sne_repulsion_f64_kernel<3, false>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc >= 1 && nc <= 8) hipLaunchKernelGGL((sne_repulsion_f64_kernel<8, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else if (nc >= 1 && nc <= 16) hipLaunchKernelGGL((sne_repulsion_f64_kernel<16, true>), dim3(grid), dim3(256), 0, st, Z, n_total, row0, n_rows, R, coef, grad, nc);
    else return TDR_ERR_UNSUPPORTED;
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* float64 twin of tdr_pacmap_grad_f32 (pacmap.py:213-265): grad (n, nc) zero-initialised by the caller; nc <= 32. */
int tdr_pacmap_grad_f64(const double* Z, int nc, int64_t n, const int64_t* near_idx, int m_near, double w_nb, const int64_t* mid_idx,
                        int m_mid, double w_mn, const int64_t* far_idx, int m_far, double w_fp, double* grad, void* stream) {
    if (!Z || !grad || n <= 0 || m_near < 0 || m_mid < 0 || m_far < 0) return TDR_ERR_BAD_ARG;
    if ((m_near > 0 && !near_idx) || (m_mid > 0 && !mid_idx) || (m_far > 0 && !far_idx)) return TDR_ERR_BAD_ARG;
    if (nc < 1 || nc > 32) return TDR_ERR_UNSUPPORTED;
    PacmapParamsD P;
    P.Z = Z; P.n = n; P.near = near_idx; P.m_near = m_near; P.w_nb = w_nb; P.mid = mid_idx; P.m_mid = m_mid; P.w_mn = w_mn;
    P.far_ = far_idx; P.m_far = m_far; P.w_fp = w_fp; P.grad = grad; P.nc = nc;
    hipStream_t st = (hipStream_t)stream;
    if (nc == 2) return launch_group_d<16>(pacmap_grad_f64_kernel<2, 16, false>, P, n, st);
    if (nc == 3) return launch_group_d<16>(pacmap_grad_f64_kernel<3, 16, false>, P, n, st);
    if (nc <= 8) return launch_group_d<16>(pacmap_grad_f64_kernel<8, 16, true>, P, n, st);
    if (nc <= 16) return launch_group_d<16>(pacmap_grad_f64_kernel<16, 16, true>, P, n, st);
    return launch_group_d<16>(pacmap_grad_f64_kernel<32, 16, true>, P, n, st);
}

}  // extern "C"
