// K0 -- the steps either side of the hot path inside fit_transform, on the device:
//   utils/validation.py:308           torch.isfinite(X).all()                -> nonfinite_count_kernel
//   base.py:132-148                   torch.unique(X, dim=0, return_inverse) -> row hashes + open-addressing table + exact
//                                                                               row compare (the lexicographic merge sort
//                                                                               of the N x D block is not needed to FIND
//                                                                               duplicates)
//   spectral_embedding/pca.py:151-184 PCA initialisation: column means, the D x D Gram matrix of the centred block on
//                                     the fp32 matrix pipe, projection on the leading eigenvectors (the D x D
//                                     eigen-decomposition itself stays a library call on a tiny matrix)
#include "tdr_common.h"

namespace tdr {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// ---- isfinite scan: number of inf / nan entries of an (n, d) block -------------------------------------------------
__global__ __launch_bounds__(256) void nonfinite_count_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                              unsigned long long* __restrict__ count) {
    unsigned bad = 0;
    if (ldx == d && (((uintptr_t)X) & 15) == 0) {  // dense block: 16-byte reads over the flat range
        const int64_t total = n * (int64_t)d, n4 = total >> 2;
        const uint4* X4 = reinterpret_cast<const uint4*>(X);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const uint4 v = X4[i];
            bad += ((v.x & 0x7f800000u) == 0x7f800000u) + ((v.y & 0x7f800000u) == 0x7f800000u) +
                   ((v.z & 0x7f800000u) == 0x7f800000u) + ((v.w & 0x7f800000u) == 0x7f800000u);
        }
        if (blockIdx.x == 0 && threadIdx.x < (total & 3))
            bad += (__float_as_uint(X[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    } else {
        const int64_t total = n * (int64_t)d;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int64_t r = i / d;
            bad += (__float_as_uint(X[r * ldx + (i - r * d)]) & 0x7f800000u) == 0x7f800000u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, (unsigned long long)bad);
}

// ---- duplicate rows ---------------------------------------------------------------------------------------------------
// 64-bit hash of every row (one wavefront per row): position-keyed mixes of the element bits, summed.  -0.0 hashes as
// +0.0 (torch.unique compares values); the result is never 0 (0 marks an empty table slot).
__global__ __launch_bounds__(256) void row_hash_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                       uint64_t* __restrict__ hash) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    uint64_t h = 0;
    for (int c = lane; c < d; c += 64) {
        const float x = X[row * ldx + c];
        const uint32_t b = (x == 0.f) ? 0u : __float_as_uint(x);
        h += mix64(((uint64_t)(uint32_t)c << 32 | b) + 0x9E3779B97F4A7C15ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    h = mix64(h);
    if (lane == 0) hash[row] = h ? h : 1ull;
}

// insert every row's hash into an open-addressing table (linear probing); the slot remembers the smallest row index
__global__ __launch_bounds__(256) void dedup_insert_kernel(const uint64_t* __restrict__ hash, int64_t n,
                                                           unsigned long long* __restrict__ table, int* __restrict__ slot_rep,
                                                           int log2size, int* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = hash[i];
    const uint64_t mask = ((uint64_t)1 << log2size) - 1;
    uint64_t slot = (h * 0x9E3779B97F4A7C15ull) >> (64 - log2size);
    for (;;) {
        const unsigned long long prev = atomicCAS(&table[slot], 0ull, h);
        if (prev == 0ull || prev == h) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&slot_rep[slot], (int)i);
    slot_of[i] = (int)slot;
}

// rep[i] = smallest row index whose row equals row i (i itself for first occurrences).  A row whose hash slot is led
// by a DIFFERENT row (a 64-bit collision) is counted in counters[1]; the caller then falls back to the exact sort.
__global__ __launch_bounds__(256) void dedup_resolve_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                            const int* __restrict__ slot_of, const int* __restrict__ slot_rep,
                                                            int* __restrict__ rep, unsigned* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = slot_rep[slot_of[i]];
    if (r == (int)i) { rep[i] = (int)i; return; }
    bool same = true;
    for (int c = 0; c < d && same; ++c) same = X[i * ldx + c] == X[(int64_t)r * ldx + c];
    rep[i] = same ? r : (int)i;
    atomicAdd(&counters[same ? 0 : 1], 1u);
}

// ---- PCA initialisation ---------------------------------------------------------------------------------------------
// column sums of a row strip per workgroup (fp32 over <= rows_per_wg rows), combined in fixed order.  Vector form
// (d and ldx multiples of 4, 16-byte aligned rows): thread = (float4 column, row lane), 256 / (d / 4) row lanes walk
// the strip interleaved with four loads in flight each; the lanes' sums meet in LDS in lane order.  Scalar form:
// thread = column.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                             int64_t rows_per_wg, int vec, float* __restrict__ partial) {
    __shared__ float red[256 * 4];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
    if (vec) {
        const int nv = d / 4;          // float4 columns (<= 64)
        const int lanes = 256 / nv;    // row lanes
        const int c = threadIdx.x % nv, rl = threadIdx.x / nv;
        float4 s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < lanes) {
            int64_t r = r0 + rl;
            for (; r + 3 * lanes < r1; r += 4 * lanes) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(X + (r + (int64_t)u * lanes) * ldx + 4 * c);
                    s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
                }
            }
            for (; r < r1; r += lanes) {
                const float4 v = *reinterpret_cast<const float4*>(X + r * ldx + 4 * c);
                s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
            }
        }
        const float4 t = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                                     (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
        *reinterpret_cast<float4*>(red + threadIdx.x * 4) = t;
        __syncthreads();
        if (threadIdx.x < nv) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 0; l < lanes; ++l) {
                const float4 v = *reinterpret_cast<const float4*>(red + (l * nv + threadIdx.x) * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * d + 4 * threadIdx.x) = a;
        }
        return;
    }
    for (int c = threadIdx.x; c < d; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += X[r * ldx + c];
        partial[(size_t)blockIdx.x * d + c] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int n_wg, int d, double inv_n,
                                                           float* __restrict__ mean) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int w = 0; w < n_wg; ++w) s += (double)partial[(size_t)w * d + c];
    mean[c] = (float)(s * inv_n);
}

// Gram matrix of the centred block, G = sum_k (x_k - mean)(x_k - mean)^T, per workgroup over its row strip: DT
// wavefronts, wavefront w owns the 32-row band m0 = 32 w of G (DT accumulator tiles); strips of 32 rows go through LDS
// and v_mfma_f32_32x32x2_f32 contracts two rows per instruction (A[m][k] = xc[k][m0 + m], B[k][n] = xc[k][n0 + n]).
template <int DT>
__global__ __launch_bounds__(64 * DT) void gram_partial_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                               const float* __restrict__ mean, int64_t rows_per_wg,
                                                               float* __restrict__ partial) {
    constexpr int W = 32 * DT;
    __shared__ float tile[32 * W];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
    for (int64_t rt = r0; rt < r1; rt += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * W; e += 64 * DT) {
            const int rr = e / W, c = e - rr * W;
            const int64_t row = rt + rr;
            tile[e] = (row < r1 && c < d) ? X[row * ldx + c] - mean[c] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const float a = tile[(2 * s + h) * W + 32 * w + i];
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const float b = tile[(2 * s + h) * W + 32 * j + i];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * W * W;
#pragma unroll
    for (int j = 0; j < DT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)m * W + 32 * j + i] = acc[j][r];
        }
}
__global__ __launch_bounds__(256) void gram_final_kernel(const float* __restrict__ partial, int n_wg, int W, int d,
                                                         double* __restrict__ G) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d * d) return;
    const int m = e / d, c = e - m * d;
    double s = 0.0;
    for (int w = 0; w < n_wg; ++w) s += (double)partial[(size_t)w * W * W + (size_t)m * W + c];
    G[e] = s;
}

// scores E = (X - mean) V for nc <= 4 components, one wavefront per row
__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                      const float* __restrict__ mean, const float* __restrict__ V, int nc,
                                                      float* __restrict__ E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < d; c += 64) {
        const float x = X[row * ldx + c] - mean[c];
        for (int k = 0; k < nc; ++k) s[k] += x * V[(size_t)c * nc + k];
    }
    for (int k = 0; k < nc; ++k) {
        const float t = wave_sum(s[k]);
        if (lane == 0) E[row * nc + k] = t;
    }
}

// ---- eigen-decomposition of the D x D Gram matrix (PCA initialisation, spectral_embedding/pca.py:169-178) ------------------
// One-sided Jacobi (Hestenes) in ONE workgroup: the columns of W (= G at the start) are orthogonalised by plane rotations,
// the same rotations accumulate in V.  For a symmetric positive semi-definite G the limit is W = V diag(lambda): column
// norms = eigenvalues, V = eigenvectors (relative accuracy ~1e-14 after 6-9 sweeps at D <= 256).  A sweep is D' - 1
// rounds of D' / 2 disjoint column pairs (round-robin tournament), a pair is worked by `lp` lanes of one wavefront
// (three dot products, a butterfly reduction that leaves identical bits in every lane, two column updates in W and V);
// rounds are separated by a workgroup barrier.  No host read anywhere: the decomposition can run on a side stream under
// the kNN search (the library `eigh` reads its status word back).  W and V live in the caller's workspace (L2-resident).
constexpr int EIG_TH = 1024;
constexpr int EIG_MAX_D = 256;
__global__ __launch_bounds__(EIG_TH) void eigh_jacobi_kernel(const double* __restrict__ G, int d, double* W, double* V,
                                                             double* __restrict__ evals, double* __restrict__ evecs, int max_sweeps,
                                                             double tol) {
    __shared__ int rotated;
    __shared__ double sig[EIG_MAX_D];
    __shared__ int perm[EIG_MAX_D];
    const int tid = threadIdx.x;
    const int dp = d + (d & 1), n_pairs = dp / 2, m = dp - 1;
    int lp = 64;
    while (lp * n_pairs > EIG_TH) lp >>= 1;  // n_pairs <= 128: at least 8 lanes per pair
    const int pair = tid / lp, l = tid % lp;
    for (int e = tid; e < d * d; e += EIG_TH) {
        W[e] = G[e];  // symmetric: column-major = row-major
        V[e] = (e / d == e % d) ? 1.0 : 0.0;
    }
    __threadfence_block();  // one workgroup = one CU = one L1: workgroup scope is enough, no L2 write-back per round
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < m; ++r) {
            if (pair < n_pairs) {
                int a, b;
                if (pair == 0) { a = dp - 1; b = r; }
                else { a = (r + pair) % m; b = (r - pair + m) % m; }
                const int p = a < b ? a : b, q = a < b ? b : a;
                if (q < d) {  // odd d: the pair with the phantom column rests
                    double* wp = W + (size_t)p * d;
                    double* wq = W + (size_t)q * d;
                    double al = 0.0, be = 0.0, ga = 0.0;
                    for (int i = l; i < d; i += lp) {
                        const double x = wp[i], y = wq[i];
                        al = fma(x, x, al); be = fma(y, y, be); ga = fma(x, y, ga);
                    }
                    for (int o = lp >> 1; o > 0; o >>= 1) {
                        al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64);
                    }
                    const double ab = al * be;
                    if (ab != 0.0 && fabs(ga) > tol * sqrt(ab)) {
                        const double z = (be - al) / (2.0 * ga);
                        const double t = (z < 0.0 ? -1.0 : 1.0) / (fabs(z) + sqrt(1.0 + z * z));
                        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                        double* vp = V + (size_t)p * d;
                        double* vq = V + (size_t)q * d;
                        for (int i = l; i < d; i += lp) {
                            const double x = wp[i], y = wq[i];
                            wp[i] = c * x - sn * y; wq[i] = sn * x + c * y;
                            const double u = vp[i], v = vq[i];
                            vp[i] = c * u - sn * v; vq[i] = sn * u + c * v;
                        }
                        if (l == 0) rotated = 1;
                    }
                }
            }
            __threadfence_block();  // one workgroup = one CU = one L1: workgroup scope is enough, no L2 write-back per round
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (!any) break;
    }
    for (int j = tid; j < d; j += EIG_TH) {
        double s2 = 0.0;
        for (int i = 0; i < d; ++i) { const double x = W[(size_t)j * d + i]; s2 = fma(x, x, s2); }
        sig[j] = sqrt(s2);
    }
    __syncthreads();
    for (int j = tid; j < d; j += EIG_TH) {  // descending, ties by column
        int rank = 0;
        for (int k = 0; k < d; ++k) rank += (sig[k] > sig[j] || (sig[k] == sig[j] && k < j)) ? 1 : 0;
        evals[rank] = sig[j];
        perm[rank] = j;
    }
    __syncthreads();
    for (int e = tid; e < d * d; e += EIG_TH) {  // evecs (d, d) row-major: column r = eigenvector of the r-th largest eigenvalue
        const int i = e / d, r = e % d;
        evecs[e] = V[(size_t)perm[r] * d + i];
    }
}

// ---- the LEADING eigenpairs only (round 6; what the PCA initialisation needs: n_components <= 4 of D <= 256) ----------------
// The one-workgroup Jacobi above decomposes the whole matrix: 12.5 ms at D = 128 and 127 ms at D = 256 -- hidden under the kNN
// search of a single-GPU fit, but as long as a rank's whole kNN stage in an 8-rank fit (profiles/r06_rank_share_c4_first.jsonl:
// the projection waited for it).  Here, still ONE workgroup and no host read (same bits on every rank):
//   1. Householder tridiagonalisation of the full symmetric matrix in the workspace A (L2-resident): D - 2 steps of one
//      column-sum product p = beta A v (thread (g, j) sums rows i = g mod G of column j: coalesced), w = p - (beta v.p / 2) v and
//      the rank-2 update A -= v w^T + w v^T (rounded products added, so A stays symmetric bit for bit); reflector k stays in
//      row k of A for step 4;
//   2. the nc largest eigenvalues of the tridiagonal matrix by Sturm-count MULTISECTION: 256 threads per eigenvalue evaluate
//      the count at 256 interior points of the bracket, 9 rounds shrink it 257^9-fold (below one ulp);
//   3. inverse iteration on T - lambda I (tridiagonal LU with partial pivoting, one thread per eigenvalue, three solves from a
//      hashed start vector), then modified Gram-Schmidt among the nc vectors (equal / close eigenvalues);
//   4. back-transformation by the reflectors in reverse order, one wavefront per vector (butterfly sums: same bits in all lanes).
constexpr int EIGT_TH = 1024;
constexpr int EIGT_MAX_NC = 4;
__device__ __forceinline__ double wave_sum_f64(double s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
__global__ __launch_bounds__(EIGT_TH) void eigh_top_kernel(const double* __restrict__ G, int d, int nc, double* A,
                                                           double* __restrict__ evals, double* __restrict__ evecs) {
    constexpr int MD = EIG_MAX_D;
    __shared__ double diag[MD], off[MD], betas[MD];
    __shared__ double scal[8];
    __shared__ double un[MD + 4 * 4 * MD + EIGT_MAX_NC * MD];   // phase 1: x, v, w, part[1024]; phase 2: e2, lu[4][4][MD], y[4][MD]
    __shared__ unsigned char piv[EIGT_MAX_NC][MD];
    __shared__ int cnt[EIGT_MAX_NC];
    __shared__ double lam[EIGT_MAX_NC];
    double* x = un;
    double* v = un + MD;
    double* w = un + 2 * MD;
    double* part = un + 3 * MD;       // G_ * dc = 1024 entries
    double* e2 = un;
    double* lu = un + MD;             // [q][4][MD]: dl, b, c, du2
    double* yv = un + MD + 4 * 4 * MD;    // [q][MD]
    const int tid = threadIdx.x, lane = tid & 63;
    int dc = 64;
    while (dc < d) dc <<= 1;
    const int G_ = EIGT_TH / dc, g = tid / dc, j = tid % dc;
    const int npl = dc / 64;          // elements of a length-d vector a lane of ONE wavefront holds (lane + 64 c)
    for (int e = tid; e < d * d; e += EIGT_TH) A[e] = G[e];
    __threadfence_block();  // one workgroup = one CU = one L1 (as in eigh_jacobi_kernel)
    __syncthreads();
    // 1. tridiagonalisation
    for (int k = 0; k + 2 < d; ++k) {
        if (g == 0 && j < d) x[j] = j > k ? A[(size_t)k * d + j] : 0.0;
        if (tid == 0) diag[k] = A[(size_t)k * d + k];
        __syncthreads();
        if (tid < 64) {
            double s = 0.0;
            for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; const double xv = jj < d ? x[jj] : 0.0; s = fma(xv, xv, s); }
            s = wave_sum_f64(s);
            const double x1 = x[k + 1];
            double alpha = 0.0, beta = 0.0;
            if (s > 0.0) {
                alpha = x1 >= 0.0 ? -sqrt(s) : sqrt(s);
                beta = 2.0 / (2.0 * s - 2.0 * alpha * x1);
            }
            for (int c = 0; c < npl; ++c) {
                const int jj = lane + 64 * c;
                if (jj < MD) v[jj] = (jj < d && s > 0.0) ? (jj == k + 1 ? x1 - alpha : x[jj]) : 0.0;
            }
            if (lane == 0) { off[k] = alpha; betas[k] = beta; scal[0] = beta; }
        }
        __syncthreads();
        const double beta = scal[0];
        if (beta != 0.0) {   // the same for every thread
            const bool mine = j < d && j > k;
            if (g == 0 && mine) A[(size_t)k * d + j] = v[j];
            if (mine) {
                double s = 0.0;
                for (int i = k + 1 + g; i < d; i += G_) s = fma(A[(size_t)i * d + j], v[i], s);
                part[g * dc + j] = s;
            }
            __syncthreads();
            if (tid < 64) {
                double pj[EIG_MAX_D / 64];
                double kp = 0.0;
                for (int c = 0; c < npl; ++c) {
                    const int jj = lane + 64 * c;
                    pj[c] = 0.0;
                    if (jj < d && jj > k) {
                        double s = 0.0;
                        for (int gg = 0; gg < G_; ++gg) s += part[gg * dc + jj];
                        pj[c] = beta * s;
                        kp = fma(v[jj], pj[c], kp);
                    }
                }
                kp = wave_sum_f64(kp);
                const double K = 0.5 * beta * kp;
                for (int c = 0; c < npl; ++c) {
                    const int jj = lane + 64 * c;
                    if (jj < MD) w[jj] = (jj < d && jj > k) ? pj[c] - K * v[jj] : 0.0;
                }
            }
            __syncthreads();
            if (mine) {
                const double vj = v[j], wj = w[j];
                for (int i = k + 1 + g; i < d; i += G_) {
                    const double t1 = v[i] * wj, t2 = w[i] * vj;     // (i, j) and (j, i) add the same two rounded products
                    A[(size_t)i * d + j] -= t1 + t2;
                }
            }
            __threadfence_block();
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (d >= 2) { diag[d - 2] = A[(size_t)(d - 2) * d + d - 2]; off[d - 2] = A[(size_t)(d - 2) * d + d - 1]; }
        diag[d - 1] = A[(size_t)(d - 1) * d + d - 1];
        off[d - 1] = 0.0;
    }
    __syncthreads();
    // 2. brackets (Gershgorin), pivot floor, squared off-diagonals
    if (tid < 64) {
        double tn = 0.0, glo = 1e300, ghi = -1e300;
        for (int c = 0; c < npl; ++c) {
            const int jj = lane + 64 * c;
            if (jj < d) {
                const double r = fabs(off[jj]) + (jj > 0 ? fabs(off[jj - 1]) : 0.0);
                tn = fmax(tn, fmax(fabs(diag[jj]), fabs(off[jj])));
                glo = fmin(glo, diag[jj] - r);
                ghi = fmax(ghi, diag[jj] + r);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tn = fmax(tn, __shfl_xor(tn, o, 64)); glo = fmin(glo, __shfl_xor(glo, o, 64)); ghi = fmax(ghi, __shfl_xor(ghi, o, 64));
        }
        if (lane == 0) {
            const double et = 2.220446049250313e-16 * tn;
            const double pivmin = fmax(2.2250738585072014e-298, et * et);
            const double span = ghi - glo;
            scal[1] = tn; scal[2] = glo - (1e-12 * span + pivmin); scal[3] = ghi + (1e-12 * span + pivmin); scal[4] = pivmin;
        }
    }
    __syncthreads();   // phase 1's x, v, w, part are dead from here
    if (g == 0 && j < d) e2[j] = off[j] * off[j];
    const double tn = scal[1], pivmin = scal[4];
    const int q = tid >> 8, t = tid & 255;
    const bool active = q < nc;
    if (tn == 0.0) {       // the zero matrix
        if (tid < nc) evals[tid] = 0.0;
        for (int e = tid; e < d * nc; e += EIGT_TH) evecs[e] = (e / nc == e % nc) ? 1.0 : 0.0;
        return;
    }
    __syncthreads();
    {
        double lo = scal[2], hi = scal[3];
        const int idx = d - 1 - q;      // ascending index of the q-th largest eigenvalue
        for (int it = 0; it < 9; ++it) {
            if (t == 0 && active) cnt[q] = 0;
            __syncthreads();
            if (active) {
                const double xs = lo + (hi - lo) * ((double)(t + 1) / 257.0);
                int c = 0;
                double qq = diag[0] - xs;
                if (fabs(qq) < pivmin) qq = -pivmin;
                c += qq < 0.0;
                for (int i = 1; i < d; ++i) {
                    qq = diag[i] - xs - e2[i - 1] / qq;
                    if (fabs(qq) < pivmin) qq = -pivmin;
                    c += qq < 0.0;
                }
                if (c <= idx) atomicAdd(&cnt[q], 1);
            }
            __syncthreads();
            if (active) {
                const int T = cnt[q];
                const double nlo = T > 0 ? lo + (hi - lo) * ((double)T / 257.0) : lo;
                const double nhi = T < 256 ? lo + (hi - lo) * ((double)(T + 1) / 257.0) : hi;
                lo = nlo; hi = nhi;
            }
            __syncthreads();
        }
        if (active && t == 0) lam[q] = 0.5 * (lo + hi);
    }
    __syncthreads();
    // 3. inverse iteration, one thread per eigenvalue
    if (active && t == 0) {
        double* dl = lu + (size_t)q * 4 * MD;
        double* b = dl + MD;
        double* c = b + MD;
        double* du2 = c + MD;
        double* y = yv + (size_t)q * MD;
        const double l = lam[q], eps3 = 2.220446049250313e-16 * tn;
        for (int i = 0; i < d; ++i) { dl[i] = off[i]; b[i] = diag[i] - l; c[i] = i + 1 < d ? off[i] : 0.0; du2[i] = 0.0; piv[q][i] = 0; }
        for (int i = 0; i + 1 < d; ++i) {
            if (fabs(b[i]) >= fabs(dl[i])) {
                if (b[i] == 0.0) b[i] = eps3;
                const double f = dl[i] / b[i];
                dl[i] = f; b[i + 1] -= f * c[i];
            } else {
                const double f = b[i] / dl[i];
                b[i] = dl[i]; dl[i] = f;
                const double tmp = c[i];
                c[i] = b[i + 1]; b[i + 1] = tmp - f * c[i];
                du2[i] = c[i + 1]; c[i + 1] = -f * du2[i];
                piv[q][i] = 1;
            }
        }
        if (b[d - 1] == 0.0) b[d - 1] = eps3;
        for (int i = 0; i < d; ++i) y[i] = (double)(mix64((uint64_t)(i * 4 + q + 1)) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        for (int itr = 0; itr < 3; ++itr) {
            for (int i = 0; i + 1 < d; ++i) {
                if (!piv[q][i]) y[i + 1] -= dl[i] * y[i];
                else { const double tmp = y[i]; y[i] = y[i + 1]; y[i + 1] = tmp - dl[i] * y[i + 1]; }
            }
            y[d - 1] /= b[d - 1];
            if (d >= 2) y[d - 2] = (y[d - 2] - c[d - 2] * y[d - 1]) / b[d - 2];
            for (int i = d - 3; i >= 0; --i) y[i] = (y[i] - c[i] * y[i + 1] - du2[i] * y[i + 2]) / b[i];
            double m = 0.0;
            for (int i = 0; i < d; ++i) m = fmax(m, fabs(y[i]));
            const double r = 1.0 / m;
            for (int i = 0; i < d; ++i) y[i] *= r;
        }
    }
    __syncthreads();
    // modified Gram-Schmidt among the nc vectors (wavefront 0), unit norm
    if (tid < 64) {
        for (int a = 0; a < nc; ++a) {
            double ya[EIG_MAX_D / 64];
            for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; ya[c] = jj < d ? yv[a * MD + jj] : 0.0; }
            for (int p = 0; p < a; ++p) {
                double s = 0.0;
                for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; if (jj < d) s = fma(ya[c], yv[p * MD + jj], s); }
                s = wave_sum_f64(s);
                for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; if (jj < d) ya[c] -= s * yv[p * MD + jj]; }
            }
            double s = 0.0;
            for (int c = 0; c < npl; ++c) s = fma(ya[c], ya[c], s);
            s = 1.0 / sqrt(wave_sum_f64(s));
            for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; if (jj < d) yv[a * MD + jj] = ya[c] * s; }
        }
    }
    __syncthreads();
    // 4. back-transformation: wavefront a applies the reflectors d-3 .. 0 to vector a
    const int a = tid >> 6;
    if (a < nc) {
        double ya[EIG_MAX_D / 64];
        for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; ya[c] = jj < d ? yv[a * MD + jj] : 0.0; }
        for (int k = d - 3; k >= 0; --k) {
            const double beta = betas[k];
            if (beta == 0.0) continue;
            double vr[EIG_MAX_D / 64];
            double s = 0.0;
            for (int c = 0; c < npl; ++c) {
                const int jj = lane + 64 * c;
                vr[c] = (jj < d && jj > k) ? A[(size_t)k * d + jj] : 0.0;
                s = fma(vr[c], ya[c], s);
            }
            s = beta * wave_sum_f64(s);
            for (int c = 0; c < npl; ++c) ya[c] -= s * vr[c];
        }
        for (int c = 0; c < npl; ++c) { const int jj = lane + 64 * c; if (jj < d) evecs[(size_t)jj * nc + a] = ya[c]; }
        if (lane == 0) evals[a] = lam[a];
    }
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* utils/validation.py:308: *count (device uint64, caller-zeroed) += number of inf / nan entries of the (n, d) block. */
int tdr_nonfinite_count_f32(const float* X, int64_t n, int d, int64_t ldx, void* count, void* stream) {
    if (!X || !count || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    int64_t blocks = (n * (int64_t)d / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nonfinite_count_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, n, d, ldx,
                       (unsigned long long*)count);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* workspace of tdr_dedup_rows_f32 in bytes */
int64_t tdr_dedup_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    int log2size = 4;
    while (((int64_t)1 << log2size) < 2 * n) ++log2size;
    return n * 8 + ((int64_t)1 << log2size) * (8 + 4) + n * 4;
}

/* base.py:132-148 (torch.unique(dim=0)): rep (n) int32 = smallest index of a row equal to row i; counters (2 x uint32,
 * device) = {rows that duplicate an earlier row, rows whose 64-bit hash collided with a different row (then rep is not
 * reliable and the caller must use an exact method)}.  n < 2^31. */
int tdr_dedup_rows_f32(const float* X, int64_t n, int d, int64_t ldx, int32_t* rep, void* counters, void* ws, int64_t ws_bytes,
                       void* stream) {
    if (!X || !rep || !counters || !ws || n <= 0 || d <= 0 || ldx < d || n >= 0x7fffffffLL) return TDR_ERR_BAD_ARG;
    if (ws_bytes < tdr_dedup_workspace_bytes(n)) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int log2size = 4;
    while (((int64_t)1 << log2size) < 2 * n) ++log2size;
    const int64_t slots = (int64_t)1 << log2size;
    uint64_t* hash = (uint64_t*)ws;
    unsigned long long* table = (unsigned long long*)(hash + n);
    int* slot_rep = (int*)(table + slots);
    int* slot_of = slot_rep + slots;
    hipError_t e = hipMemsetAsync(table, 0, slots * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(slot_rep, 0x7f, slots * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 8, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(row_hash_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X, n, d, ldx, hash);
    hipLaunchKernelGGL(dedup_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint64_t*)hash, n, table,
                       slot_rep, log2size, slot_of);
    hipLaunchKernelGGL(dedup_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, n, d, ldx,
                       (const int*)slot_of, (const int*)slot_rep, rep, (unsigned*)counters);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* workspace (floats) of tdr_pca_gram_f32 */
int64_t tdr_pca_gram_workspace_floats(int64_t n, int d) {
    if (n <= 0 || d <= 0 || d > 256) return 0;
    const int W = 32 * ((d + 31) / 32);
    int64_t n_wg = (n + 1023) / 1024;
    if (n_wg > 1024) n_wg = 1024;
    return n_wg * ((int64_t)W * W + d);
}

/* spectral_embedding/pca.py:151-160: mean (d) fp32 = column means, G (d, d) fp64 = Gram matrix of the centred block
 * (deterministic: fixed partition into row strips, partial results combined in order).  d <= 256. */
int tdr_pca_gram_f32(const float* X, int64_t n, int d, int64_t ldx, float* mean, double* G, float* ws, int64_t ws_floats,
                     void* stream) {
    if (!X || !mean || !G || !ws || n <= 0 || d <= 0 || ldx < d) return TDR_ERR_BAD_ARG;
    if (d > 256) return TDR_ERR_UNSUPPORTED;
    if (ws_floats < tdr_pca_gram_workspace_floats(n, d)) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int DT = (d + 31) / 32, W = 32 * DT;
    int64_t n_wg = (n + 1023) / 1024;  // several workgroups per CU hide each other's load latency
    if (n_wg > 1024) n_wg = 1024;
    int64_t rows_per_wg = (n + n_wg - 1) / n_wg;
    rows_per_wg = (rows_per_wg + 31) / 32 * 32;
    n_wg = (n + rows_per_wg - 1) / rows_per_wg;
    float* part_g = ws;
    float* part_s = ws + n_wg * (int64_t)W * W;
    const int vec = (d % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)part_s & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, X, n, d, ldx, rows_per_wg, vec, part_s);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, (const float*)part_s, (int)n_wg, d,
                       1.0 / (double)n, mean);
#define TDR_GRAM(DTV)                                                                                                    \
    hipLaunchKernelGGL(gram_partial_kernel<DTV>, dim3((unsigned)n_wg), dim3(64 * DTV), 0, st, X, n, d, ldx, (const float*)mean, \
                       rows_per_wg, part_g)
    switch (DT) {
        case 1: TDR_GRAM(1); break;
        case 2: TDR_GRAM(2); break;
        case 3: TDR_GRAM(3); break;
        case 4: TDR_GRAM(4); break;
        case 5: TDR_GRAM(5); break;
        case 6: TDR_GRAM(6); break;
        case 7: TDR_GRAM(7); break;
        default: TDR_GRAM(8); break;
    }
#undef TDR_GRAM
    hipLaunchKernelGGL(gram_final_kernel, dim3((unsigned)((d * d + 255) / 256)), dim3(256), 0, st, (const float*)part_g, (int)n_wg, W,
                       d, G);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* spectral_embedding/pca.py:171-178: E (n, nc) = (X - mean) V, V (d, nc) row-major, nc <= 4. */
int tdr_pca_project_f32(const float* X, int64_t n, int d, int64_t ldx, const float* mean, const float* V, int nc, float* E,
                        void* stream) {
    if (!X || !mean || !V || !E || n <= 0 || d <= 0 || ldx < d || nc <= 0) return TDR_ERR_BAD_ARG;
    if (nc > 4) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X, n, d, ldx, mean, V, nc, E);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}
/* Eigen-decomposition of a symmetric positive semi-definite d x d matrix (d <= 256) without a host read: evals (d)
 * descending, evecs (d, d) row-major with column r the eigenvector of evals[r]; ws = 2 d^2 doubles. */
int tdr_eigh_jacobi_f64(const double* G, int d, double* evals, double* evecs, double* ws, void* stream) {
    if (!G || !evals || !evecs || !ws || d <= 0) return TDR_ERR_BAD_ARG;
    if (d > EIG_MAX_D) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(eigh_jacobi_kernel, dim3(1), dim3(EIG_TH), 0, (hipStream_t)stream, G, d, ws, ws + (size_t)d * d, evals, evecs, 40,
                       1e-15);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* The nc <= 4 LARGEST eigenpairs of a symmetric d x d matrix (d <= 256, nc <= d) without a host read: evals (nc) descending,
 * evecs (d, nc) row-major with unit columns; ws = d^2 doubles (eigh_top_kernel: Householder + Sturm multisection + inverse
 * iteration in one workgroup; ~0.5 ms at d = 128 where the full Jacobi decomposition takes 12.5). */
int tdr_eigh_top_f64(const double* G, int d, int nc, double* evals, double* evecs, double* ws, void* stream) {
    if (!G || !evals || !evecs || !ws || d <= 0 || nc <= 0 || nc > d) return TDR_ERR_BAD_ARG;
    if (d > EIG_MAX_D || nc > EIGT_MAX_NC) return TDR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(eigh_top_kernel, dim3(1), dim3(EIGT_TH), 0, (hipStream_t)stream, G, d, nc, ws, evals, evecs);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
