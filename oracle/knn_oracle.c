/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the exact kNN / pairwise-distance path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (torchdr_amd/) never does.
 *
 * What it restates (reference = /root/reference/torchdr, read-only):
 *   - distance/torch.py:82-91   X_norm = (X**2).sum(-1);  C = X_norm[:,None] + Y_norm[None,:] - 2*(X @ Y.T)
 *   - distance/torch.py:93-95   euclidean = sqrt(clamp(C, 0))
 *   - distance/torch.py:96-98   manhattan = (X[:,None,:] - Y[None,:,:]).abs().sum(-1)
 *   - distance/torch.py:99-100  angular   = -(X @ Y.T)            (no normalisation)
 *   - distance/torch.py:111-116 self exclusion (diag += 1e12)     == "skip j == i" whenever k < N
 *   - utils/utils.py:215-216    kmin = topk(k, largest=False), indices -> int32
 *
 * The reference's arithmetic lives in two third-party pieces that this file restates
 * bit-for-bit as measured in the build container (torch 2.10.0+rocm7.0 CPU, MKL 2024.2):
 *   (1) MKL sgemm, K <= 256: every C[i][j] is ONE k-ordered fp32 FMA chain starting from 0
 *       (verified bit-identical on 4096x8192x128, 256x4096x256, ...).  For K > ~380 MKL
 *       splits K adaptively into blocks; this oracle keeps the single chain (documented
 *       divergence, DESIGN.md "parity").
 *   (2) ATen sum over a contiguous last dim (aten/src/ATen/native/cpu/SumKernel.cpp,
 *       cascade_sum / vectorized_inner_sum) in its AVX2 build (8-lane vectors, 4-way ILP,
 *       4 cascade levels).  oracle_sqnorms_f32 restates that summation order exactly
 *       (verified bit-identical for every D in 1..4096 probed).
 * Ties: the oracle orders candidates by (distance, index) lexicographically -- the canonical
 * order of the parity protocol (torch.topk's own tie order is unspecified).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- (2) ATen AVX2 sum order -------------------------------------------------------- */

static int ceil_log2_i64(int64_t x) {
    if (x <= 2) return 1;
    uint64_t v = (uint64_t)(x - 1);
    int bits = 0;
    while (v) { bits++; v >>= 1; }
    return bits;
}

/* multi_row_sum<nrows=4> over `size` items, each item = `w` lanes (w = 8 vector path, 1 scalar).
 * item i of row r lives at sq[(i*4 + r)*w + lane]. out[r*w + lane]. */
static void multi_row_sum4(const float *sq, int64_t size, int w, float *out) {
    enum { NL = 4, NR = 4 };
    float acc[NL][NR][8];
    memset(acc, 0, sizeof(acc));
    int lp = ceil_log2_i64(size) / NL;
    if (lp < 4) lp = 4;
    const int64_t level_step = (int64_t)1 << lp;
    const int64_t level_mask = level_step - 1;
    int64_t i = 0;
    for (; i + level_step <= size;) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int r = 0; r < NR; ++r)
                for (int l = 0; l < w; ++l) acc[0][r][l] += sq[(i * 4 + r) * w + l];
        for (int j = 1; j < NL; ++j) {
            for (int r = 0; r < NR; ++r)
                for (int l = 0; l < w; ++l) { acc[j][r][l] += acc[j - 1][r][l]; acc[j - 1][r][l] = 0.f; }
            const int64_t mask = level_mask << (j * lp);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int r = 0; r < NR; ++r)
            for (int l = 0; l < w; ++l) acc[0][r][l] += sq[(i * 4 + r) * w + l];
    for (int j = 1; j < NL; ++j)
        for (int r = 0; r < NR; ++r)
            for (int l = 0; l < w; ++l) acc[0][r][l] += acc[j][r][l];
    for (int r = 0; r < NR; ++r)
        for (int l = 0; l < w; ++l) out[r * w + l] = acc[0][r][l];
}

/* row_sum: items of width w; returns w partial sums in out[] */
static void row_sum_w(const float *sq, int64_t size, int w, float *out) {
    float part[4 * 8];
    const int64_t size_ilp = size / 4;
    multi_row_sum4(sq, size_ilp, w, part);
    for (int64_t i = size_ilp * 4; i < size; ++i)
        for (int l = 0; l < w; ++l) part[l] += sq[i * w + l];
    for (int r = 1; r < 4; ++r)
        for (int l = 0; l < w; ++l) part[l] += part[r * w + l];
    for (int l = 0; l < w; ++l) out[l] = part[l];
}

/* ATen's sum of d contiguous floats (the order every last-dim .sum(-1) of the reference's CPU path uses) */
static float sum_aten(const float *sq, int d) {
    const int V = 8;
    if (d >= V) {
        const int64_t vec_size = d / V;
        float p[8];
        row_sum_w(sq, vec_size, V, p);
        float fin = 0.f;
        for (int k = (int)(vec_size * V); k < d; ++k) fin += sq[k];
        for (int l = 0; l < V; ++l) fin += p[l];
        return fin;
    }
    float p1[1];
    row_sum_w(sq, d, 1, p1);
    return p1[0];
}

static float sqnorm_aten(const float *x, int d, float *sq /* scratch d floats */) {
    for (int k = 0; k < d; ++k) sq[k] = x[k] * x[k]; /* X**2 is materialised (rounded) first */
    return sum_aten(sq, d);
}

/* distance/torch.py:96-98: (x - y).abs().sum(-1) -- difference materialised (rounded), |.| exact, ATen sum order */
static float l1_aten(const float *x, const float *y, int d, float *sq /* scratch d floats */) {
    for (int k = 0; k < d; ++k) sq[k] = fabsf(x[k] - y[k]);
    return sum_aten(sq, d);
}

void oracle_sqnorms_f32(const float *X, int64_t n, int d, float *out) {
#pragma omp parallel
    {
        float *sq = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) out[i] = sqnorm_aten(X + (size_t)i * d, d, sq);
        free(sq);
    }
}

/* ---- (1) distances + exact top-k ------------------------------------------------------ */

static inline uint32_t f2u(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
static inline float u2f(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline uint64_t mkkey(float d, uint32_t idx) { return ((uint64_t)f2u(d) << 32) | idx; }

/* max-heap of k keys */
static void heap_sift(uint64_t *h, int k, int p) {
    for (;;) {
        int l = 2 * p + 1, r = l + 1, m = p;
        if (l < k && h[l] > h[m]) m = l;
        if (r < k && h[r] > h[m]) m = r;
        if (m == p) return;
        uint64_t t = h[p]; h[p] = h[m]; h[m] = t;
        p = m;
    }
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

enum { METRIC_SQEUCLIDEAN = 0, METRIC_EUCLIDEAN = 1, METRIC_ANGULAR = 2, METRIC_MANHATTAN = 3 };

/* dot products of one query against a block of database rows, database stored transposed
 * (YT[k*ldt + j]) so the compiler vectorises over j; each acc[j] is a k-ordered FMA chain. */
static void dots_block(const float *x, const float *YT, size_t ldt, int64_t j0, int64_t nb, int d, float *acc) {
    for (int64_t j = 0; j < nb; ++j) acc[j] = 0.f;
    for (int k = 0; k < d; ++k) {
        const float xv = x[k];
        const float *yt = YT + (size_t)k * ldt + j0;
        for (int64_t j = 0; j < nb; ++j) acc[j] = fmaf(xv, yt[j], acc[j]);
    }
}

/*
 * kNN of queries X[0:nq] (global ids q_offset + i) against database Y[0:n].
 * exclude_self: skip database row j == q_offset + i  (distance/torch.py:111-116).
 * Output rows sorted ascending by (distance, index).  out_full (optional, may be NULL):
 * the dense nq x n distance matrix (raw metric values, diagonal NOT modified).
 * Returns 0, or -1 on bad arguments.
 */
int oracle_knn_f32(const float *X, int64_t nq, int64_t q_offset, const float *Y, int64_t n, int d, int k,
                   int metric, int exclude_self, float *out_d, int32_t *out_i, float *out_full) {
    if (nq < 0 || n <= 0 || d <= 0 || k < 0 || metric < 0 || metric > 3) return -1;
    if (k > 0 && k > n - (exclude_self ? 1 : 0)) return -1;
    float *xn = (float *)malloc(sizeof(float) * (size_t)(nq > 0 ? nq : 1));
    float *yn = (float *)malloc(sizeof(float) * (size_t)n);
    float *YT = (float *)malloc(sizeof(float) * (size_t)n * d);
    if (!xn || !yn || !YT) return -1;
    oracle_sqnorms_f32(X, nq, d, xn);
    oracle_sqnorms_f32(Y, n, d, yn);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j)
        for (int kk = 0; kk < d; ++kk) YT[(size_t)kk * n + j] = Y[(size_t)j * d + kk];
    const int64_t BLK = 2048;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)BLK);
        float *dif = (float *)malloc(sizeof(float) * (size_t)d);
        uint64_t *heap = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k > 0 ? k : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < nq; ++i) {
            const float *x = X + (size_t)i * d;
            int filled = 0;
            for (int64_t j0 = 0; j0 < n; j0 += BLK) {
                const int64_t nb = (n - j0 < BLK) ? (n - j0) : BLK;
                if (metric == METRIC_MANHATTAN)
                    for (int64_t jj = 0; jj < nb; ++jj) acc[jj] = l1_aten(x, Y + (size_t)(j0 + jj) * d, d, dif);
                else
                    dots_block(x, YT, (size_t)n, j0, nb, d, acc);
                for (int64_t jj = 0; jj < nb; ++jj) {
                    const int64_t j = j0 + jj;
                    float c;
                    if (metric == METRIC_MANHATTAN) c = acc[jj];
                    else if (metric == METRIC_ANGULAR) c = -acc[jj];
                    else {
                        const float s = xn[i] + yn[j];
                        const float t = 2.0f * acc[jj];
                        c = s - t;
                    }
                    if (out_full) {
                        float v = c;
                        if (metric == METRIC_EUCLIDEAN) v = sqrtf(c > 0.f ? c : 0.f);
                        out_full[(size_t)i * n + j] = v;
                    }
                    if (k == 0) continue;
                    if (exclude_self && j == q_offset + i) continue;
                    const uint64_t key = mkkey(c, (uint32_t)j);
                    if (filled < k) {
                        heap[filled++] = key;
                        if (filled == k)
                            for (int p = k / 2 - 1; p >= 0; --p) heap_sift(heap, k, p);
                    } else if (key < heap[0]) {
                        heap[0] = key;
                        heap_sift(heap, k, 0);
                    }
                }
            }
            if (k > 0) {
                qsort(heap, (size_t)k, sizeof(uint64_t), cmp_u64);
                for (int p = 0; p < k; ++p) {
                    float c = u2f((uint32_t)(heap[p] >> 32));
                    if (metric == METRIC_EUCLIDEAN) c = sqrtf(c > 0.f ? c : 0.f);
                    out_d[(size_t)i * k + p] = c;
                    out_i[(size_t)i * k + p] = (int32_t)(heap[p] & 0xffffffffu);
                }
            }
        }
        free(acc);
        free(dif);
        free(heap);
    }
    free(xn); free(yn); free(YT);
    return 0;
}
