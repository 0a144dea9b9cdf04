// K4 -- sparse symmetrisation  Q = P + P^T - P o P^T  (or P + P^T) of a row-wise (n, k) affinity.
//
// Replaces utils/sparse.py:7-206 of the reference (flatten_sparse -> int64 keys i*n+j for P and P^T ->
// torch.unique(sorted) -> 2x scatter_add -> pack_to_rowwise).  Instead of sorting 2*n*k 64-bit keys
// globally, the GPU version works row-locally:
//   1. rowsort : each row's (col, val) list is sorted by column and duplicate columns are summed
//                (what scatter_add does to duplicates); rows live in L2-resident (n, k) arrays.
//   2. count   : for each edge i->j, binary-search i in row j.  Missing => row j gains one extra
//                entry (the transpose of a non-mutual edge): atomic in-degree count.
//   3. scan    : row pointers = exclusive scan of (own + incoming) degrees.
//   4. fill    : own entries get  (P_ij + P_ji) - P_ij*P_ji  (P_ji by binary search, 0 if absent);
//                non-mutual edges are appended to row j through an atomic cursor.
//   5. finalize: every row is rank-sorted by column, so the result is deterministic and in the
//                reference's order (torch.unique(sorted=True) => ascending columns, sparse.py:73).
// Output is CSR (rowptr int64, cols int32, vals fp32) plus an optional padded (n, max_deg) view with the
// reference's (0, -1) padding for API parity (pack_to_rowwise, sparse.py:89-135).
// Works on a row CHUNK [row_offset, row_offset + n) of an n_total x n_total matrix when the
// transposed entries have already been routed to their owner (multi-GPU path).
#include "tdr_common.h"
#include <limits.h>

namespace tdr {

// ---- 1. rowsort ---------------------------------------------------------------------------------
// One wavefront per row, up to 256 entries per row (4 per lane).
__global__ __launch_bounds__(256) void sym_rowsort_kernel(const float* __restrict__ vals, const int32_t* __restrict__ cols,
                                                          int64_t n, int k, float* __restrict__ svals,
                                                          int32_t* __restrict__ scols, int32_t* __restrict__ slen) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int32_t* rc = cols + (size_t)row * k;
    const float* rv = vals + (size_t)row * k;
    if (k <= 64) {
        // one entry per lane: duplicates are summed in original order, distinct columns ranked -- two sweeps of
        // wave-uniform (readlane) comparisons instead of the general path's O(k^3) scalar loops
        const bool have = lane < k;
        const int32_t mycol = have ? rc[lane] : INT_MAX;
        const float myval = have ? rv[lane] : 0.f;
        bool first = have;
        float sum = 0.f;
        for (int q = 0; q < k; ++q) {
            const int32_t cq = __builtin_amdgcn_readlane(mycol, q);
            const float vq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), q));
            if (have && cq == mycol) {
                sum = __fadd_rn(sum, vq);
                if (q < lane) first = false;
            }
        }
        unsigned long long fm = __ballot(first);
        const int uniq = __popcll(fm);
        int rank = 0;
        while (fm) {
            const int q = __builtin_ctzll(fm);
            fm &= fm - 1;
            rank += (__builtin_amdgcn_readlane(mycol, q) < mycol) ? 1 : 0;
        }
        if (first) {
            scols[(size_t)row * k + rank] = mycol;
            svals[(size_t)row * k + rank] = sum;
        }
        if (lane >= uniq && lane < k) {
            scols[(size_t)row * k + lane] = INT_MAX;
            svals[(size_t)row * k + lane] = 0.f;
        }
        if (lane == 0) slen[row] = uniq;
        return;
    }
    int uniq_total = 0;
    for (int p0 = 0; p0 < k; p0 += 64) {
        const int p = p0 + lane;
        const bool have = p < k;
        const int32_t mycol = have ? rc[p] : INT_MAX;
        bool first = have;
        int rank = 0;      // number of DISTINCT columns smaller than mine
        float sum = 0.f;   // sum over duplicates in original order (scatter_add order)
        for (int q = 0; q < k; ++q) {
            const int32_t cq = rc[q];
            bool q_first = true;  // is q the first occurrence of its column?
            for (int r = 0; r < q; ++r) if (rc[r] == cq) { q_first = false; break; }
            if (have) {
                if (cq == mycol) { sum = __fadd_rn(sum, rv[q]); if (q < p) first = false; }
                else if (cq < mycol && q_first) rank++;
            }
        }
        if (have && first) {
            scols[(size_t)row * k + rank] = mycol;
            svals[(size_t)row * k + rank] = sum;
        }
        uniq_total += __popcll(__ballot(have && first));
    }
    for (int p = uniq_total + lane; p < k; p += 64) {
        scols[(size_t)row * k + p] = INT_MAX;
        svals[(size_t)row * k + p] = 0.f;
    }
    if (lane == 0) slen[row] = uniq_total;
}

__device__ __forceinline__ int bsearch_row(const int32_t* __restrict__ c, int len, int32_t key) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t v = c[mid];
        if (v < key) lo = mid + 1; else hi = mid;
    }
    return (lo < len && c[lo] == key) ? lo : -1;
}

// ---- 2. count -----------------------------------------------------------------------------------
// Edge (gi -> j): if column j is owned here (row_offset <= j < row_offset + n) and row j lacks gi,
// row j receives one extra entry.  `ext_*`: edges received from other ranks, already transposed
// (ext_row = local row that RECEIVES, ext_col = global source), counted unconditionally when the
// receiving row lacks that column.
// The visit of row j (a random 128-byte line or two of scols, one of svals) is the expensive part of an edge: it is
// made ONCE, here, and the transposed value P_ji (or VT_MISSING) is left in `vt` for the fill pass, which then reads
// nothing at random.
constexpr uint32_t VT_MISSING = 0xFFC0DEADu;  // a NaN pattern no arithmetic produces: row j has no entry for column i
__global__ __launch_bounds__(256) void sym_count_kernel(const float* __restrict__ svals, const int32_t* __restrict__ scols,
                                                        const int32_t* __restrict__ slen, int64_t n, int k, int64_t row_offset,
                                                        int32_t* __restrict__ incnt, uint32_t* __restrict__ vt,
                                                        const int32_t* __restrict__ order) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * k) return;
    int64_t i = idx / k;
    const int p = (int)(idx - i * k);
    // `order` (optional): position -> row.  The rows are VISITED in that order -- the cluster-sorted order of a pruned kNN search,
    // in which a row's neighbours are rows of the same few clusters: the random visits of row j then fall into ~100 KB of rows
    // that the L2 already holds instead of 30 M distinct lines spread over the whole block (sym_count 2.24 -> see DESIGN)
    if (order) { i = order[i]; idx = i * k + p; }
    if (p >= slen[i]) return;
    const int64_t j = scols[idx];
    const int64_t lj = j - row_offset;
    if (lj < 0 || lj >= n) { vt[idx] = 0u; return; }  // transpose belongs to another rank (arrives as an ext edge)
    const int32_t gi = (int32_t)(i + row_offset);
    const int pos = bsearch_row(scols + (size_t)lj * k, slen[lj], gi);
    if (pos < 0) { atomicAdd(&incnt[lj], 1); vt[idx] = VT_MISSING; }
    else vt[idx] = __builtin_bit_cast(uint32_t, svals[(size_t)lj * k + pos]);
}

__global__ __launch_bounds__(256) void sym_count_ext_kernel(const int32_t* __restrict__ scols, const int32_t* __restrict__ slen,
                                                            int k, const int32_t* __restrict__ ext_row,
                                                            const int32_t* __restrict__ ext_col, int64_t n_ext,
                                                            int32_t* __restrict__ incnt) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ext) return;
    const int64_t lj = ext_row[e];
    if (bsearch_row(scols + (size_t)lj * k, slen[lj], ext_col[e]) < 0) atomicAdd(&incnt[lj], 1);
}

// ---- 3. scan (3 small kernels; n <= 2^31) ---------------------------------------------------------
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(256) void scan_partial_kernel(const int32_t* __restrict__ slen, const int32_t* __restrict__ incnt,
                                                           int64_t n, int64_t* __restrict__ block_sums,
                                                           int32_t* __restrict__ max_deg) {
    __shared__ int64_t red[256];
    __shared__ int redm[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
    int64_t s = 0;
    int m = 0;
    for (int t = threadIdx.x; t < SCAN_BLOCK; t += 256) {
        const int64_t i = base + t;
        if (i < n) { const int dg = slen[i] + incnt[i]; s += dg; m = max(m, dg); }
    }
    red[threadIdx.x] = s; redm[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; redm[threadIdx.x] = max(redm[threadIdx.x], redm[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { block_sums[blockIdx.x] = red[0]; atomicMax(max_deg, redm[0]); }
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(int64_t* __restrict__ block_sums, int64_t nb, int64_t* __restrict__ total) {
    // ONE workgroup of 256 threads: exclusive scan of the block sums, 256 at a time with a running carry (a single thread walking
    // the ~1000 sums of the headline took 0.11 ms on the symmetrisation's critical path)
    __shared__ int64_t sh[256];
    __shared__ int64_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 256) {
        const int64_t i = base + t;
        const int64_t v = i < nb ? block_sums[i] : 0;
        sh[t] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int64_t a = t >= o ? sh[t - o] : 0;
            __syncthreads();
            sh[t] += a;
            __syncthreads();
        }
        const int64_t incl = sh[t], c0 = carry;
        if (i < nb) block_sums[i] = c0 + incl - v;
        __syncthreads();
        if (t == 255) carry = c0 + incl;
        __syncthreads();
    }
    if (t == 0) *total = carry;
}

__global__ __launch_bounds__(256) void scan_final_kernel(const int32_t* __restrict__ slen, const int32_t* __restrict__ incnt,
                                                         int64_t n, const int64_t* __restrict__ block_sums,
                                                         int64_t* __restrict__ rowptr) {
    // one block per SCAN_BLOCK rows: serial-in-thread chunks of 4 + block scan
    __shared__ int64_t part[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * 4;
    int64_t v[4];
    int64_t s = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t i = base + t;
        v[t] = (i < n) ? (int64_t)(slen[i] + incnt[i]) : 0;
        s += v[t];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partials
    for (int o = 1; o < 256; o <<= 1) {
        int64_t add = (threadIdx.x >= o) ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = block_sums[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t i = base + t;
        if (i < n) rowptr[i] = run;
        run += v[t];
    }
}

__global__ void scan_tail_kernel(const int64_t* __restrict__ total, int64_t n, int64_t* __restrict__ rowptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) rowptr[n] = *total;
}

// ---- 4. fill --------------------------------------------------------------------------------------
// mode 0: P + P^T - P o P^T ; mode 1: P + P^T   (sparse.py:138-165)
__device__ __forceinline__ float combine(float vP, float vPT, int mode) {
    const float s = __fadd_rn(vP, vPT);
    return mode == 1 ? s : __fsub_rn(s, __fmul_rn(vP, vPT));
}

__global__ __launch_bounds__(256) void sym_fill_kernel(const float* __restrict__ svals, const int32_t* __restrict__ scols,
                                                       const int32_t* __restrict__ slen, int64_t n, int k,
                                                       int64_t row_offset, int mode, const int64_t* __restrict__ rowptr,
                                                       int32_t* __restrict__ cursor, const uint32_t* __restrict__ vtw,
                                                       int32_t* __restrict__ tcols, float* __restrict__ tvals,
                                                       const int32_t* __restrict__ order) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * k) return;
    int64_t i = idx / k;
    const int p = (int)(idx - i * k);
    if (order) { i = order[i]; idx = i * k + p; }      // visit order of the rows (see sym_count_kernel): local appends to the rows j
    if (p >= slen[i]) return;
    const int32_t j = scols[idx];
    const float v = svals[idx];
    const int64_t lj = (int64_t)j - row_offset;
    const uint32_t w = vtw[idx];  // P_ji found by the count pass (0 for remote rows: their transposes arrive via ext edges)
    float vt = 0.f;
    const int32_t gi = (int32_t)(i + row_offset);
    if (w == VT_MISSING) {
        const int64_t dst = rowptr[lj] + slen[lj] + atomicAdd(&cursor[lj], 1);
        tcols[dst] = gi;
        tvals[dst] = combine(0.f, v, mode);
    } else {
        vt = __builtin_bit_cast(float, w);
    }
    const int64_t dst = rowptr[i] + p;
    tcols[dst] = j;
    tvals[dst] = combine(v, vt, mode);
}

// Edges received from other ranks (already transposed): (ext_row <- ext_col, v = P[ext_col][ext_row]).
// If the receiving row has that column, fold v in as the P^T part of its own entry (the own entry was
// written with vPT = 0 by sym_fill_kernel); otherwise append.
__global__ __launch_bounds__(256) void sym_fill_ext_kernel(const float* __restrict__ svals, const int32_t* __restrict__ scols,
                                                           const int32_t* __restrict__ slen, int k, int mode,
                                                           const int64_t* __restrict__ rowptr, int32_t* __restrict__ cursor,
                                                           const int32_t* __restrict__ ext_row, const int32_t* __restrict__ ext_col,
                                                           const float* __restrict__ ext_val, int64_t n_ext,
                                                           int32_t* __restrict__ tcols, float* __restrict__ tvals) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ext) return;
    const int64_t lj = ext_row[e];
    const int32_t src = ext_col[e];
    const float v = ext_val[e];
    const int pos = bsearch_row(scols + (size_t)lj * k, slen[lj], src);
    if (pos >= 0) {
        tvals[rowptr[lj] + pos] = combine(svals[(size_t)lj * k + pos], v, mode);
    } else {
        const int64_t dst = rowptr[lj] + slen[lj] + atomicAdd(&cursor[lj], 1);
        tcols[dst] = src;
        tvals[dst] = combine(0.f, v, mode);
    }
}

// ---- 5. finalize: rank-sort every row by column (columns are unique within a row) --------------------
// rows of 65 .. 64 K entries: K entries per lane in registers, every entry broadcast once and compared with all K of the lane
// (len x (1 broadcast + K compares) instead of (len / 64)^2 passes that re-read the row; 16 % of the headline graph's rows)
template <int K>
__device__ __forceinline__ void finalize_row_regs(const int64_t b, const int len, const int lane, const int32_t* __restrict__ tcols,
                                                  const float* __restrict__ tvals, int32_t* __restrict__ cols, float* __restrict__ vals) {
    int32_t mine[K];
    float v[K];
    int rank[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = lane + 64 * k;
        mine[k] = p < len ? tcols[b + p] : INT_MAX;
        v[k] = p < len ? tvals[b + p] : 0.f;
        rank[k] = 0;
    }
#pragma unroll
    for (int kq = 0; kq < K; ++kq) {
        const int nq = len - 64 * kq < 64 ? len - 64 * kq : 64;
        for (int q = 0; q < nq; ++q) {
            const int32_t o = __builtin_amdgcn_readlane(mine[kq], q);
#pragma unroll
            for (int k = 0; k < K; ++k) rank[k] += (o < mine[k]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < len) { cols[b + rank[k]] = mine[k]; vals[b + rank[k]] = v[k]; }
}

__global__ __launch_bounds__(256) void sym_finalize_kernel(const int64_t* __restrict__ rowptr, int64_t n,
                                                           const int32_t* __restrict__ tcols, const float* __restrict__ tvals,
                                                           int32_t* __restrict__ cols, float* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    const int len = (int)(e - b);
    if (len <= 64) {  // the row sits in one register per lane; every entry is broadcast once through the scalar unit
        const bool have = lane < len;
        const int32_t mine = have ? tcols[b + lane] : INT_MAX;
        const float v = have ? tvals[b + lane] : 0.f;
        int rank = 0;
        for (int q = 0; q < len; ++q) rank += (__builtin_amdgcn_readlane(mine, q) < mine) ? 1 : 0;
        if (have) { cols[b + rank] = mine; vals[b + rank] = v; }
        return;
    }
    if (len <= 128) { finalize_row_regs<2>(b, len, lane, tcols, tvals, cols, vals); return; }
    if (len <= 256) { finalize_row_regs<4>(b, len, lane, tcols, tvals, cols, vals); return; }
    if (len <= 512) { finalize_row_regs<8>(b, len, lane, tcols, tvals, cols, vals); return; }
    if (len <= 1024) { finalize_row_regs<16>(b, len, lane, tcols, tvals, cols, vals); return; }
    for (int p0 = 0; p0 < len; p0 += 64) {
        const int p = p0 + lane;
        const bool have = p < len;
        const int32_t mine = have ? tcols[b + p] : INT_MAX;
        int rank = 0;
        for (int q0 = 0; q0 < len; q0 += 64) {
            const int32_t theirs = (q0 + lane < len) ? tcols[b + q0 + lane] : INT_MAX;
            const int nq = (len - q0 < 64) ? len - q0 : 64;
            for (int q = 0; q < nq; ++q) rank += (__builtin_amdgcn_readlane(theirs, q) < mine) ? 1 : 0;
        }
        if (have) { cols[b + rank] = mine; vals[b + rank] = tvals[b + p]; }
    }
}

// pack_to_rowwise (sparse.py:89-135): CSR -> (n, width) padded with (0, -1); indices int64.
__global__ __launch_bounds__(256) void csr_to_padded_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                            const float* __restrict__ vals, int64_t n, int64_t width,
                                                            float* __restrict__ pv, int64_t* __restrict__ pi) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * width) return;
    const int64_t i = idx / width, s = idx - i * width;
    const int64_t b = rowptr[i], len = rowptr[i + 1] - b;
    if (s < len) { pv[idx] = vals[b + s]; pi[idx] = cols[b + s]; }
    else { pv[idx] = 0.f; pi[idx] = -1; }
}

// Renumbered copy of a square CSR graph: new row j is old row perm[j], every column c becomes inv[c] (inv[perm[j]] = j);
// new_rowptr is the running sum of the permuted degrees (the caller's).  16 lanes per row.
__global__ __launch_bounds__(256) void csr_permute_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                          const float* __restrict__ vals, int64_t n, const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ inv, const int64_t* __restrict__ new_rowptr,
                                                          int32_t* __restrict__ new_cols, float* __restrict__ new_vals) {
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (j >= n) return;
    const int64_t o = perm[j];
    const int64_t src = rowptr[o], len = rowptr[o + 1] - src, dst = new_rowptr[j];
    for (int64_t e = l; e < len; e += 16) {
        new_cols[dst + e] = inv[cols[src + e]];
        new_vals[dst + e] = vals[src + e];
    }
}

}  // namespace tdr

using namespace tdr;

extern "C" {

/* Bytes of workspace for the symmetrisation of an (n,k) block (+ n_ext received edges are caller-owned). */
int64_t tdr_sym_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k <= 0) return 0;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    int64_t b = 0;
    b += n * k * 4;  // svals
    b += n * k * 4;  // scols
    b += n * 4;      // slen
    b += n * 4;      // incnt
    b += n * 4;      // cursor
    b += nb * 8;     // block sums
    b += 16;         // total (int64) + max_deg (int32)
    b += n * k * 4;  // transposed values found by the count pass
    return b + 256;
}

/*
 * Phase A: sort rows, count, scan.  Writes rowptr (n+1, int64); the host reads rowptr[n] (= nnz)
 * back to size the CSR arrays -- the one host sync of the symmetrisation, as in the reference
 * (sparse.py:119 `.max().item()`).
 */
int tdr_sym_count_ordered_f32(const float* vals, const int32_t* cols, int64_t n, int k, int64_t row_offset,
                              const int32_t* ext_row, const int32_t* ext_col, int64_t n_ext, const int32_t* order, void* ws,
                              int64_t ws_bytes, int64_t* rowptr, void* stream);
int tdr_sym_count_f32(const float* vals, const int32_t* cols, int64_t n, int k, int64_t row_offset,
                      const int32_t* ext_row, const int32_t* ext_col, int64_t n_ext, void* ws, int64_t ws_bytes,
                      int64_t* rowptr, void* stream) {
    return tdr_sym_count_ordered_f32(vals, cols, n, k, row_offset, ext_row, ext_col, n_ext, nullptr, ws, ws_bytes, rowptr, stream);
}

/* tdr_sym_count_f32 with a VISIT ORDER of the rows (optional int32 permutation of 0 .. n - 1, position -> local row): same
 * outputs; only the order in which the count pass walks the rows -- and with it the locality of its visits of rows j -- changes. */
int tdr_sym_count_ordered_f32(const float* vals, const int32_t* cols, int64_t n, int k, int64_t row_offset,
                              const int32_t* ext_row, const int32_t* ext_col, int64_t n_ext, const int32_t* order, void* ws,
                              int64_t ws_bytes, int64_t* rowptr, void* stream) {
    if (!vals || !cols || !ws || !rowptr || n <= 0 || k <= 0 || k > 256) return TDR_ERR_BAD_ARG;
    if (ws_bytes < tdr_sym_workspace_bytes(n, k)) return TDR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* svals = (float*)w; w += n * k * 4;
    int32_t* scols = (int32_t*)w; w += n * k * 4;
    int32_t* slen = (int32_t*)w; w += n * 4;
    int32_t* incnt = (int32_t*)w; w += n * 4;
    int32_t* cursor = (int32_t*)w; w += n * 4;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    w = (char*)(((uintptr_t)w + 7) & ~(uintptr_t)7);
    int64_t* block_sums = (int64_t*)w; w += nb * 8;
    int64_t* total = (int64_t*)w; w += 8;
    int32_t* max_deg = (int32_t*)w; w += 8;
    uint32_t* vt = (uint32_t*)w;
    hipError_t e = hipMemsetAsync(incnt, 0, (size_t)n * 8, st);  // incnt + cursor
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(total, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sym_rowsort_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, vals, cols, n, k, svals, scols, slen);
    hipLaunchKernelGGL(sym_count_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, st, svals, scols, slen, n, k, row_offset, incnt, vt, order);
    if (n_ext > 0) {
        if (!ext_row || !ext_col) return TDR_ERR_BAD_ARG;
        hipLaunchKernelGGL(sym_count_ext_kernel, dim3((unsigned)((n_ext + 255) / 256)), dim3(256), 0, st, scols, slen, k, ext_row, ext_col, n_ext, incnt);
    }
    hipLaunchKernelGGL(scan_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, slen, incnt, n, block_sums, max_deg);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, st, block_sums, nb, total);
    hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nb), dim3(256), 0, st, slen, incnt, n, block_sums, rowptr);
    hipLaunchKernelGGL(scan_tail_kernel, dim3(1), dim3(64), 0, st, total, n, rowptr);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

/* Phase B: fill + finalize into caller-allocated CSR arrays (cols int32, vals fp32, nnz entries) using
 * two nnz-sized temporaries. mode 0 = sum_minus_prod, 1 = sum. */
int tdr_sym_fill_ordered_f32(int64_t n, int k, int64_t row_offset, int mode, const int32_t* ext_row, const int32_t* ext_col,
                             const float* ext_val, int64_t n_ext, const int32_t* order, void* ws, const int64_t* rowptr, int32_t* tcols,
                             float* tvals, int32_t* cols, float* vals, void* stream);
int tdr_sym_fill_f32(int64_t n, int k, int64_t row_offset, int mode, const int32_t* ext_row, const int32_t* ext_col,
                     const float* ext_val, int64_t n_ext, void* ws, const int64_t* rowptr, int32_t* tcols,
                     float* tvals, int32_t* cols, float* vals, void* stream) {
    return tdr_sym_fill_ordered_f32(n, k, row_offset, mode, ext_row, ext_col, ext_val, n_ext, nullptr, ws, rowptr, tcols, tvals, cols, vals, stream);
}

/* tdr_sym_fill_f32 with the visit order of tdr_sym_count_ordered_f32 (optional): same CSR (the column sort that ends the
 * phase makes the order in which transposed entries were appended irrelevant). */
int tdr_sym_fill_ordered_f32(int64_t n, int k, int64_t row_offset, int mode, const int32_t* ext_row, const int32_t* ext_col,
                             const float* ext_val, int64_t n_ext, const int32_t* order, void* ws, const int64_t* rowptr, int32_t* tcols,
                             float* tvals, int32_t* cols, float* vals, void* stream) {
    if (!ws || !rowptr || !tcols || !tvals || !cols || !vals || n <= 0 || k <= 0) return TDR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* svals = (float*)w; w += n * k * 4;
    int32_t* scols = (int32_t*)w; w += n * k * 4;
    int32_t* slen = (int32_t*)w; w += n * 4;
    w += n * 4;  // incnt
    int32_t* cursor = (int32_t*)w; w += n * 4;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    w = (char*)(((uintptr_t)w + 7) & ~(uintptr_t)7);
    w += nb * 8 + 16;  // block sums, total + max_deg
    const uint32_t* vt = (const uint32_t*)w;
    hipLaunchKernelGGL(sym_fill_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, st, svals, scols, slen, n, k, row_offset, mode, rowptr, cursor, vt, tcols, tvals, order);
    if (n_ext > 0) {
        if (!ext_row || !ext_col || !ext_val) return TDR_ERR_BAD_ARG;
        hipLaunchKernelGGL(sym_fill_ext_kernel, dim3((unsigned)((n_ext + 255) / 256)), dim3(256), 0, st, svals, scols, slen, k, mode, rowptr, cursor, ext_row, ext_col, ext_val, n_ext, tcols, tvals);
    }
    hipLaunchKernelGGL(sym_finalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, rowptr, n, tcols, tvals, cols, vals);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

int tdr_csr_to_padded_f32(const int64_t* rowptr, const int32_t* cols, const float* vals, int64_t n, int64_t width,
                          float* pv, int64_t* pi, void* stream) {
    if (!rowptr || !pv || !pi || n <= 0 || width < 0) return TDR_ERR_BAD_ARG;
    if (width == 0) return TDR_OK;
    hipLaunchKernelGGL(csr_to_padded_kernel, dim3((unsigned)((n * width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, cols, vals, n, width, pv, pi);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}
int tdr_csr_permute_f32(const int64_t* rowptr, const int32_t* cols, const float* vals, int64_t n, const int32_t* perm,
                        const int32_t* inv, const int64_t* new_rowptr, int32_t* new_cols, float* new_vals, void* stream) {
    if (!rowptr || !cols || !vals || !perm || !inv || !new_rowptr || !new_cols || !new_vals || n <= 0) return TDR_ERR_BAD_ARG;
    hipLaunchKernelGGL(csr_permute_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, cols, vals,
                       n, perm, inv, new_rowptr, new_cols, new_vals);
    TDR_CHECK_LAUNCH();
    return TDR_OK;
}

}  // extern "C"
