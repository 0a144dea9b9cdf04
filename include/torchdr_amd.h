/*
 * torchdr_amd -- C ABI of the MI355X (gfx950) neighbor-embedding hot path.
 *
 * The reference (TorchDR, /root/reference/torchdr) is 100 % Python: its "FFI" for this path is the
 * sequence of ATen / MKL / Faiss calls made by the Python plugin classes.  Each entry point below
 * replaces one such sequence; the reference interface it stands in for is cited as file:line.
 *
 * Conventions
 *   - plain `extern "C"`, raw DEVICE pointers + sizes, `void* stream` = hipStream_t (NULL = default);
 *   - return 0 on success, negative = argument / capability error (TDR_ERR_*), positive = hipError_t;
 *   - asynchronous on `stream`; never allocates caller-visible memory (outputs and workspaces are
 *     passed in, with `*_workspace_bytes` / `*_floats` size queries);
 *   - all matrices row-major fp32; kNN indices int32 (utils/utils.py:216), CSR row pointers int64.
 */
#ifndef TORCHDR_AMD_H
#define TORCHDR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDR_OK 0
#define TDR_ERR_BAD_ARG (-1)
#define TDR_ERR_UNSUPPORTED (-2)
#define TDR_ERR_WORKSPACE (-3)

#define TDR_METRIC_SQEUCLIDEAN 0
#define TDR_METRIC_EUCLIDEAN 1
#define TDR_METRIC_ANGULAR 2

/* ---- K1: pairwise distances / exact kNN ------------------------------------------------------
 * replaces distance/torch.py:21-125 (pairwise_distances_torch), utils/utils.py:173-216 (kmin),
 * distance/faiss.py:224-403 (IndexFlatL2 search) and the chunked form distance/base.py:183-206. */

/* floats needed by the packed (MFMA tile image) copy of an n x d block; 0 if d is unsupported (> 256). */
int64_t tdr_packed_floats(int64_t n, int d);

/* X (n x d, row stride ldx) -> packed tile images (+ squared norms, ATen summation order,
 * distance/torch.py:82).  norms_out (n) may be NULL. */
int tdr_pack_rows_f32(const float* X, int64_t n, int d, int64_t ldx, float* packed, float* norms_out, void* stream);

int64_t tdr_knn_workspace_bytes(int64_t nq, int64_t n_db, int k);
int tdr_knn_max_k(int d);

/* k smallest distances of each of nq packed queries against n_db packed database rows.
 * q_offset: global row id of query 0 (for exclude_self: skip database row q_offset + i,
 * distance/torch.py:111-116).  out_d (nq,k) ascending, out_i (nq,k) int32, ties by index. */
int tdr_knn_packed_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d, int k,
                       int metric, int exclude_self, float* out_d, int32_t* out_i, void* ws, int64_t ws_bytes,
                       void* stream);
/* D > 256 (distance/torch.py:82-120 at e.g. 784 features): the same tile images with the feature dimension padded to a
 * multiple of 32, and a K-chunked scan (a wavefront accumulates 4 database tiles per pass over its queries' features)
 * with the running top-k fused; same contract as tdr_knn_packed_f32, workspace from tdr_knn_workspace_bytes.  Every
 * distance is one k-ordered fp32 fma chain over the row (MKL splits contractions beyond K ~ 380, so parity with the CPU
 * reference is to fp32 rounding there). */
int64_t tdr_packed_floats_wide(int64_t n, int d);
int tdr_pack_rows_wide_f32(const float* X, int64_t n, int d, int64_t ldx, float* packed, float* norms_out, void* stream);
int tdr_knn_wide_max_k(void);
int tdr_knn_wide_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d, int k, int metric,
                     int exclude_self, float* out_d, int32_t* out_i, void* ws, int64_t ws_bytes, void* stream);
int tdr_dense_dist_wide_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d, int metric,
                            int exclude_self, float diag_add, float* out, int64_t ldo, void* stream);

/* dense nq x n_db matrix (k=None path, distance/torch.py:91-116); diag_add (1e12) on C[i][q_offset+i]. */
int tdr_dense_dist_packed_f32(const float* qp, int64_t nq, int64_t q_offset, const float* yp, int64_t n_db, int d,
                              int metric, int exclude_self, float diag_add, float* out, int64_t ldo, void* stream);

/* gathered distances out[i][c] = ||X[q_i] - Y[keys[i][c]]||^2 (distance/base.py:384-385; take_sqrt = 1: its square
 * root, :386-387; take_sqrt = 2: sum_c |x - y|, manhattan, :388-389; 3: -sum_c x y, angular, :390-391; 4: sqhyperbolic, :392-398); negative indices wrap like PyTorch indexing. */
int tdr_indexed_sqdist_f32(const float* X, int64_t nx, int d, const float* Y, int64_t ny, const int64_t* q,
                           int64_t nq, int nk, int take_sqrt, const int64_t* keys, float* out, void* stream);

/* General feature dimension (D > 256): the contraction is a plain library GEMM per (query chunk x database chunk)
 * block, issued by the host; these three kernels are the rest of distance/torch.py:91-120 + utils/utils.py:215
 * (norm expansion, self exclusion, running top-k with the canonical (distance, index) order). */
int tdr_topk_max_k(void);
int tdr_topk_init(uint64_t* run_keys, int64_t nq, int k, void* stream);
int tdr_topk_merge_f32(const float* G, int64_t ldg, int64_t nq, int64_t nd, const float* xn, const float* yn,
                       int64_t q_global0, int64_t d_global0, int k, int metric, int exclude_self, uint64_t* run_keys,
                       void* stream);
int tdr_topk_emit_f32(const uint64_t* run_keys, int64_t nq, int k, int metric, float* out_d, int32_t* out_i, void* stream);
/* metric "sqhyperbolic" (distance/torch.py:101-107): kNN = tdr_topk_merge_f32 with metric 4; dense form = this in-place
 * epilogue on the Gram block G = X Y^T */
int tdr_hyperbolic_from_gram_f32(float* G, int64_t ld, int64_t nq, int64_t nd, const float* xn, const float* yn,
                                 void* stream);
/* the same fold over per-query candidate lists: E (nq, nc) distances, cand (nq, nc) database indices (row stride ld,
 * negative = skip); ranks exactly re-evaluated candidates by (distance, index) */
int tdr_topk_merge_cand_f32(const float* E, const int32_t* cand, int64_t ld, int64_t nq, int64_t nc, int k,
                            uint64_t* run_keys, void* stream);

/* Manhattan metric (distance/torch.py:96-98, distance/base.py:368): out[i][j] = sum_c |X[i][c] - Y[j][c]| for an
 * (nq x nd) block with row strides ldx / ldy / ldo.  kNN = this block + tdr_topk_merge_f32 with metric 3. */
int tdr_l1_block_f32(const float* X, int64_t ldx, int64_t nq, const float* Y, int64_t ldy, int64_t nd, int d, float* out,
                     int64_t ldo, void* stream);
/* The same distance evaluated in the reference's own summation order (ATen's vectorised inner sum: 32 interleaved
 * running sums, cascade every 16 rounds -- the float the CPU backend returns, distance/torch.py:98), one thread per
 * (query, candidate): query i = row q_rows[i] of X (or row i; global id = q_global0 + that row), database row
 * cand[i][c] (negative -> +inf) or j0 + c when cand is NULL; exclude_self: own row -> +inf.  d < 8192. */
int tdr_l1_exact_f32(const float* X, int64_t ldx, const int64_t* q_rows, int64_t nq, int64_t q_global0, const float* Y,
                     int64_t ldy, const int32_t* cand, int64_t ldc, int64_t nc, int64_t j0, int d, int exclude_self,
                     float* out, int64_t ldo, void* stream);

/* kNN consumers: eval/neighborhood_preservation.py:175-181 (per-row overlap of two neighbour lists) */
int tdr_knn_overlap_i32(const int32_t* a, const int32_t* b, int64_t n, int K, float* out, void* stream);

/* ---- K1s: two-stage exact kNN (fp16-split screening on the f16 matrix pipe + exact fp32 rescoring) ----
 * Same results, bit for bit, as tdr_knn_packed_f32 (hence as distance/torch.py:82-120 + utils/utils.py:215):
 * the screening pass only decides WHICH pairs get their reference-arithmetic distance evaluated, with a
 * worst-case error band (torchdr_amd/csrc/tdr_knn_screen.hip header).  sqeuclidean / euclidean, D <= 256. */
int tdr_knn_screen_supported(int d, int k);
int64_t tdr_packed16_floats(int64_t n, int d);
/* meta: 2 x uint32 on the device, zeroed by the caller; accumulates max |X| (and max norms[i] when norms != NULL)
 * over every block that will share one fp16 scale (queries + database). */
int tdr_screen_meta_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, uint32_t* meta, void* stream);
int tdr_pack16_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, const uint32_t* meta,
                   float* packed16, void* stream);
/* as tdr_pack16_f32 for a re-ordered, padded image: image row r <- source row row_map[r] (-1 = padding row) */
int tdr_pack16_mapped_f32(const float* X, int64_t n, int d, int64_t ldx, const float* norms, const uint32_t* meta,
                          const int32_t* row_map, float* packed16, void* stream);
int64_t tdr_knn_screen_workspace_bytes(int64_t nq, int64_t n_db, int d, int k, int tier);
/* flags[i] = 1: the screening list of query i overflowed, its output rows are invalid and must be recomputed with
 * tdr_knn_packed_f32; *n_flagged (device int32, caller-zeroed) counts such queries.  tier 0: one-term screening
 * (h.h' only: a third of the matrix work, band ~2^-10 |x||y|; unsupported when k + 16 list slots do not fit);
 * tier 1: three-term screening, k + ~17..24 spare slots; tier 2: three terms, up to k + 72 spare slots.
 * predict_unsplit = 1: pilot slice -- flag what an un-sliced launch of the same search would flag. */
int tdr_knn_screen_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                       const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                       int metric, int exclude_self, int tier, int predict_unsplit, const uint32_t* meta, float* out_d,
                       int32_t* out_i, int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream);

/* Coarse cluster index of the pruned self search, built with the package's own kernels (csrc/tdr_cluster.hip; the
 * search result never depends on it, only the number of skipped tiles does): stratified sample, farthest-point seeding
 * of C clusters by ONE workgroup on the sample's exact distance matrix (tdr_dense_dist_packed_f32), Lloyd updates
 * (assignments come from tdr_knn_packed_f32 with k = 1), radii / centre distances by direct differences with outward
 * rounding, padded cluster-sorted layout. */
int tdr_cluster_sample_i32(int64_t n, int S, uint32_t seed, int32_t* sample_idx, void* stream);
int tdr_cluster_maxmin_capacity(void);
/* measurement / test switch of the farthest-point seeding: 1 (default since round 6) = every dependent step finds winner AND
 * runner-up and reads both rows of the distance matrix; the runner-up is taken as the following seed when it is at least as far from
 * the winner as from the seeds before (it is then exactly the next greedy pick); 0 = one seed per step; same seeds; returns the
 * previous value */
int tdr_cluster_maxmin_mode(int two_per_step);
int tdr_cluster_maxmin_f32(const float* D2, int64_t ld, int S, int C, int32_t* seeds, void* stream);
/* the same with the number of seeds read off the data: up to c_max seeds, ended at the first step t >= c_min at which the
 * max-min squared distance falls below `drop` x the previous one (every well-separated group holds a seed); no such step:
 * c_min seeds.  *n_seeds: device int32; seeds: c_max entries */
int tdr_cluster_maxmin_adaptive_f32(const float* D2, int64_t ld, int S, int c_min, int c_max, float drop, int32_t* seeds,
                                    int32_t* n_seeds, void* stream);
int tdr_gather_rows_f32(const float* X, int64_t ldx, int d, const int32_t* idx, const int32_t* idx2, int64_t m, float* out,
                        void* stream);
int tdr_cluster_update_f32(const float* Xs, int64_t S, int d, const int32_t* labels, int C, float* cent, void* ws, void* stream);
/* the layout is a stable counting sort (members of a cluster by ascending row): every rank of a row-sharded fit builds the
 * SAME index by itself.  perm / inv / ppos (nullable together, n int32 each): the cluster-sorted order without padding
 * (position -> row, row -> position, position -> position in the padded layout). */
int64_t tdr_cluster_tables_workspace_bytes(int64_t n, int C);
int tdr_cluster_tables_f32(const float* X, int64_t n, int d, int64_t ldx, const int32_t* labels, const float* cent, int C,
                           float* radius, int32_t* tile_begin, int32_t* tiles, int32_t* tile_cluster, int32_t* row_map,
                           int64_t* n_img, float* dist, int32_t* order, int32_t* perm, int32_t* inv, int32_t* ppos, void* ws,
                           int64_t ws_bytes, void* stream);
/* Self search with cluster-bound pruning: the points are sorted by a coarse clustering and padded so that clusters
 * start on tile boundaries (row_map); a workgroup visits clusters by increasing centre distance and skips every
 * cluster whose ball cannot reach its queries' current thresholds.  Same results as tdr_knn_screen_f32 (and hence
 * as tdr_knn_packed_f32) whatever the clustering; outputs are indexed by source row and hold source indices.
 * [q_pos_begin, q_pos_end): range of the sorted order answered by this launch (0, 0 = everything). */
int tdr_knn_screen_clustered_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k,
                                 int metric, int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map,
                                 int n_clusters, const int32_t* tile_cluster, const int32_t* clus_tile_begin,
                                 const float* clus_radius, const float* clus_dist, const int32_t* clus_order,
                                 int64_t q_pos_begin, int64_t q_pos_end, float* out_d, int32_t* out_i, int32_t* flags,
                                 int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream);
/* workspace bytes of the cluster-pruned searches (this one, _tb_f32, tdr_knn_ivf_f32) over n_img image rows; never less than
 * tdr_knn_screen_workspace_bytes(n_img, n_img, d, k, tier); 0 = unsupported */
int64_t tdr_knn_screen_clustered_workspace_bytes(int64_t n_img, int d, int k, int tier);
/* measurement / test switch of the exact cluster-pruned searches: 1 (default since round 6) = per-query UNSORTED candidate buffers
 * in LDS (up to 126 entries), every lane appending its own survivors, compacted -- bisection for the k-th smallest screening value,
 * entries beyond it + the error band dropped -- only when full and at the end of a cluster; 0 = the sorted lists of rounds 2-5.
 * Same neighbours either way; returns the previous value */
int tdr_knn_screen_clustered_lists(int lazy);
/* tdr_knn_screen_clustered_f32 with a per-tile table: tile_cdist (n_img / 32, n_clusters) = for every 32-row tile of the sorted
 * order a LOWER bound of the distance from any of its rows to every cluster centre (tdr_cluster_tile_cdist_f32).  A cluster is
 * skipped when |x - c| - R_c of the workgroup's own query rows already exceeds their thresholds: sharper than the ball-to-ball
 * bound, clusters whose balls overlap are still told apart.  Same results bit for bit. */
int tdr_knn_screen_clustered_tb_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k,
                                    int metric, int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map,
                                    int n_clusters, const int32_t* tile_cluster, const int32_t* clus_tile_begin,
                                    const float* clus_radius, const float* clus_dist, const int32_t* clus_order,
                                    const float* tile_cdist, int64_t q_pos_begin, int64_t q_pos_end, float* out_d, int32_t* out_i,
                                    int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream);
/* predicted share of the database tiles a pruned scan still visits at threshold tau: out (2 x uint64, caller-zeroed) = {sum over
 * cluster pairs (w, c) with max(0, dist[w, c] - radius[w] - radius[c])^2 <= tau of tiles[w] tiles[c], sum of tiles}; share = out[0] /
 * out[1]^2 (ClusterIndex.scan_fraction; exact integers, the same on every rank) */
int tdr_cluster_scan_fraction_f32(const float* dist, const float* radius, const int32_t* tiles, int C, float tau, void* out, void* stream);
/* rows of that table from a block of exact squared distances d2 (rows, C) of consecutive rows of the padded sorted order to the
 * C centres: out (rows / 32, C) = sqrt(max(0, min over the tile's valid rows of d2 - (d + 16) 2^-23 (|x|^2 + |c|^2))) rounded
 * down; row_map (rows): source row or -1; xn (rows), cn (C): squared norms */
int tdr_cluster_tile_cdist_f32(const float* d2, int64_t ld, int64_t rows, int C, int d, const int32_t* row_map, const float* xn,
                               const float* cn, float* out, void* stream);
/* The UNPRUNED two-stage search as a threshold scan (round 5, csrc/tdr_knn_flat.hip; replaces the list-keeping kernel where no
 * tile can be skipped -- distance/torch.py:82-122 on structureless data, benchmarks/faiss/run_benchmark.py:143-146):
 * seed (tdr_knn_flat_seed_f32: every screening value of 256 rows) -> select -> per-query threshold tau = a_(k) + 2E -> passes of
 * tdr_knn_flat_scan_f32 over ranges of tile positions growing geometrically to the whole database (by 4 for k <= 30, by less for
 * larger k: a pass appends ~ (r - 1) k entries to a query's 256-entry region; every candidate with screening value <= tau is
 * appended; no lists in LDS, two query tiles per wavefront, two database tiles per barrier) with a tdr_knn_flat_select_f32 after
 * each (list + appended -> the L smallest, new tau) -> the rescoring kernel.  Same operands, outputs and flag contract as
 * tdr_knn_screen_f32; results are bit-identical.  terms: 1 (h.h'), 2 (h.h' + h.l') or 3; L: list length per query (k <= L <= 128).
 * The workspace query returns 0 when the threshold scan does not serve the search (D > 256, or D > 128 with more than one term;
 * fewer than 4096 database tiles; unsupported terms / L). */
int tdr_knn_flat_supported(int d);
int64_t tdr_knn_screen_flat_workspace_bytes(int64_t nq, int64_t n_db, int d, int k, int terms, int L);
/* its pass plan (host arithmetic only): bounds[0 .. n) tile positions -- seed [0, bounds[0]), pass i [bounds[i], bounds[i + 1]) --,
 * *stride = stride of the visiting order; returns n, 0 when the search is not served */
int tdr_knn_screen_flat_plan(int64_t nq, int64_t n_db, int d, int k, int terms, int L, int32_t* bounds, int max_bounds, int32_t* stride);
int tdr_knn_screen_flat_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                            const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                            int metric, int exclude_self, int terms, int L, const uint32_t* meta, float* out_d, int32_t* out_i,
                            int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream);
/* its stages, exposed for tests and measurement.  Tiles are visited in the order position j -> tile (j * tile_stride) mod n_tiles
 * (tile_stride coprime to the tile count; 1 = natural order): seed and passes take ranges of POSITIONS, so each sees rows from all
 * over the database.  seed: buf[q * cap + 32 j + r] = key of row r of the tile at position j < seed_tiles (sentinel for the query
 * itself / padding).  scan: positions [tile_begin, tile_end); buf (nq, cap) keys (screening value bits << 32 | database row),
 * cnt (nq) candidates met by THIS launch (> cap: the surplus was dropped); terms 1, 2 (h.h' + h.l') or 3.
 * select: list (nq, L) in/out ascending, sentinel 0xFF800000FFFFFFFF; extra = the scan's buf with extra_cnt = cnt (n_sets 1,
 * stride cap), or n_sets x (nq, stride) keys with extra_cnt NULL (every entry counts, sentinels allowed: the seed); tau (nq) out = min(a_(k) + 2E, a_(L) when full); lost (nq) set to 1 where cnt > stride. */
int tdr_knn_flat_seed_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int seed_tiles, int tile_stride, const uint32_t* meta, uint64_t* buf, int cap, void* stream);
int tdr_knn_flat_scan_f32(const float* q16, int64_t nq, int64_t q_offset, const float* y16, int64_t n_db, int d, int terms,
                          int exclude_self, int tile_begin, int tile_end, int tile_stride, const uint32_t* meta, const float* tau,
                          uint64_t* buf, int32_t* cnt, int cap, void* stream);
int tdr_knn_flat_select_f32(uint64_t* list, int have_list, const uint64_t* extra, const int32_t* extra_cnt, int n_sets,
                            int stride, const float* norms_q, const uint32_t* meta, int64_t nq, int d, int k, int L, int terms,
                            float* tau, int32_t* lost, void* stream);
/* cluster index, step 3 (round 5): labels[i] = centre nearest to point i by the one-term screening value of the fp16-split images
 * (same meta); approximate (2^-10 relative) and deterministic -- the clustering only decides how much a pruned search can skip */
int tdr_cluster_assign16_supported(int d);
int tdr_cluster_assign16_f32(const float* x16, int64_t n, const float* c16, int n_centres, int d, const uint32_t* meta,
                             int32_t* labels, void* stream);
/* tdr_knn_screen_f32 in pilot mode (predict_unsplit = 1) with the prediction made for lists of pred_L entries; pred_terms = 0: the
 * band of the pilot's own tier, 2 (tier >= 1): the band of the two-term split h.h' + h.l' counted on the three-term pilot's values */
int tdr_knn_screen_pilot_f32(const float* q16, const float* Xq, int64_t ldq, const float* norms_q, int64_t nq, int64_t q_offset,
                             const float* y16, const float* Y, int64_t ldy, const float* norms_y, int64_t n_db, int d, int k,
                             int metric, int exclude_self, int tier, int pred_L, int pred_terms, const uint32_t* meta, float* out_d,
                             int32_t* out_i, int32_t* flags, int32_t* n_flagged, void* ws, int64_t ws_bytes, void* stream);
/* Approximate IVF-style self search on the same cluster index (distance/faiss.py:331-349: nlist = n_clusters, nprobe):
 * a workgroup scans its own clusters and then the nearest ones, nprobe scans in all; candidates are rescored exactly.
 * out_d / out_i must be pre-filled by the caller (+inf / -1): rows with fewer than k candidates keep that tail. */
int tdr_knn_ivf_f32(const float* x16, const float* X, int64_t ldx, const float* norms, int64_t n_img, int d, int k, int metric,
                    int exclude_self, int tier, const uint32_t* meta, const int32_t* row_map, int n_clusters,
                    const int32_t* tile_cluster, const int32_t* clus_tile_begin, const float* clus_radius, const float* clus_dist,
                    const int32_t* clus_order, int nprobe, float* out_d, int32_t* out_i, int32_t* flags, int32_t* n_flagged,
                    void* ws, int64_t ws_bytes, void* stream);

/* ---- K2 / K3: per-row root searches --------------------------------------------------------------
 * replace utils/root_search.py:17-77,147-198 driven by affinity/knn_normalized.py:445-465 (UMAP) and
 * affinity/entropic.py:272-310 (+ bounds :96-113). */
int tdr_umap_search_f32(const float* C, int64_t n, int k, float target, int max_iter, float tol, float* rho,
                        float* eps, float* P, void* stream);

int tdr_entropic_search_f32(const float* C, int64_t n, int k, float target, float log_n, int max_iter, float tol,
                            int use_bounds, float tN_logratio, float tN_m1, float log_ratio, float beta_u_num,
                            float* eps, float* log_norm, float* log_P, void* stream);

/* ---- K4: sparse symmetrisation -------------------------------------------------------------------
 * replaces utils/sparse.py:7-206 (symmetrize_sparse) and, with ext edges, :209-342
 * (distributed_symmetrize_sparse).  Two phases around the one host read of nnz = rowptr[n]. */
int64_t tdr_sym_workspace_bytes(int64_t n, int k);
int tdr_sym_count_f32(const float* vals, const int32_t* cols, int64_t n, int k, int64_t row_offset,
                      const int32_t* ext_row, const int32_t* ext_col, int64_t n_ext, void* ws, int64_t ws_bytes,
                      int64_t* rowptr, void* stream);
/* tdr_sym_count_f32 with a visit order of the rows (optional int32 permutation of 0 .. n - 1, position -> local row, e.g. the
 * cluster-sorted order of the kNN search that produced the block): same outputs, local visits of the transposed rows;
 * tdr_sym_fill_ordered_f32 takes the same order for the second phase. */
int tdr_sym_count_ordered_f32(const float* vals, const int32_t* cols, int64_t n, int k, int64_t row_offset,
                              const int32_t* ext_row, const int32_t* ext_col, int64_t n_ext, const int32_t* order, void* ws,
                              int64_t ws_bytes, int64_t* rowptr, void* stream);
int tdr_sym_fill_f32(int64_t n, int k, int64_t row_offset, int mode, const int32_t* ext_row, const int32_t* ext_col,
                     const float* ext_val, int64_t n_ext, void* ws, const int64_t* rowptr, int32_t* tcols,
                     float* tvals, int32_t* cols, float* vals, void* stream);
int tdr_sym_fill_ordered_f32(int64_t n, int k, int64_t row_offset, int mode, const int32_t* ext_row, const int32_t* ext_col,
                             const float* ext_val, int64_t n_ext, const int32_t* order, void* ws, const int64_t* rowptr, int32_t* tcols,
                             float* tvals, int32_t* cols, float* vals, void* stream);
/* CSR -> padded (n, width) with (0, -1) fill (pack_to_rowwise, utils/sparse.py:89-135). */
int tdr_csr_to_padded_f32(const int64_t* rowptr, const int32_t* cols, const float* vals, int64_t n, int64_t width,
                          float* pv, int64_t* pi, void* stream);
/* Renumbered copy of a square CSR graph (no reference counterpart: the embedding loop of UMAP numbers the points in the
 * cluster-sorted order of the kNN stage so that a row's neighbours share cache lines): new row j = old row perm[j], column
 * c -> inv[c] with inv[perm[j]] = j; new_rowptr (n + 1) = running sum of the permuted degrees. */
int tdr_csr_permute_f32(const int64_t* rowptr, const int32_t* cols, const float* vals, int64_t n, const int32_t* perm,
                        const int32_t* inv, const int64_t* new_rowptr, int32_t* new_cols, float* new_vals, void* stream);

/* ---- multi-GPU context: an RCCL communicator behind the C ABI (csrc/tdr_ctx.hip) ----------------------------------
 * One process per GPU.  The collectives are enqueued on the caller's stream (no host synchronisation; capturable into the
 * HIP graphs of tdr_umap_loop_*).  librccl is dlopen'ed from rccl_path ("" = "librccl.so"), normally the copy PyTorch
 * loaded.  Replaces affinity_matcher.py:395-413 (zero-padded all-reduce of the stepped rows -> in-place all-gather) and
 * :425 (gradient all-reduce).  RCCL errors are returned as 1000 + ncclResult_t. */
int tdr_ctx_unique_id(const char* rccl_path, void* out128);
int tdr_ctx_create(void** ctx, int rank, int world, const void* unique_id128, const char* rccl_path, int64_t n_total);
int tdr_ctx_set_rows(void* ctx, int64_t n_total);
int tdr_ctx_allgather_rows(void* ctx, float* Z, int nc, void* stream);
int tdr_ctx_allreduce_f32(void* ctx, float* buf, int64_t count, void* stream);
int tdr_ctx_destroy(void* ctx);

/* ---- float64 twins of the affinity side (csrc/tdr_f64.hip): the reference computes in its input's dtype ---------------
 * K1 on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64), K2 / K3 root searches, the gathered distances and the values of the
 * symmetrised graph on the pattern built by the float32 pipeline.  metric: 0 sqeuclidean, 1 euclidean, 2 angular.
 * tdr_knn_f64: k > 0 -> (nq, k) smallest distances ascending by (distance, index) + indices; k == 0 -> dense (nq, n_db)
 * matrix into out_d (row stride ldo).  ws: (nq + n_db) doubles.  tdr_knn_f64_lds_bytes == 0: shape unsupported. */
int64_t tdr_knn_f64_lds_bytes(int d, int k);
int tdr_knn_f64(const double* Xq, int64_t nq, int64_t ldq, int64_t q_global0, const double* Y, int64_t n_db, int64_t ldy, int d, int k,
                int metric, int exclude_self, double diag_add, double* out_d, int32_t* out_i, int64_t ldo, double* ws, void* stream);
int tdr_umap_search_f64(const double* C, int64_t n, int k, double target, int max_iter, double tol, double* rho, double* eps,
                        double* P, void* stream);
int tdr_entropic_search_f64(const double* C, int64_t n, int k, double target, double log_n, int max_iter, double tol, int use_bounds,
                            double tN, double perplexity, double p1, double* eps, double* lognorm, double* logP, void* stream);
int tdr_indexed_sqdist_f64(const double* X, int64_t nx, int d, const double* Y, int64_t ny, const int64_t* q, int64_t nq, int nk, int mode,
                           const int64_t* keys, double* out, void* stream);
int tdr_sym_values_f64(const int64_t* rowptr, const int32_t* cols, int64_t n, const int32_t* nn, const double* P, int k,
                       int64_t row_offset, int mode, double* vals, void* stream);
/* the same for one rank's rows of a row-sharded graph (utils/sparse.py:209-342): transposed entries from other ranks as a CSR
 * over the local rows (ext_rowptr (n + 1), ext_col = global source row, ext_val) */
int tdr_sym_values_ext_f64(const int64_t* rowptr, const int32_t* cols, int64_t n, const int32_t* nn, const double* P, int k,
                           int64_t row_offset, int mode, const int64_t* ext_rowptr, const int32_t* ext_col, const double* ext_val,
                           double* vals, void* stream);

/* ---- K0: the steps either side of the path inside fit_transform (csrc/tdr_prep.hip) ---------------------------- */
/* utils/validation.py:308 (torch.isfinite(X).all()): *count (device uint64, caller-zeroed) += number of inf / nan entries */
int tdr_nonfinite_count_f32(const float* X, int64_t n, int d, int64_t ldx, void* count, void* stream);
/* base.py:132-148 (torch.unique(X, dim=0, return_inverse=True)): rep (n) int32 = smallest index of a row equal to row i;
 * counters (2 x uint32, device) = {rows duplicating an earlier row, rows whose 64-bit hash collided with a different
 * row (rep is then not reliable: use an exact method)}.  ws: tdr_dedup_workspace_bytes(n). */
int64_t tdr_dedup_workspace_bytes(int64_t n);
int tdr_dedup_rows_f32(const float* X, int64_t n, int d, int64_t ldx, int32_t* rep, void* counters, void* ws, int64_t ws_bytes,
                       void* stream);
/* spectral_embedding/pca.py:151-184: column means (d fp32) and the Gram matrix of the centred block (d x d fp64) on the
 * fp32 matrix pipe (d <= 256), and the projection E (n, nc <= 4) = (X - mean) V. */
int64_t tdr_pca_gram_workspace_floats(int64_t n, int d);
int tdr_pca_gram_f32(const float* X, int64_t n, int d, int64_t ldx, float* mean, double* G, float* ws, int64_t ws_floats,
                     void* stream);
int tdr_pca_project_f32(const float* X, int64_t n, int d, int64_t ldx, const float* mean, const float* V, int nc, float* E,
                        void* stream);
/* the d x d eigenproblem between the two (spectral_embedding/pca.py:169-178 uses a thin SVD of the centred block): one-sided
 * Jacobi in one workgroup, no host read -- evals (d) descending, evecs (d, d) row-major with column r the eigenvector of
 * evals[r]; G symmetric positive semi-definite, d <= 256; ws = 2 d^2 doubles. */
int tdr_eigh_jacobi_f64(const double* G, int d, double* evals, double* evecs, double* ws, void* stream);
/* the same eigenproblem when only the nc <= 4 LEADING pairs are wanted (n_components of the PCA initialisation): Householder
 * tridiagonalisation + Sturm multisection + inverse iteration in one workgroup, no host read, same bits on every rank -- evals
 * (nc) descending, evecs (d, nc) row-major with unit columns; G symmetric, d <= 256, nc <= min(d, 4); ws = d^2 doubles. */
int tdr_eigh_top_f64(const double* G, int d, int nc, double* evals, double* evecs, double* ws, void* stream);

/* ---- K5 / K6 / K9: embedding loop ----------------------------------------------------------------- */
/* neighbor_embedding/umap.py:215-234 */
int tdr_umap_prepare_f32(const float* vals, int64_t nnz, int max_iter, float* eps_per, float* next, void* scratch,
                         void* stream);
/* neighbor_embedding/umap.py:236-292 + neighbor_embedding/base.py:617-649 (negatives) */
int64_t tdr_umap_grad_workspace_bytes(int64_t n_total, int64_t n_rows, int nc);
int tdr_umap_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int64_t* rowptr,
                      const int32_t* cols, const float* eps_per, float* next, float a, float b, int n_iter,
                      int neg_rate, int n_negatives, const int64_t* neg_inj, uint64_t seed, float exag, float rep,
                      float eps, float* grad, int neg_slices, void* ws, int64_t ws_bytes, void* stream);
/* test hook: the negatives the dense L2-sliced negative passes (and, for n_slices in {1, 2, 4, 8}, the scheduled passes
 * below) draw for rows with nuse[r] negatives -- distribution checks and oracle parity of the in-kernel sampler */
int tdr_umap_debug_negatives(uint64_t seed, int n_iter, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nuse,
                             int n_slices, int width, int64_t* out, void* stream);
/* The same gradient (umap.py:236-292) on per-iteration firing lists ("scheduled epochs", csrc/tdr_umap_sched.hip): which
 * edge fires at which iteration (umap.py:243-247: act = next <= n_iter + 1; next[act] += eps_per[act]) is independent of
 * the embedding, so `build` advances `next` by up to 32 iterations at once (bit-exact per iteration) and emits, per
 * iteration and per L2 slice of the embedding, the compacted columns of the edges that fire; `grad` reads those lists.
 *   plan : blk_base (ceil(n_rows/64) + 1) int64 = static list regions of the 64-row schedule blocks; the last entry is
 *          the number of int32 entries `list` needs, to which the caller adds 64 entries of slack.  scratch:
 *          ceil(n_rows/64) int64.
 *   build: hdr = tdr_umap_sched_hdr_entries(...) 8-byte records {segment start, length | active count << 16} per
 *          (iteration, slice, row); err = device int (1: region overflow, 2: segment > 65535 / list > 2^32 entries)
 *   grad : nc = row width of Z / grad (1..32: exact kernels for 2 and 3, zero-padded register instances for the rest);
 *          t_local = n_iter - t0 of the last build; acc = (n_rows, 2 nc) floats when n_slices > 1; n_slices in
 *          {1, 2, 4, 8} (tdr_umap_sched_slices = automatic choice); geom: low 4 bits = lanes per row (0 = default), bit 4
 *          (16) = all slices in ONE launch spread over the XCDs (workgroup b takes slice (b % 8) / (8 / n_slices)) plus a
 *          combine kernel -- acc then holds n_slices planes of (n_rows, 2 nc) floats; same gradient bit for bit; bit 5 (32):
 *          see tdr_umap_sched_step_f32. */
int tdr_umap_sched_slices(int64_t n_total, int nc);
/* loop layout: every row's (cols, eps_per) reordered by ascending eps_per (often-firing edges first) */
int tdr_umap_sched_layout_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, int64_t n_rows,
                              int32_t* cols_out, float* eps_out, void* stream);
int64_t tdr_umap_sched_hdr_entries(int64_t n_rows, int block_iters, int n_slices);
int tdr_umap_sched_plan_f32(const int64_t* rowptr, const float* eps_per, int64_t n_rows, int block_iters, int64_t* scratch,
                            int64_t* blk_base, void* stream);
int tdr_umap_sched_build_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, float* next, int64_t n_rows,
                             int64_t n_total, int t0, int n_iters, int n_slices, const int64_t* blk_base, int32_t* list,
                             void* hdr, int* err, void* stream);
/* Round 4: the schedule build on GROUP-ORDERED loop state (csrc/tdr_umap_sched.hip, umap_sched_build2_kernel).  `group`
 * sorts the edges of every 16 consecutive rows (row-major, rows already in (period, column) order: `layout`) stably by
 * firing-period class -> cols_g / eps_g (nnz), rs_g (nnz bytes: local row | slice << 4), order_g (nnz int32: row-major
 * position relative to the group's first edge); `ungroup` moves per-edge values (epoch_of_next_sample) back to the
 * row-major order; `plan_groups` = `plan` with 16-row regions (grp_base: ceil(n_rows/16) + 1 int64, scratch:
 * ceil(n_rows/16) int64); `build_groups` = `build` on that state: one wavefront per group with equally busy lanes, the
 * list written through an LDS stage (stage = entries, 0 = default) with coalesced stores.  Same records, same counters
 * (bit-exact umap.py:243-247); a segment lists the row's firing edges in (period, column) order. */
int tdr_umap_sched_group_f32(const int64_t* rowptr, const int32_t* cols, const float* eps_per, int64_t n_rows, int64_t n_total,
                             int n_slices, int32_t* cols_g, float* eps_g, uint8_t* rs_g, int32_t* order_g, int* err, void* stream);
int tdr_umap_sched_ungroup_f32(const int64_t* rowptr, const int32_t* order_g, const float* vals_g, int64_t n_rows, float* vals_rm,
                               void* stream);
int tdr_umap_sched_plan_groups_f32(const int64_t* rowptr, const float* eps_per, int64_t n_rows, int block_iters, int64_t* scratch,
                                   int64_t* grp_base, void* stream);
int tdr_umap_sched_build_groups_f32(const int64_t* rowptr, const int32_t* cols_g, const float* eps_g, const uint8_t* rs_g, float* next_g,
                                    int64_t n_rows, int t0, int n_iters, int n_slices, const int64_t* grp_base, int32_t* list,
                                    void* hdr, int* err, int stage, void* stream);
int tdr_umap_sched_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list,
                            const void* hdr, int t_local, int n_slices, float a, float b, int n_iter, int neg_rate,
                            int n_negatives, const int64_t* neg_inj, uint64_t seed, float exag, float rep, float eps,
                            float* grad, float* acc, int geom, void* stream);
/* geom & 32 (with 16): the joint launch leaves the per-slice planes in acc and this call finishes the iteration: combine
 * (clamps of umap.py:262,290) -> grad (n_rows, nc), then the torch.optim.SGD(momentum) step of tdr_sgd_step_f32 on the rows
 * Z (n_rows, nc), in one kernel with the same bits as the two. */
int tdr_umap_sched_step_f32(const float* acc, int n_slices, int nc, int64_t n_rows, float exag, float rep, float* grad, float* Z,
                            float* buf, float lr, float momentum, int first, int* nan_flag, int n_iter, void* stream);
/* Round 6: the same gradient with the negatives served from LDS (csrc/tdr_umap_pool.hip; replaces the sampling of
 * neighbor_embedding/base.py:617-649 for the UMAP loop).  Lists and records of a ONE-slice schedule (n_slices = 1 in plan /
 * build).  A workgroup owns a global block of rows and stages, per iteration, a pool of runs of 16 (8) consecutive rows of Z
 * (each run uniform over the ceil(N / 16) runs, counter hash of (seed, iteration, block, slot)) into LDS with coalesced loads;
 * every row draws its min(neg_rate * act, n_negatives) items uniformly from the pool.  Marginal law of an item: uniform over
 * the rows; a draw of the row itself or of the padding of the last run contributes zero (dropped).  One lane per row, rows of a
 * block sorted by active count; a row's sums depend on the row alone (a row-sharded launch gives the same bits).  Z must be
 * 16-byte aligned; nc in {2, 3} (tdr_umap_pool_supported).  geom: 0 = the default geometry (512 threads x 2 rows, 256 runs of 8 rows), 1..6 = tuning geometries
 * (threads per block / rows per thread / pool runs / rows per run: TDR_POOL_GEOMS of csrc/tdr_umap_pool.hip) -- the sampler's
 * stream depends on rows per block, runs and run length.  tdr_umap_pool_grad_debug_f32: measurement hook (tools/umap_pool_perf.py):
 * the same launch through an instrumented instance with parts switched off (`ablate`) and per-block phase time stamps (`times`).
 * tdr_umap_pool_debug_negatives: test hook, the global row of every item drawn for rows with nuse[r] items into out (n_rows,
 * width) int64: -1 beyond the row's count, -2 a dropped draw. */
int tdr_umap_pool_supported(int nc);
int tdr_umap_pool_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list, const void* hdr,
                           int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed, float exag,
                           float rep, float eps, float* grad, int geom, void* stream);
/* the same launch with torch.optim.SGD's step (affinity_matcher.py:427; momentum 0) in it: stepped rows z - lr g -> Z_out, the OTHER
 * embedding buffer (n_total, nc) -- every row of the launch reads the old positions in Z --, nan_flag as tdr_sgd_step_f32 sets it
 * (:315); grad may be NULL (nobody reads the gradient of this iteration). */
int tdr_umap_pool_grad_step_f32(const float* Z, float* Z_out, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list,
                                const void* hdr, int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed,
                                float exag, float rep, float eps, float* grad, float lr, int* nan_flag, int geom, void* stream);
int tdr_umap_pool_debug_negatives(uint64_t seed, int n_iter, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nuse, int geom,
                                  int width, int64_t* out, void* stream);
int tdr_umap_pool_grad_debug_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* list, const void* hdr,
                                 int t_local, float a, float b, int n_iter, int neg_rate, int n_negatives, uint64_t seed, float* grad,
                                 int geom, int ablate, void* times, void* stream);
/* Round 4: peer exchange (csrc/tdr_peerx.hip) -- tdr_ctx_allgather_rows's contract without a collective library: every rank
 * writes its stepped rows into a (fine-grained) staging block of every peer over xGMI, raises a generation flag there, waits
 * for the flags raised at it and copies the staged rows into its embedding; two launches per exchange.  Peers are mapped
 * through HIP IPC handles (tdr_peerx_handles -> any transport -> tdr_peerx_open).  world <= 16, one node.
 * tdr_peerx_allgather_rows has the signature of tdr_umap_loop_desc.gather (never captured into a graph, see there).
 * tdr_peerx_error stays 1 once a wait ran out: the context is then to be destroyed on every rank (a new one starts clean). */
int tdr_peerx_create(void** out, int rank, int world, int64_t capacity_floats);
int tdr_peerx_handles(void* ctx, void* out128);
int tdr_peerx_open(void* ctx, const void* all_handles);
int tdr_peerx_set_rows(void* ctx, int64_t n_total);
int tdr_peerx_set_wait_limit(void* ctx, int64_t spins);   /* bounded flag wait of a pull (default 2^22 spins, a few seconds) */
int tdr_peerx_fine_grained(void* ctx);
int tdr_peerx_allgather_rows(void* ctx, float* Z, int nc, void* stream);
int tdr_peerx_error(void* ctx);
int tdr_peerx_destroy(void* ctx);
/* Round 6, measurement only (tools/rank_share.py, bench.py --emulate-rank): loopback stand-in of the exchange for ONE rank of a
 * `world`-rank fit run alone on one GPU -- the bytes it would send (its chunk, world - 1 times) and receive (every other row, as
 * they stood at the first call) are moved inside the device; tdr_emulx_allgather_rows has tdr_umap_loop_desc.gather's signature. */
int tdr_emulx_create(void** out, int rank, int world, int64_t n_total, int nc);
int tdr_emulx_allgather_rows(void* ctx, float* Z, int nc, void* stream);
int tdr_emulx_destroy(void* ctx);
/* The optimisation loop of affinity_matcher.py:288-352 for UMAP's closed-form step + torch.optim.SGD behind one handle
 * (csrc/tdr_umap_sched.hip): windows of <= block_iters iterations (schedule build + per iteration n_slices gradient
 * passes + the SGD step [+ a row all-gather]) are captured into HIP graphs and replayed; the iteration base lives in
 * device memory.  lr_table: max_iter device floats; norm2: ceil(max_iter / check_interval) device floats, caller-zeroed
 * (squared gradient norm at the iterations the reference inspects, :331-349); nan_flag: device int (first NaN iteration
 * + 1, :315); snap: optional (n_rows, nc) copy of the stepped rows taken at those iterations (lets a host that runs whole
 * windows ahead return the state at which the reference stops, :343-349); scratch: >= 4 bytes of device memory; gather: optional `int (*)(void* ctx, float* Z, int nc, void* stream)`
 * run after every step (tdr_ctx_allgather_rows of a tdr_ctx_create context, or tdr_peerx_allgather_rows), NULL = single
 * process.  With a gather callback tdr_umap_loop_run ignores use_graph and enqueues plain launches unless the descriptor says
 * `gather_capturable` (a communicator call belongs to its library's capture rules; the peer exchange keeps its generation in
 * device memory since round 6 and may be replayed). */
typedef struct tdr_umap_loop_desc {
    float* Z; int nc; int64_t n_total, row0, n_rows;
    const int64_t* rowptr; const int32_t* cols; const float* eps_per; float* next;
    const int64_t* blk_base; int32_t* list; void* hdr; int* err; float* acc; float* grad; float* mom_buf;
    float a, b; int neg_rate, n_negatives; uint64_t seed; float exag, rep, eps; int n_slices, block_iters;
    const float* lr_table; int max_iter; float momentum; int first_iter; int check_interval; float* norm2; float* snap; int* nan_flag;
    void* scratch; void* gather; void* gather_ctx; int geom;
    const uint8_t* rs;   /* non-NULL: cols / eps_per / next are group-ordered (tdr_umap_sched_group_f32), blk_base = grp_base */
    int pool;            /* 0: i.i.d. negatives gathered from L2; g + 1: negatives from the LDS pool, geometry g (tdr_umap_pool_grad_f32; n_slices = 1) */
    int gather_capturable;   /* 1: the gather callback enqueues kernels only and bakes no per-call state into their arguments
                                (tdr_peerx_allgather_rows since round 6, tdr_emulx_allgather_rows): windows are captured and replayed with it */
    float* Z_alt;            /* optional second embedding buffer (n_total, nc), 16-byte aligned, != Z: with pool negatives and momentum 0 the
                                gradient launch carries the SGD step -- iteration t of a window reads buffer t & 1 and writes the other, the
                                gather callback is handed the written one; every window ends with the current rows in Z */
} tdr_umap_loop_desc;
int tdr_umap_loop_create(void** out, const tdr_umap_loop_desc* d);
int tdr_umap_loop_run(void* loop, int it0, int n_iters, int use_graph, void* stream);
int tdr_umap_loop_destroy(void* loop);
/* gradients of neighbor_embedding/largevis.py:181-201 (kind 0), tsne.py:162-170 (kind 1, attraction only),
 * sne.py:160-168 (kind 2, attraction only) and infotsne.py:178-197 (kind 3: Student-t attraction + the row
 * log-sum-exp over the sampled negatives, rep_coef = 2 * repulsion_strength / N) */
int tdr_ne_grad_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                    const float* P, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, int kind,
                    float exag, float rep_coef, int n_neg, const int64_t* neg_inj, uint64_t seed, int n_iter,
                    float* grad, void* stream);
/* gradient of neighbor_embedding/sne.py:170-179 (dense row log-sum-exp of -d), two passes */
int tdr_sne_rowsum_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* R, void* stream);
int tdr_sne_repulsion_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const float* R,
                          float coef, float* grad, void* stream);
/* float64 twins (round 4): SNE's dense repulsion and PaCMAP's pair gradient for float64 inputs (tests/test_neighbor_embedding.py:34,
 * 55-74 of the reference run every method in both dtypes); same argument meaning. */
int tdr_sne_rowsum_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double* R, void* stream);
int tdr_sne_repulsion_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const double* R, double coef,
                          double* grad, void* stream);
int tdr_pacmap_grad_f64(const double* Z, int nc, int64_t n, const int64_t* near_idx, int m_near, double w_nb, const int64_t* mid_idx,
                        int m_mid, double w_mn, const int64_t* far_idx, int m_far, double w_fp, double* grad, void* stream);
/* float64 SNEkhorn (csrc/tdr_khorn_f64.hip; round 4): the reductions of tdr_sea_rowstats_f32 / tdr_sinkhorn_pass_f32 /
 * tdr_student_matvec_f32 / tdr_khorn_grad_nc_f32 / tdr_khorn_grad_unrolled_f32 in float64 on a DENSE (n, n) float64 matrix of
 * squared distances (tdr_knn_f64 with k = 0; symmetric), n <= tdr_pairs_f64_max_rows().  ws: tdr_pairs_f64_workspace_bytes(n,
 * n_state) bytes, n_state = 3 (row statistics) / 1 (student sum) / nc (forces).
 *   affinity/entropic.py:518-534: side (n, 2) = (mu, e) -> out (n, 3) = (sum p, sum p lp, sum p C), lp = (mu_i + mu_j - 2 C_ij) / (e_i + e_j)
 *   affinity/entropic.py:728-748: side (n, nc + 1) = (z, v) -> out_j = sum_i v_i / (1 + |z_j - z_i|^2 [+ diag_add at i = j])
 *   neighbor_embedding/tsnekhorn.py:210-230: side (n, nc + 3) = (mu, e, z, exp(dual)) -> grad (n, nc)
 *   neighbor_embedding/tsnekhorn.py:134,224-227 (unrolling): side (n, nc + 12) = (mu, e, z, a[5], b[5]) -> grad (n, nc) */
int64_t tdr_pairs_f64_max_rows(void);
int64_t tdr_pairs_f64_workspace_bytes(int64_t n, int n_state);
int tdr_sea_rowstats_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, double* out, void* ws, int64_t ws_bytes,
                               void* stream);
int tdr_student_sum_f64(const double* side, int nc, int64_t n, int zero_diag, double diag_add, double* out, void* ws, int64_t ws_bytes,
                        void* stream);
int tdr_khorn_grad_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, int nc, double log_n, double* grad, void* ws,
                             int64_t ws_bytes, void* stream);
int tdr_khorn_grad_unrolled_dense_f64(const double* C, int64_t n, int64_t ldc, const double* side, int nc, double log_n, double* grad,
                                      void* ws, int64_t ws_bytes, void* stream);
/* COSNE (neighbor_embedding/cosne.py:162-193, float64 like the reference's ManifoldParameter): closed-form gradient of
 *   -sum P log Q (kNN graph) + log sum_ij Q_ij (dense, never materialised) + lam * mean (||x||^2 - d_H(z,0)^2)^2,
 *   Q = gamma / (d_H^2 + gamma^2), d_H^2 = arccosh(1 + 2|zi-zj|^2/((1-|zi|^2)(1-|zj|^2)) + 1e-8)^2
 * (distance/torch.py:101-107, distance/base.py:392-398).  Pass 1 = all-pairs sums for the chunk rows (rowsum of Q +
 * unscaled gradient sums kept in ws), pass 2 = attraction over both ends of every kNN edge + scaling by S = sum of all
 * rowsums (device scalar, all-reduced by the caller) + norm term.  nc in 2..8. */
int tdr_cosne_splits(int64_t n_total, int64_t n_rows);
int64_t tdr_cosne_workspace_bytes(int64_t n_total, int64_t n_rows, int nc);
int tdr_cosne_pairs_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double gamma,
                        double* rowsum, void* ws, int64_t ws_bytes, void* stream);
int tdr_cosne_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                       const float* P_in, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val,
                       const double* S, const float* x_norm, double gamma, double lam, double exag, double rep,
                       const void* ws, int64_t ws_bytes, double* grad, void* stream);
/* utils/radam.py:139-167 + utils/manifold.py:207-330 (PoincareBallManifold, c = 1): one RiemannianAdam step on n_rows
 * rows, in place; step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) from the caller; maxnorm = 1 - 1e-5 (float64). */
int tdr_radam_poincare_f64(double* Z, const double* egrad, double* exp_avg, double* exp_avg_sq, double* rgrad,
                           int64_t n_rows, int nc, double beta1, double beta2, double eps, double step_size,
                           double maxnorm, int* nan_flag, int n_iter, void* stream);
/* neighbor_embedding/pacmap.py:213-239, the mid-near pairs of one iteration for all rows and slots in one launch: 6 candidates
 * j = r + (r >= i), r uniform in [1, n - 2] (counter hash keyed by seed / iteration / row / slot / candidate), ranked by their
 * input-space distance to x_i (mode 0 squared or plain Euclidean, 2 manhattan, 3 angular), the second nearest kept.
 * X (n, d) row stride ldx; out (n, n_mid) int64: emit_index 0 = the POSITION (0..5) of that candidate among the six, which is what
 * the reference stores and uses (its `topk(...).indices[:, 1]`), 1 = the candidate's row (test hook) */
int tdr_pacmap_mid_near_f32(const float* X, int64_t ldx, int d, int64_t n, int n_mid, int mode, uint64_t seed, int n_iter,
                            int emit_index, int64_t* out, void* stream);
/* gradient of neighbor_embedding/pacmap.py:213-265 (near / mid-near / further pair losses) */
int tdr_pacmap_grad_f32(const float* Z, int nc, int64_t n, const int64_t* near_idx, int m_near, float w_nb,
                        const int64_t* mid_idx, int m_mid, float w_mn, const int64_t* far_idx, int m_far, float w_fp,
                        float* grad, void* stream);
/* measurement / test switch of tdr_ne_grad_perm_f32: 1 = never split the launch into the two halves of the index range (one
 * visit per row), 0 (default) = split where the negatives dominate a row (>= 64 items) and the gathered tables exceed an XCD's
 * L2; returns the previous value */
int tdr_ne_grad_perm_halves(int mode);
/* Round 6: LargeVis's gradient (neighbor_embedding/largevis.py:181-201) of rows [row0, row0 + n_rows) -- nn / P (n_rows, k) and the
 * in-edges t_* of those rows, grad (n_rows, nc) COMPLETE for them: a rank of a row-sharded fit steps its rows -- with the negatives drawn by the
 * RUN-permutation sampler -- the keyed cyclic order of tdr_ne_grad_perm_f32 over runs of 16 consecutive rows, a hashed rotation
 * inside the run -- and served from LDS: a row's draw is uniform over the rows outside its own run, every row is the far endpoint of
 * n_neg pairs, both shares of a pair are pulled (csrc/tdr_embed.hip, ne_pull4_runs_kernel).  grad is WRITTEN.  2 / 3 components,
 * n_neg <= 8, Z 16-byte aligned. */
int tdr_ne_grad_runs_supported(int nc, int64_t n_total, int n_neg);
int tdr_ne_grad_runs_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn, const float* P, int k,
                         const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, float exag, float rep_coef, int n_neg,
                         uint64_t seed, int n_iter, float* grad, void* stream);
int tdr_runs_negatives_debug(uint64_t seed, int n_iter, int64_t n_total, int n_neg, int64_t* fwd, int64_t* inv, void* stream);
/* measurement / test switch of tdr_ne_grad_perm_f32's LargeVis launch (kind 0, <= 8 negatives, 2 / 3 components): lanes per row --
 * 4 (default since round 6: every lane issues all its index loads and gathers before it uses any) or 16 (rounds 3-5); same terms,
 * another association of a row's fp32 sum; returns the previous value */
int tdr_ne_grad_perm_lanes(int lanes);
/* tdr_ne_grad_f32 with the negatives drawn as keyed permutations of the rows (kinds 0 = LargeVis, 3 = InfoTSNE): a row pulls
 * its own draws j = P(i) AND the draws that hit it (i' = P^-1(i)) -- no atomics on the far endpoints.  Per row the draws are
 * uniform and independent across columns / iterations like neighbor_embedding/base.py:628-636; within one column they are
 * distinct across rows.  Needs the transposed graph.  rowsum_ws: kind 3 only, n_total floats; kind 3 takes all rows at once. */
int tdr_ne_grad_perm_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn,
                         const float* P_, int k, const int64_t* t_rowptr, const int32_t* t_src, const float* t_val, int kind,
                         float exag, float rep_coef, int n_neg, uint64_t seed, int n_iter, float* rowsum_ws, float* grad,
                         void* stream);
int tdr_perm_negatives_debug(uint64_t seed, int n_iter, int64_t n_total, int n_neg, int64_t* fwd, int64_t* inv, void* stream);
/* gradient pieces of neighbor_embedding/tsne.py:172-180 (dense Student-t partition function) */
int tdr_tsne_repulsion_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* F, double* S,
                           void* stream);
/* the same with the columns spread over several workgroups per row block (a 256-row block against all columns is one
 * workgroup: N = 50k is 196 workgroups for 256 CUs): `ws` of tdr_tsne_repulsion_workspace_bytes() bytes holds the
 * per-segment partial forces, added in segment order; NULL / too small / 0 bytes needed: the unsplit launch */
int64_t tdr_tsne_repulsion_workspace_bytes(int64_t n_total, int64_t n_rows, int nc);
int tdr_tsne_repulsion_split_f32(const float* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, float* F, double* S,
                                 void* ws, int64_t ws_bytes, void* stream);
int tdr_add_scaled_f32(float* grad, const float* F, const double* S, float coef, int64_t n, void* stream);
/* torch.optim.SGD step of affinity_matcher.py:427 (+ the NaN guard of :315) */
int tdr_sgd_step_f32(float* Z, const float* grad, float* buf, int64_t n, float lr, float momentum, int first,
                     int* nan_flag, int n_iter, void* stream);
int tdr_fill_f32(float* p, int64_t n, float v, void* stream);

/* Optional workspace of the matrix-free pair scans below (ws / ws_bytes; NULL = none): with a buffer of
 * tdr_pair_scan_workspace_bytes(n, n_state) bytes the database is split into segments so that the launch fills the chip
 * evenly (unsplit, N = 200k is 1.5 rounds of workgroups); per-segment statistics are folded in segment order by a second
 * kernel.  n_state: floats of running statistics per row -- 4 for the row statistics, 2 for the log-sum-exp, the instance
 * width nc for the forces.  0 = a scan of this size is not split. */
int64_t tdr_pair_scan_workspace_bytes(int64_t n, int n_state);
/* ---- K7 / K8: matrix-free dense affinities (TSNEkhorn) ------------------------------------------------
 * affinity/entropic.py:37-42,518-565 (_log_Pse, row entropy / logsumexp of the dual-ascent loop) */
int tdr_sea_rowstats_f32(const float* packed, int64_t n, int d, const float* side, int exclude_diag, float diag_add,
                         float* psum, float* ent, void* ws, int64_t ws_bytes, void* stream);
/* the same with energy[i] = sum_j exp(lp_ij) C_ij: the dual objective of the LBFGS path (entropic.py:483-491) is
 * -sum(energy) - <e, target - ent> + <mu, psum - 1> */
int tdr_sea_rowstats3_f32(const float* packed, int64_t n, int d, const float* side, int exclude_diag, float diag_add,
                          float* psum, float* ent, float* energy, void* ws, int64_t ws_bytes, void* stream);
/* affinity/entropic.py:728-734 on the input points, matrix-free: lse[i] = LSE_j(log K_ij + f_j), log K = -C / eps
 * (student != 0: -log(1 + C) / eps); the caller forms the symmetric Sinkhorn update f <- 0.5 (f - lse). */
int tdr_sinkhorn_lse_f32(const float* packed, int64_t n, int d, const float* f, float inv_eps, int student, int exclude_diag,
                         float diag_add, float* lse, void* ws, int64_t ws_bytes, void* stream);
/* gradient of neighbor_embedding/tsnekhorn.py:210-230 w.r.t. the embedding (duals detached) */
int tdr_khorn_grad_f32(const float* packed, int64_t n, int d, const float* side, float log_n, float* grad, void* stream);
/* the same for an (n, nc) embedding, nc in {2, 3, 4, 8, 16, 32} (other widths: pad z with zero columns up to the next one);
 * side: (n, 3 + nc) row-major (mu, e, z_0 .. z_{nc-1}, exp(dual)) */
int tdr_khorn_grad_nc_f32(const float* packed, int64_t n, int d, const float* side, int nc, float log_n, float* grad,
                          void* ws, int64_t ws_bytes, void* stream);
/* TSNEkhorn(unrolling=True), neighbor_embedding/tsnekhorn.py:134,224-227: gradient of CE(P, log Q) with autograd THROUGH the
 * <= 5 Sinkhorn updates (affinity/entropic.py:729-736 with with_grad=True), in closed form:
 *   4 sum_j [P_ij + w_ij sum_k (a^k_i b^k_j + a^k_j b^k_i)] w_ij (z_i - z_j),  w = 1/(1+d_ij);
 * side: (n, 2 + nc + 10) row-major (mu, e, z_0 .. z_{nc-1}, a^1..a^5, b^1..b^5) -- a^k = adjoint of update k / (4 s^k),
 * b^k = exp(f^{k-1} - max), zero for updates that did not run. */
int tdr_khorn_grad_unrolled_f32(const float* packed, int64_t n, int d, const float* side, int nc, float log_n, float* grad,
                                void* ws, int64_t ws_bytes, void* stream);
/* optional workspace of the two all-pairs passes on the embedding below (ws / ws_bytes; NULL = none): the columns are then
 * spread over several workgroups per block of 256 rows and the per-segment sums added in order; 0 = not split at this size */
int64_t tdr_student_workspace_bytes(int64_t n);
/* affinity/entropic.py:733-740: one symmetric log-domain Sinkhorn update, student kernel on the embedding
 * (nc in {2, 3, 4, 8, 16, 32}) */
int tdr_sinkhorn_pass_f32(const float* Z, int nc, const float* f, const float* Ef, float fmax, int64_t n, int zero_diag,
                          float diag_add, float* f_new, float* resid2, void* ws, int64_t ws_bytes, void* stream);
/* out_j = sum_i v_i / (1 + |z_i - z_j|^2): the mat-vec of the adjoint of that update (what autograd runs backwards through
 * affinity/entropic.py:735 when with_grad=True); the diagonal term is weighted 1 / (1 + diag_add) when zero_diag */
int tdr_student_matvec_f32(const float* Z, int nc, const float* v, int64_t n, int zero_diag, float diag_add, float* out,
                           void* ws, int64_t ws_bytes, void* stream);

/* ---- float64 embedding loop (csrc/tdr_embed_f64.hip) ------------------------------------------------------------------
 * The reference computes in the dtype of its input and runs every neighbour-embedding method in float32 and float64
 * (tests/test_neighbor_embedding.py:34,55-74): float64 twins of tdr_umap_prepare_f32 / tdr_umap_grad_f32 (per-step form,
 * umap.py:215-292), tdr_ne_grad_f32 (largevis.py:181-201, tsne.py:162-170, sne.py, infotsne.py), tdr_tsne_repulsion_f32 /
 * tdr_add_scaled_f32 (tsne.py:172-180) and tdr_sgd_step_f32 (affinity_matcher.py:427-429); same argument meaning. */
int tdr_umap_prepare_f64(const double* vals, int64_t nnz, int max_iter, double* eps_per, double* next, void* scratch, void* stream);
int tdr_umap_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int64_t* rowptr,
                      const int32_t* cols, const double* eps_per, double* next, double a, double b, int n_iter, int neg_rate,
                      int n_negatives, const int64_t* neg_inj, uint64_t seed, double exag, double rep, double eps, double* grad,
                      void* stream);
int tdr_ne_grad_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, const int32_t* nn, const double* P_,
                    int k, const int64_t* t_rowptr, const int32_t* t_src, const double* t_val, int kind, double exag,
                    double rep_coef, int n_neg, const int64_t* neg_inj, uint64_t seed, int n_iter, double* grad, void* stream);
int tdr_tsne_repulsion_f64(const double* Z, int nc, int64_t n_total, int64_t row0, int64_t n_rows, double* F, double* S, void* stream);
int tdr_add_scaled_f64(double* grad, const double* F, const double* S, double coef, int64_t n, void* stream);
int tdr_sgd_step_f64(double* Z, const double* grad, double* buf, int64_t n, double lr, double momentum, int first, int* nan_flag,
                     int n_iter, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TORCHDR_AMD_H */
