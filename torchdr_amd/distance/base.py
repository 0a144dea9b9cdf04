"""Pairwise distances / exact kNN on MI355X -- drop-in for ``torchdr.distance``.

Mirrors the reference's function surface (``/root/reference/torchdr/distance/base.py:22-249``
``pairwise_distances`` and ``:252-405`` ``pairwise_distances_indexed``): same argument names,
meaning and error messages.  ``backend`` is accepted for signature compatibility; every backend
value ("faiss", "keops", None, a FaissConfig-like object) maps onto the one exact HIP kernel
(K1, ``csrc/tdr_knn.hip``) -- exact brute-force search is what Faiss ``Flat`` and the torch
path both compute.
"""

from typing import Optional

import torch

from torchdr_amd import _lib
from torchdr_amd.distributed import DistributedContext

LIST_METRICS = ["euclidean", "sqeuclidean", "angular"]
_METRIC_ID = {"sqeuclidean": 0, "euclidean": 1, "angular": 2}

# Value the reference adds to the diagonal when exclude_diag=True (distance/torch.py:115).
_DIAG_ADD = 1e12

# bench.py sets this to a list to collect (start_event, end_event, n_queries) around every scan launch
# (HIP events on the launch stream); None = no instrumentation.
PROFILE = None


class PackedPoints:
    """A point block rewritten into MFMA tile images (+ squared norms) on the device.

    Packing costs one pass over the N x D block; callers that search the same database
    repeatedly (row-chunked / multi-GPU search) keep the object and reuse it.
    """

    def __init__(self, X: torch.Tensor):
        _lib.require_gpu(X, "X")
        if X.dim() != 2:
            raise ValueError("[TorchDR] ERROR : input must be 2-D (n_samples, n_features).")
        if X.dtype != torch.float32:
            raise NotImplementedError(
                "[torchdr_amd] only float32 inputs are supported by the HIP distance kernels "
                f"(got {X.dtype})."
            )
        if X.stride(1) != 1:
            X = X.contiguous()
        L = _lib.lib()
        self.n, self.d = X.shape
        nfl = L.tdr_packed_floats(self.n, self.d)
        if nfl == 0:
            raise NotImplementedError(
                f"[torchdr_amd] feature dimension {self.d} > 256 is not supported by the MFMA kNN kernel yet."
            )
        self.data = torch.empty(nfl, dtype=torch.float32, device=X.device)
        self.norms = torch.empty(self.n, dtype=torch.float32, device=X.device)
        _lib.check(
            L.tdr_pack_rows_f32(
                _lib.ptr(X), self.n, self.d, X.stride(0), _lib.ptr(self.data), _lib.ptr(self.norms),
                _lib.stream_ptr(),
            ),
            "tdr_pack_rows_f32",
        )
        self.device = X.device


def knn_packed(
    Q: PackedPoints, Y: PackedPoints, k: int, metric: str, exclude_self: bool, q_offset: int = 0,
    q_rows: Optional[slice] = None,
):
    """k nearest database rows of ``Y`` for the queries ``Q`` (or the row slice ``q_rows`` of Q,
    which must start on a multiple of 32).  Returns (values (n,k) fp32 ascending, indices (n,k) int32)."""
    L = _lib.lib()
    if Q.d != Y.d:
        raise ValueError("[TorchDR] ERROR : X and Y must have the same number of features.")
    d = Q.d
    if q_rows is None:
        q0, q1 = 0, Q.n
    else:
        q0, q1 = q_rows.start, q_rows.stop
        if q0 % 32 != 0:
            raise ValueError("[torchdr_amd] query slices must start on a multiple of 32 rows.")
    nq = q1 - q0
    kmax = L.tdr_knn_max_k(d)
    if k > kmax:
        raise NotImplementedError(
            f"[torchdr_amd] k={k} exceeds the LDS-resident list capacity ({kmax}) for D={d}."
        )
    dev = Y.device
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    ws_bytes = L.tdr_knn_workspace_bytes(nq, Y.n, k)
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.int64, device=dev)
    tile_floats = L.tdr_packed_floats(32, d)
    qdata = Q.data[(q0 // 32) * tile_floats:]
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(
        L.tdr_knn_packed_f32(
            _lib.ptr(qdata), nq, q_offset + q0, _lib.ptr(Y.data), Y.n, d, k, _METRIC_ID[metric],
            1 if exclude_self else 0, _lib.ptr(out_d), _lib.ptr(out_i), _lib.ptr(ws), ws_bytes,
            _lib.stream_ptr(),
        ),
        "tdr_knn_packed_f32",
    )
    if PROFILE is not None:
        ev1.record()
        PROFILE.append((ev0, ev1, nq))
    return out_d, out_i


def dense_packed(Q: PackedPoints, Y: PackedPoints, metric: str, exclude_self: bool, q_offset: int = 0):
    L = _lib.lib()
    out = torch.empty((Q.n, Y.n), dtype=torch.float32, device=Y.device)
    _lib.check(
        L.tdr_dense_dist_packed_f32(
            _lib.ptr(Q.data), Q.n, q_offset, _lib.ptr(Y.data), Y.n, Q.d, _METRIC_ID[metric],
            1 if exclude_self else 0, _DIAG_ADD, _lib.ptr(out), out.stride(0), _lib.stream_ptr(),
        ),
        "tdr_dense_dist_packed_f32",
    )
    return out


def _to_device(X, device):
    if device != "auto" and str(X.device) != str(device):
        X = X.to(device)
    return X


def pairwise_distances(
    X: torch.Tensor,
    Y: Optional[torch.Tensor] = None,
    metric: str = "euclidean",
    backend=None,
    exclude_diag: bool = False,
    k: Optional[int] = None,
    return_indices: bool = False,
    device: str = "auto",
    distributed_ctx: Optional[DistributedContext] = None,
):
    r"""Compute pairwise distances (or the k smallest per row) between two point sets.

    Same contract as the reference (``distance/base.py:22-249``, ``distance/torch.py:21-125``):
    ``C_ij = (||x_i||^2 + ||y_j||^2) - 2 x_i.y_j`` in fp32 (``sqeuclidean``, not clamped),
    ``sqrt(clamp(C, 0))`` (``euclidean``), ``-x_i.y_j`` (``angular``); when ``Y`` is None / is X
    and ``exclude_diag`` the self distance is excluded (``+1e12`` on the diagonal of the dense
    matrix); with ``k`` the k smallest per row, ascending, indices int32; ``k >= n_columns``
    returns the dense matrix and ``None`` indices (``utils/utils.py:203-204``).  With a
    ``distributed_ctx`` each rank searches its row chunk against the full database
    (``distance/base.py:160-211``).

    Rows are ordered by (distance, index); ``torch.topk``'s order among exactly tied
    distances is unspecified, so that is the only place results can differ from the
    reference's CPU backend (parity protocol).
    """
    if metric not in LIST_METRICS:
        raise ValueError(f"[TorchDR] ERROR : The '{metric}' distance is not supported.")
    if not isinstance(X, torch.Tensor):
        raise NotImplementedError("[torchdr_amd] DataLoader input is out of scope (SURVEY.md section 8f).")

    X = _to_device(X, device)
    _lib.require_gpu(X, "X")
    self_search = Y is None or Y is X
    if not self_search:
        Y = _to_device(Y, device)
        _lib.require_gpu(Y, "Y")

    # --- distributed: queries = this rank's chunk, database = full X (base.py:160-211)
    if distributed_ctx is not None and distributed_ctx.is_initialized:
        if k is None:
            raise ValueError(
                "[TorchDR] Distributed mode requires sparse computation with k-NN. "
                "k cannot be None when distributed_ctx is provided."
            )
        if Y is not None:
            raise ValueError(
                "[TorchDR] Distributed mode does not support cross-distance computation. "
                "Y must be None when distributed_ctx is provided."
            )
        n = X.shape[0]
        c0, c1 = distributed_ctx.compute_chunk_bounds(n)
        Yp = PackedPoints(X)
        Qp = Yp if c0 % 32 == 0 else PackedPoints(X[c0:c1])
        rows = slice(c0, c1) if Qp is Yp else None
        # the reference asks Faiss for k+1 and drops column 0; excluding the query's own row
        # by index is the same thing whenever the point is its own strict nearest neighbour.
        C, I = knn_packed(
            Qp, Yp, k, metric, exclude_self=exclude_diag,
            q_offset=0 if Qp is Yp else c0, q_rows=rows,
        )
        return (C, I) if return_indices else C

    Xp = PackedPoints(X)
    Yp = Xp if self_search else PackedPoints(Y)
    do_exclude = bool(exclude_diag) and self_search
    n_cols = Yp.n

    if k is not None and k < n_cols:
        if do_exclude and k > n_cols - 1:
            raise ValueError("[TorchDR] ERROR : k must be smaller than the number of samples.")
        C, I = knn_packed(Xp, Yp, int(k), metric, do_exclude)
        return (C, I) if return_indices else C

    C = dense_packed(Xp, Yp, metric, do_exclude)
    if return_indices:
        return C, None
    return C


def pairwise_distances_indexed(
    X: torch.Tensor,
    query_indices: Optional[torch.Tensor] = None,
    key_indices: Optional[torch.Tensor] = None,
    Y: Optional[torch.Tensor] = None,
    metric: str = "sqeuclidean",
    backend=None,
    device: str = "auto",
):
    r"""Distances between indexed subsets (reference ``distance/base.py:252-405``).

    The per-query-keys form (``key_indices`` 2-D, the one the embedding loop uses) runs the
    gather kernel ``tdr_indexed_sqdist_f32``: ``sum_c (x_ic - y_jc)^2`` by direct difference
    (``base.py:384-385``).  A key index of -1 wraps to the last row, as PyTorch indexing does
    in the reference.
    """
    if Y is None:
        Y = X
    X = _to_device(X, device)
    Y = _to_device(Y, device)
    _lib.require_gpu(X, "X")
    if metric not in ("sqeuclidean", "euclidean"):
        raise NotImplementedError(f"Metric '{metric}' not implemented for indexed distances")
    if query_indices is not None and query_indices.dim() != 1:
        raise NotImplementedError("2D query indices not yet supported")
    if key_indices is not None and key_indices.dim() not in (1, 2):
        raise ValueError(f"key_indices must be 1D or 2D, got {key_indices.dim()}D")
    if key_indices is None or key_indices.dim() == 1:
        # queries x keys block (reference base.py:357-376, torch.cdist): gather the rows, dense MFMA kernel
        Xq = X if query_indices is None else X[query_indices.to(X.device).long()]
        Yk = Y if key_indices is None else Y[key_indices.to(Y.device).long()]
        return dense_packed(PackedPoints(Xq.float().contiguous()), PackedPoints(Yk.float().contiguous()), metric, False)
    L = _lib.lib()
    Xc = X.contiguous().float()
    Yc = Y.contiguous().float()
    keys = key_indices.to(device=X.device, dtype=torch.int64).contiguous()
    nq = keys.shape[0]
    if query_indices is None:
        q = torch.arange(nq, device=X.device, dtype=torch.int64)
    else:
        q = query_indices.to(device=X.device, dtype=torch.int64).contiguous()
        assert keys.shape[0] == len(q), (
            f"key_indices first dim {keys.shape[0]} must match number of queries {len(q)}"
        )
    out = torch.empty(keys.shape, dtype=torch.float32, device=X.device)
    _lib.check(
        L.tdr_indexed_sqdist_f32(
            _lib.ptr(Xc), Xc.shape[0], Xc.shape[1], _lib.ptr(Yc), Yc.shape[0], _lib.ptr(q), nq,
            keys.shape[1], 1 if metric == "euclidean" else 0, _lib.ptr(keys), _lib.ptr(out),
            _lib.stream_ptr(),
        ),
        "tdr_indexed_sqdist_f32",
    )
    return out
