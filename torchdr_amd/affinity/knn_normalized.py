"""UMAP input affinity on the GPU -- mirror of ``UMAPAffinity``
(reference ``affinity/knn_normalized.py:335-496``)."""

from typing import Union

import torch

from torchdr_amd import _lib
from torchdr_amd.affinity.base import SparseAffinity
from torchdr_amd.utils import check_neighbor_param
from torchdr_amd.utils.sparse import symmetrize_to_csr

_TOL = 1e-6  # utils/root_search.py:13


def umap_sigma_search(C: torch.Tensor, n_neighbors, max_iter: int):
    """rho_i = min_j C_ij; eps_i with sum_j exp(-(C_ij - rho_i)/eps_i) = log2(n_neighbors); P = exp(.).
    (knn_normalized.py:445-465, K3 ``tdr_umap_search_f32``.)"""
    _lib.require_gpu(C, "C")
    if C.dtype == torch.float64:
        C = C.contiguous()
        n, k = C.shape
        rho, eps, P = torch.empty(n, dtype=torch.float64, device=C.device), torch.empty(n, dtype=torch.float64, device=C.device), torch.empty_like(C)
        target = float(torch.log2(torch.tensor(n_neighbors, dtype=torch.float64)))
        _lib.check(_lib.lib().tdr_umap_search_f64(_lib.ptr(C), n, k, target, int(max_iter), _TOL, _lib.ptr(rho), _lib.ptr(eps),
                                                  _lib.ptr(P), _lib.stream_ptr()), "tdr_umap_search_f64")
        return rho, eps, P
    C = C.contiguous().float()
    n, k = C.shape
    rho = torch.empty(n, dtype=torch.float32, device=C.device)
    eps = torch.empty(n, dtype=torch.float32, device=C.device)
    P = torch.empty_like(C)
    target = float(torch.log2(torch.tensor(n_neighbors, dtype=torch.float32)))
    _lib.check(
        _lib.lib().tdr_umap_search_f32(_lib.ptr(C), n, k, target, int(max_iter), _TOL, _lib.ptr(rho), _lib.ptr(eps),
                                       _lib.ptr(P), _lib.stream_ptr()),
        "tdr_umap_search_f32",
    )
    return rho, eps, P


class UMAPAffinity(SparseAffinity):
    r"""UMAP input affinity: :math:`P_{ij} = \exp(-(C_{ij} - \rho_i)/\sigma_i)`,
    :math:`\sum_j P_{ij} = \log_2(\mathrm{n\_neighbors})`, symmetrised as
    :math:`P + P^\top - P \circ P^\top`.  Constructor arguments as in the reference
    (``knn_normalized.py:385-415``)."""

    _float64_kernels = True   # float64 inputs are computed in float64 (csrc/tdr_f64.hip)

    def __init__(self, n_neighbors: float = 30, max_iter: int = 1000, sparsity: bool = True,
                 metric: str = "sqeuclidean", zero_diag: bool = True, device: str = "auto",
                 backend: Union[str, None] = None, verbose: bool = False, compile: bool = False,
                 symmetrize: bool = True, distributed: Union[bool, str] = "auto", _pre_processed: bool = False):
        self.n_neighbors = n_neighbors
        self.max_iter = max_iter
        self.symmetrize = symmetrize
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=backend, verbose=verbose,
                         sparsity=sparsity, compile=compile, distributed=distributed,
                         _pre_processed=_pre_processed)

    def _compute_sparse_affinity(self, X: torch.Tensor, return_indices: bool = True, return_csr: bool = False,
                                 **kwargs):
        n_samples_in = self._get_n_samples(X)
        n_neighbors = check_neighbor_param(self.n_neighbors, n_samples_in)
        if not self.sparsity:
            # dense N x N affinity (knn_normalized.py:443, 488-493): the row search streams each full row
            C_, _ = self._distance_matrix(X, return_indices=True)
            rho, eps, P = umap_sigma_search(C_, n_neighbors, self.max_iter)
            self.register_buffer("rho_", rho, persistent=False)
            self.register_buffer("eps_", eps, persistent=False)
            if self.symmetrize:
                P = P + P.T - P * P.T
            return (P, None) if return_indices else P
        if self.verbose:
            self.logger.info(f"Sparsity mode enabled, computing {n_neighbors} nearest neighbors...")
        from torchdr_amd.utils.phases import phase

        # the cluster-sorted row order a pruned search worked in comes back in `info` (UMAP numbers the points of its loop
        # in that order).  Row-sharded: a UMAP that can run in that numbering says so (`_accept_loop_order`), and the rank
        # then KEEPS the rows of its range of the order instead of exchanging them -- rows and neighbour indices of
        # everything below are positions of the order (`_rows_in_loop_order`).
        info = {"want_loop_order": bool(self.is_multi_gpu and getattr(self, "_accept_loop_order", False))}
        with phase("knn"):
            C_, indices = self._distance_matrix(X, k=int(n_neighbors), return_indices=True, info=info)
        self._row_order = info.get("cluster_order")
        self._rows_in_loop_order = bool(info.get("loop_order"))
        if self.is_multi_gpu and not self._rows_in_loop_order:
            self._row_order = None
        with phase("sigma search"):
            rho, eps, P = umap_sigma_search(C_, n_neighbors, self.max_iter)
        self.register_buffer("rho_", rho, persistent=False)
        self.register_buffer("eps_", eps, persistent=False)

        if not self.symmetrize:
            return (P, indices) if return_indices else P

        self.logger.info("Symmetrizing affinity matrix...")
        if self.is_multi_gpu:
            from torchdr_amd.parallel import exchange_transposed_edges

            with phase("symmetrise: edge exchange (all-to-all)"):
                ext = exchange_transposed_edges(P, indices, self.chunk_start_, n_samples_in, self.world_size)
            with phase("symmetrise"):
                csr = symmetrize_to_csr(P, indices, "sum_minus_prod", row_offset=self.chunk_start_,
                                        n_total=n_samples_in, ext=ext)
        else:
            with phase("symmetrise"):
                # the rows are visited in the search's cluster-sorted order when there is one (same result, local look-ups)
                order = self._row_order[0] if self._row_order is not None else None
                csr = symmetrize_to_csr(P, indices, "sum_minus_prod", n_total=n_samples_in, order=order)
        self._csr_ = csr
        if return_csr:
            return csr
        values, idx = csr.to_padded()
        return (values, idx) if return_indices else values


class PACMAPAffinity(SparseAffinity):
    """Neighbour selection of PaCMAP (reference ``affinity/knn_normalized.py:499-611``): the ``n_neighbors + 50``
    exact nearest neighbours by squared distance, rescaled by ``rho_i * rho_j`` with ``rho_i`` the mean Euclidean
    distance to the 4th-6th neighbours, of which the ``n_neighbors`` smallest are kept.  Returns ``(None,
    indices)`` -- PaCMAP uses the pairs only.  The search is K1 / K1s; the per-row rescale and re-selection
    (N x (n_neighbors + 50) elements) are device tensor ops.  Like the reference it refuses ``distributed``."""

    _float64_kernels = True   # float64 inputs: the float64 kNN (tdr_knn_f64) and dtype-generic tensor ops

    def __init__(self, n_neighbors: float = 10, metric: str = "sqeuclidean", zero_diag: bool = True,
                 device: str = "auto", backend=None, verbose: bool = False, compile: bool = False,
                 distributed=False, _pre_processed: bool = False):
        self.n_neighbors = n_neighbors
        if distributed:
            raise ValueError("[TorchDR] ERROR : PACMAPAffinity does not support distributed.")
        super().__init__(metric=metric, zero_diag=zero_diag, device=device, backend=backend, verbose=verbose,
                         sparsity=True, compile=compile, distributed=distributed, _pre_processed=_pre_processed)

    def _compute_sparse_affinity(self, X: torch.Tensor, return_indices: bool = True, **kwargs):
        n_samples_in = self._get_n_samples(X)
        k = min(self.n_neighbors + 50, n_samples_in)
        k = int(check_neighbor_param(k, n_samples_in))
        C_, temp_indices = self._distance_matrix(X, k=k, return_indices=True)
        # rows are ascending: columns 3..5 are the 4th-6th neighbours (kmin(C_, 6) of the reference :591)
        self.rho_ = torch.sqrt(C_[:, :6])[:, 3:6].mean(dim=1).contiguous()
        C_ = C_ / (self.rho_.unsqueeze(1) * self.rho_[temp_indices.long()])
        local = torch.topk(C_, int(self.n_neighbors), dim=1, largest=False).indices
        final_indices = torch.gather(temp_indices.long(), 1, local)
        if return_indices:
            return None, final_indices
        return C_
