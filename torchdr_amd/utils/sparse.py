"""Sparse symmetrisation on the GPU (K4) -- stands in for ``torchdr/utils/sparse.py``."""

from typing import Tuple

import torch

from torchdr_amd import _lib


class CSRAffinity:
    """Symmetrised affinity graph in CSR form (rows = this rank's chunk, columns global)."""

    def __init__(self, rowptr, cols, vals, row_offset=0, n_total=None):
        self.rowptr = rowptr  # int64 (n+1)
        self.cols = cols      # int32 (nnz)
        self.vals = vals      # fp32  (nnz)
        self.row_offset = row_offset
        self.n = rowptr.numel() - 1
        self.n_total = n_total if n_total is not None else self.n

    @property
    def nnz(self):
        return self.cols.numel()

    def max_degree(self) -> int:
        return int((self.rowptr[1:] - self.rowptr[:-1]).max().item())

    def to_padded(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(values (n, max_deg), indices int64 (n, max_deg)) padded with (0, -1): the layout the
        reference returns (``pack_to_rowwise``, sparse.py:89-135)."""
        width = self.max_degree()
        if self.vals.dtype == torch.float64:   # float64 graph (float64 input): indices by the kernel, values placed by torch
            _, pi = CSRAffinity(self.rowptr, self.cols, self.vals.float(), self.row_offset, self.n_total).to_padded()
            pv = torch.zeros((self.n, width), dtype=torch.float64, device=self.vals.device)
            deg = self.rowptr[1:] - self.rowptr[:-1]
            rows = torch.repeat_interleave(torch.arange(self.n, device=self.vals.device), deg)
            slot = torch.arange(self.nnz, device=self.vals.device) - self.rowptr[:-1][rows]
            pv[rows, slot] = self.vals
            return pv, pi
        pv = torch.empty((self.n, width), dtype=torch.float32, device=self.vals.device)
        pi = torch.empty((self.n, width), dtype=torch.int64, device=self.vals.device)
        _lib.check(
            _lib.lib().tdr_csr_to_padded_f32(
                _lib.ptr(self.rowptr), _lib.ptr(self.cols), _lib.ptr(self.vals), self.n, width, _lib.ptr(pv),
                _lib.ptr(pi), _lib.stream_ptr(),
            ),
            "tdr_csr_to_padded_f32",
        )
        return pv, pi


_MODE = {"sum_minus_prod": 0, "sum": 1}


def symmetrize_to_csr(values, indices, mode="sum_minus_prod", row_offset=0, n_total=None, ext=None, order=None) -> CSRAffinity:
    """``Q = P + P^T - P o P^T`` (or ``P + P^T``) of the row-wise (n, k) block -> CSR.

    ``ext`` = (rows int32 local, cols int32 global, vals fp32): transposed edges received from
    other ranks (multi-GPU path, reference ``sparse.py:209-342``).  ``order``: optional permutation of the local rows
    (position -> row) in which the count pass visits them -- the cluster-sorted order of the kNN search keeps its look-ups of the
    transposed rows inside the L2; the result does not depend on it."""
    if mode not in _MODE:
        raise ValueError(f"Unsupported mode {mode!r}")
    _lib.require_gpu(values, "values")
    L = _lib.lib()
    vals = values.contiguous().float()
    cols = indices.to(torch.int32).contiguous()
    n, k = vals.shape
    if k > 256:
        # the row-local kernels hold up to 256 entries per row; wider blocks (the kNN stage goes to 1024 neighbours) take the
        # sort-and-coalesce form below: device-side torch ops, same values, same column-sorted rows
        return _symmetrize_wide(values, indices, mode, row_offset, n_total if n_total else n, ext)
    dev = vals.device
    ws_bytes = L.tdr_sym_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    if ext is not None and ext[0].numel() > 0:
        er, ec, ev = (ext[0].to(torch.int32).contiguous(), ext[1].to(torch.int32).contiguous(),
                      ext[2].float().contiguous())
        n_ext = er.numel()
    else:
        er = ec = ev = None
        n_ext = 0
    st = _lib.stream_ptr()
    if order is not None:
        order = order.to(torch.int32).contiguous()
        if order.numel() != n:
            order = None
    _lib.check(
        L.tdr_sym_count_ordered_f32(_lib.ptr(vals), _lib.ptr(cols), n, k, row_offset, _lib.ptr(er), _lib.ptr(ec), n_ext,
                                    _lib.ptr(order), _lib.ptr(ws), ws_bytes, _lib.ptr(rowptr), st),
        "tdr_sym_count_ordered_f32",
    )
    nnz = int(rowptr[n].item())  # the one host sync (reference: sparse.py:119 `.max().item()`)
    tcols = torch.empty(nnz, dtype=torch.int32, device=dev)
    tvals = torch.empty(nnz, dtype=torch.float32, device=dev)
    ocols = torch.empty(nnz, dtype=torch.int32, device=dev)
    ovals = torch.empty(nnz, dtype=torch.float32, device=dev)
    _lib.check(
        L.tdr_sym_fill_ordered_f32(n, k, row_offset, _MODE[mode], _lib.ptr(er), _lib.ptr(ec), _lib.ptr(ev), n_ext, _lib.ptr(order),
                                   _lib.ptr(ws), _lib.ptr(rowptr), _lib.ptr(tcols), _lib.ptr(tvals), _lib.ptr(ocols),
                                   _lib.ptr(ovals), st),
        "tdr_sym_fill_ordered_f32",
    )
    if values.dtype == torch.float64:
        # float64 input: the pattern above came from float(values); the float64 values P + P^T - P o P^T are evaluated on
        # it from the float64 block (tdr_sym_values_f64).  Row-sharded: the transposed entries received from the other ranks
        # (float64, parallel.exchange_transposed_edges) enter as a CSR over the local rows (tdr_sym_values_ext_f64).
        v64 = torch.empty(nnz, dtype=torch.float64, device=dev)
        if n_ext:
            if ext[2].dtype != torch.float64:
                raise ValueError("[torchdr_amd] float64 symmetrisation needs the transposed entries in float64.")
            order = torch.argsort(er.long(), stable=True)
            e_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            e_ptr[1:] = torch.bincount(er.long(), minlength=n).cumsum(0)
            e_col, e_val = ec[order].contiguous(), ext[2][order].contiguous()
            _lib.check(L.tdr_sym_values_ext_f64(_lib.ptr(rowptr), _lib.ptr(ocols), n, _lib.ptr(cols), _lib.ptr(values.contiguous()), k,
                                                row_offset, _MODE[mode], _lib.ptr(e_ptr), _lib.ptr(e_col), _lib.ptr(e_val), _lib.ptr(v64), st),
                       "tdr_sym_values_ext_f64")
        else:
            _lib.check(L.tdr_sym_values_f64(_lib.ptr(rowptr), _lib.ptr(ocols), n, _lib.ptr(cols), _lib.ptr(values.contiguous()), k, row_offset,
                                            _MODE[mode], _lib.ptr(v64), st), "tdr_sym_values_f64")
        ovals = v64
    return CSRAffinity(rowptr, ocols, ovals, row_offset=row_offset, n_total=n_total if n_total else n)


def _symmetrize_wide(values, indices, mode, row_offset, n_total, ext) -> CSRAffinity:
    """``symmetrize_to_csr`` for blocks wider than the row-local kernels' 256 entries per row (sparse.py:7-206 semantics): the
    directed entries (i, j, P_ij) and their transposes (rows of this rank only; `ext` = transposes received from other ranks)
    are keyed by (row, column), coalesced with one device sort, and combined as P + P^T (- P o P^T).  torch ops on the
    device, in the values' dtype; rows come out sorted by column like the kernels'."""
    dev = values.device
    n, k = values.shape
    vals = values.contiguous()
    if vals.dtype not in (torch.float32, torch.float64):
        vals = vals.float()
    cols = indices.to(torch.int64)
    rows = (torch.arange(n, device=dev, dtype=torch.int64) + row_offset)[:, None].expand(n, k)
    ok = cols >= 0
    r_a, c_a, v_a = rows[ok], cols[ok], vals[ok]
    local = (c_a >= row_offset) & (c_a < row_offset + n)
    r_t, c_t, v_t = c_a[local], r_a[local], v_a[local]
    if ext is not None and ext[0].numel() > 0:
        r_t = torch.cat([r_t, ext[0].to(torch.int64) + row_offset])
        c_t = torch.cat([c_t, ext[1].to(torch.int64)])
        v_t = torch.cat([v_t, ext[2].to(vals.dtype)])
    key = torch.cat([r_a * n_total + c_a, r_t * n_total + c_t])
    uniq, inv = torch.unique(key, return_inverse=True)        # sorted: (row, column) order
    a = torch.zeros(uniq.numel(), dtype=vals.dtype, device=dev).index_add_(0, inv[: r_a.numel()], v_a)
    b = torch.zeros(uniq.numel(), dtype=vals.dtype, device=dev).index_add_(0, inv[r_a.numel():], v_t)
    out = a + b - a * b if _MODE[mode] == 0 else a + b
    r_u = torch.div(uniq, n_total, rounding_mode="floor") - row_offset
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.bincount(r_u, minlength=n).cumsum(0)
    return CSRAffinity(rowptr, (uniq - (r_u + row_offset) * n_total).to(torch.int32), out, row_offset=row_offset, n_total=n_total)


def symmetrize_sparse(values, indices, mode="sum_minus_prod"):
    """Drop-in for ``torchdr.utils.sparse.symmetrize_sparse`` (sparse.py:170-206): padded output."""
    return symmetrize_to_csr(values, indices, mode).to_padded()


def distributed_symmetrize_sparse(values, indices, chunk_start: int, chunk_size: int, n_total: int, mode="sum_minus_prod"):
    """Drop-in for ``torchdr.utils.sparse.distributed_symmetrize_sparse`` (sparse.py:209-342): this rank's (chunk_size, k)
    block -> its rows of ``P + P^T - P o P^T`` (or ``P + P^T``), padded.  Transposed edges whose row lives on another rank
    travel by one all-to-all-v (``parallel.exchange_transposed_edges``) and enter the symmetrisation kernels as extra
    edges -- the composition ``UMAPAffinity`` uses on row-sharded runs."""
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("distributed_symmetrize requires torch.distributed to be initialized")
    if values.shape[0] != chunk_size:
        raise ValueError(f"[TorchDR] ERROR : values has {values.shape[0]} rows, chunk_size is {chunk_size}.")
    from torchdr_amd.parallel import exchange_transposed_edges

    ext = exchange_transposed_edges(values, indices, chunk_start, n_total, dist.get_world_size())
    return symmetrize_to_csr(values, indices, mode, row_offset=chunk_start, n_total=n_total, ext=ext).to_padded()
