"""ctypes binding of the C-ABI HIP library (``libtorchdr_amd.so``).

The library is the product: every numeric step of the hot path runs in it.  There is NO
CPU fallback -- if the shared object is missing or a GPU tensor is not supplied, the call
fails loudly (``RuntimeError``), as required for a drop-in whose parity claims rest on the
HIP path being the one that ran.

Entry points are declared in ``include/torchdr_amd.h``.
"""

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtorchdr_amd.so")

_lib = None

c_i32 = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_u64 = ctypes.c_uint64
c_ptr = ctypes.c_void_p

TDR_ERRORS = {
    -1: "bad argument",
    -2: "unsupported configuration",
    -3: "workspace too small",
}


def _declare(lib):
    def sig(name, restype, *argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("tdr_packed_floats", c_i64, c_i64, c_i32)
    sig("tdr_pack_rows_f32", c_i32, c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_ptr)
    sig("tdr_knn_workspace_bytes", c_i64, c_i64, c_i64, c_i32)
    sig("tdr_knn_max_k", c_i32, c_i32)
    sig(
        "tdr_knn_packed_f32", c_i32,
        c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
    )
    sig(
        "tdr_dense_dist_packed_f32", c_i32,
        c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i32, c_i32, c_i32, c_f32, c_ptr, c_i64, c_ptr,
    )
    for name, args in _OPTIONAL_SIGS.items():
        if hasattr(lib, name):
            sig(name, args[0], *args[1:])


# Filled by the modules that own the corresponding kernels (kept here so that the loader has
# one table of every exported symbol; tests check it against include/torchdr_amd.h).
_OPTIONAL_SIGS = {
    "tdr_indexed_sqdist_f32": (c_i32, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr),
    "tdr_umap_search_f32": (c_i32, c_ptr, c_i64, c_i32, c_f32, c_i32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr),
    "tdr_entropic_search_f32": (
        c_i32, c_ptr, c_i64, c_i32, c_f32, c_i32, c_f32, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr,
    ),
    "tdr_sym_workspace_bytes": (c_i64, c_i64, c_i32),
    "tdr_sym_count_f32": (c_i32, c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr),
    "tdr_sym_fill_f32": (c_i32, c_ptr, c_i64, c_i32, c_i64, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr),
    "tdr_csr_to_padded_f32": (c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr),
    "tdr_umap_prepare_f32": (c_i32, c_ptr, c_i64, c_f32, c_f32, c_ptr, c_ptr, c_ptr),
    "tdr_umap_grad_f32": (
        c_i32, c_ptr, c_i32, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_f32, c_f32, c_f32,
        c_i32, c_i32, c_ptr, c_u64, c_f32, c_f32, c_ptr, c_ptr,
    ),
    "tdr_sgd_step_f32": (c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_i32, c_ptr, c_ptr, c_ptr),
    "tdr_ne_grad_f32": (
        c_i32, c_ptr, c_i32, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_f32, c_i32, c_ptr,
        c_u64, c_i32, c_ptr, c_ptr, c_ptr,
    ),
    "tdr_tsne_repulsion_f32": (c_i32, c_ptr, c_i32, c_i64, c_i64, c_i64, c_f32, c_ptr, c_ptr, c_ptr, c_ptr),
    "tdr_fill_f32": (c_i32, c_ptr, c_i64, c_f32, c_ptr),
}


def lib():
    """Return the loaded C-ABI library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"[torchdr_amd] HIP extension not built: {LIB_PATH} is missing. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
            )
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(status, what):
    if status == 0:
        return
    if status < 0:
        raise RuntimeError(f"[torchdr_amd] {what}: {TDR_ERRORS.get(status, status)} (status {status})")
    raise RuntimeError(f"[torchdr_amd] {what}: HIP error {status}")


def require_gpu(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"[torchdr_amd] {name} must be a tensor on a HIP device (got "
            f"{getattr(t, 'device', type(t))}); this build has no CPU path."
        )


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
