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


HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "torchdr_amd.h")

_CTYPES = {
    "int": c_i32, "int64_t": c_i64, "float": c_f32, "double": c_f64, "uint64_t": c_u64, "uint32_t": ctypes.c_uint32,
}


def parse_header(path=HEADER_PATH):
    """Prototypes declared in include/torchdr_amd.h -> {name: (restype, [argtypes])}.

    The header is the single source of truth for the C ABI; the ctypes signatures are derived
    from it so they cannot drift."""
    import re

    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(tdr_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",") if x.strip()]:
            if a == "void":
                continue
            if "*" in a:
                argtypes.append(c_ptr)
            else:
                ty = a.replace("const", "").split()[0]
                argtypes.append(_CTYPES[ty])
        protos[name] = (_CTYPES[ret], argtypes)
    return protos


def _declare(lib):
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes


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


def fn(name: str, dtype):
    """Entry point `name`_f32 / `name`_f64 for a tensor dtype (the float64 twins of the embedding loop)."""
    return getattr(lib(), name + ("_f64" if dtype == torch.float64 else "_f32"))


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


class UmapLoopDesc(ctypes.Structure):
    """``tdr_umap_loop_desc`` of include/torchdr_amd.h (field order and types must match the header)."""
    _fields_ = [
        ("Z", c_ptr), ("nc", c_i32), ("n_total", c_i64), ("row0", c_i64), ("n_rows", c_i64),
        ("rowptr", c_ptr), ("cols", c_ptr), ("eps_per", c_ptr), ("next", c_ptr),
        ("blk_base", c_ptr), ("list", c_ptr), ("hdr", c_ptr), ("err", c_ptr), ("acc", c_ptr), ("grad", c_ptr), ("mom_buf", c_ptr),
        ("a", c_f32), ("b", c_f32), ("neg_rate", c_i32), ("n_negatives", c_i32), ("seed", c_u64),
        ("exag", c_f32), ("rep", c_f32), ("eps", c_f32), ("n_slices", c_i32), ("block_iters", c_i32),
        ("lr_table", c_ptr), ("max_iter", c_i32), ("momentum", c_f32), ("first_iter", c_i32), ("check_interval", c_i32),
        ("norm2", c_ptr), ("snap", c_ptr), ("nan_flag", c_ptr), ("scratch", c_ptr), ("gather", c_ptr), ("gather_ctx", c_ptr), ("geom", c_i32),
        ("rs", c_ptr), ("pool", c_i32), ("gather_capturable", c_i32), ("Z_alt", c_ptr),
    ]
