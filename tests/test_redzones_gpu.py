"""Out-of-bounds WRITE check for the two kernels the fit lives in (sanitizer coverage).  Device-side
AddressSanitizer builds of csrc/tdr_umap_sched.hip / csrc/tdr_knn_screen.hip exist (tools/build_asan.sh, tools/asan_smoke.sh) but
do not start on the pool's boxes (the ASAN runtime's interceptor of hsa_amd_memory_pool_allocate fails at start-up: XNACK page-fault
retry is not available there), so every output and scratch buffer of those kernels is carved out of an arena with 64 KiB red zones
of a known pattern on both sides, the kernels run at sizes that exercise their tails (ragged last tile / block / window, a hub row,
empty rows, several slices), and the red zones must come back untouched."""

import pytest
import torch

from tests.conftest import gmm

pytestmark = pytest.mark.gpu

RZ = 16384          # red-zone elements of 4 bytes on each side
PATTERN = 0x5A5AA5A5


class Arena:
    def __init__(self):
        self.zones = []

    def empty(self, shape, dtype):
        numel = 1
        for s in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            numel *= int(s)
        words = (numel * torch.empty(0, dtype=dtype).element_size() + 3) // 4
        buf = torch.full((2 * RZ + words + 4,), PATTERN, dtype=torch.int32, device="cuda")
        self.zones.append((buf, words))
        body = buf[RZ: RZ + words]
        return body.view(torch.uint8)[: numel * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(shape)

    def check(self):
        torch.cuda.synchronize()
        for buf, words in self.zones:
            assert bool((buf[:RZ] == PATTERN).all()), "write below a buffer"
            assert bool((buf[RZ + words + 4:] == PATTERN).all()), "write above a buffer"


@pytest.mark.parametrize("S", [1, 2, 4])
def test_schedule_and_gradient_kernels_stay_inside_their_buffers(S):
    from tests.test_umap_sched_gpu import Sched, layout, prepare, random_graph
    from torchdr_amd import _lib

    n = 3000 + 37                      # ragged last schedule block
    rowptr, cols, vals = random_graph(n, seed=11, hub=900)
    eps_csr, _ = prepare(vals.cuda(), 200)
    cols_p, eps_p = layout(rowptr.cuda(), cols.cuda(), eps_csr)
    sc = Sched(rowptr.cuda(), cols_p, eps_p, n, 32, S)
    A = Arena()
    lst, hdr, acc = A.empty(sc.list.shape, torch.int32), A.empty(sc.hdr.shape, torch.int32), A.empty(sc.acc.shape, torch.float32)
    lst.fill_(-7)
    hdr.zero_()
    sc.list, sc.hdr, sc.acc = lst, hdr, acc
    nxt = A.empty(eps_p.shape, torch.float32)
    nxt.copy_(eps_p)
    Z = (torch.randn(n, 2, device="cuda") * 3).contiguous()
    for t0, B in ((0, 32), (32, 32), (64, 7)):       # full windows and a partial last one
        sc.build(nxt, t0, B)
        for t in range(0, B, 3):
            for geom in (0, 16 | 64):
                g = sc.grad(Z, t, t0 + t, 1.577, 0.895, 150, neg=None, seed=5, geom=geom)
                assert bool(torch.isfinite(g).all())
    A.check()


@pytest.mark.parametrize("prune", ["0", "force"])
def test_two_stage_search_stays_inside_its_buffers(prune):
    """tdr_knn_screen_f32 / tdr_knn_screen_clustered_f32 + rescoring on ragged sizes, through the host path with its outputs and
    workspaces replaced by red-zoned ones (torch.empty patched for the duration of the call)."""
    from torchdr_amd import config
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.distance import pairwise_distances

    X = gmm(7001, 40, 2.0, seed=9).cuda()          # 7001 rows: ragged last tile; D = 40: padded feature slice
    with config.options(SCREEN_MODE="0"):
        Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=17, exclude_diag=True, return_indices=True)
    A = Arena()
    real_empty = torch.empty

    def guarded_empty(*shape, **kw):
        dev = kw.get("device")
        if dev is not None and torch.device(dev).type == "cuda" and kw.get("dtype") in (torch.float32, torch.int32, torch.int64):
            shp = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
            return A.empty(tuple(int(s) for s in shp), kw["dtype"])
        return real_empty(*shape, **kw)

    torch.empty = guarded_empty
    try:
        with config.options(SCREEN_MODE="force", PRUNE_MODE=prune):
            C, I = pairwise_distances(X, metric="sqeuclidean", k=17, exclude_diag=True, return_indices=True)
    finally:
        torch.empty = real_empty
    assert dbase.LAST_KNN["path"].startswith("screen")
    assert torch.equal(C, Ce) and torch.equal(I, Ie)
    assert len(A.zones) > 5
    A.check()
