"""GPU probe: sensitivity of the scheduled UMAP gradient launch (production geometry: joint slice launch, rows dealt by load,
N = 1M in the cluster-sorted numbering) to its vector-instruction blocks -- scratch builds with TDR_GRAD_ABLATE switches
(tools/grad_ablate.sh: 1 row key without hash rounds, 2 no binomial split, 4 no pow, 8 one-multiply item hash), each timed in
its own process.

    bash tools/grad_ablate.sh "1 2 4 8 15"     # in the build container
    gpurun -- 'python tools/grad_ablate.py 1 2 4 8 15'
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from torchdr_amd import _lib

    v = int(sys.argv[2])
    if v:
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", f"libtdr_gab{v}.so")
    import torch

    from tests.conftest import gmm
    from tests.test_umap_sched_gpu import Sched, layout, prepare
    from torchdr_amd.affinity import UMAPAffinity
    from torchdr_amd.distance.base import ClusterIndex, PackedPoints

    n = 1_000_000
    X = gmm(n, 128, 2.0).cuda()
    csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
    ci = ClusterIndex(PackedPoints(X))
    perm, inv = ci.perm, ci.inv
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum((csr.rowptr[1:] - csr.rowptr[:-1])[perm.long()], 0, out=rowptr[1:])
    cols, vals = torch.empty_like(csr.cols), torch.empty_like(csr.vals)
    _lib.check(_lib.lib().tdr_csr_permute_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(csr.vals), n, _lib.ptr(perm), _lib.ptr(inv),
                                              _lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr()), "permute")
    del X, ci
    eps_per, _ = prepare(vals, 1000)
    cols, eps_per = layout(rowptr, cols, eps_per)
    sc = Sched(rowptr, cols, eps_per, n, 32, 2)
    nxt = eps_per.clone()
    for t0 in (0, 32, 64, 96):
        sc.build(nxt, t0, 32)
    Z = (torch.randn(n, 2, device="cuda") * 5).contiguous()
    out = {"ablate": v}
    for geom in (16 | 64, 16):
        ts = []
        for i in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sc.grad(Z, i % 32, 96 + i % 32, 1.577, 0.895, 150, neg=None, seed=1234, geom=geom)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        out[f"grad_ms_geom{geom}"] = sum(ts[:20]) / 20
    print(json.dumps(out), flush=True)
else:
    for v in [0] + [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8", "15"])]:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(v)], capture_output=True, text=True)
        line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else json.dumps({"ablate": v, "error": o.stderr[-400:]}), flush=True)
