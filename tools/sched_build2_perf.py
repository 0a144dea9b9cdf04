"""GPU probe: schedule build of one window of 32 iterations at N = 1M (production numbering) -- the row-chunk kernel of rounds
2-3 against the group-ordered kernel of round 4 (stage sizes), plus the one-off cost of the group sort.

    python tools/sched_build2_perf.py [N]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from tests.test_umap_sched_gpu import GroupSched, Sched, group, layout, prepare
from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.distance.base import ClusterIndex, PackedPoints

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
ci = ClusterIndex(PackedPoints(X))
perm, inv = ci.perm, ci.inv
rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
torch.cumsum((csr.rowptr[1:] - csr.rowptr[:-1])[perm.long()], 0, out=rowptr[1:])
cols, vals = torch.empty_like(csr.cols), torch.empty_like(csr.vals)
_lib.check(_lib.lib().tdr_csr_permute_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(csr.vals), n, _lib.ptr(perm), _lib.ptr(inv),
                                          _lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr()), "permute")
del X, ci, csr
eps_per, _ = prepare(vals, 1000)
cols, eps_per = layout(rowptr, cols, eps_per)
S = int(_lib.lib().tdr_umap_sched_slices(n, 2))


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return sum(ts[:4]) / 4


if len(sys.argv) > 2 and sys.argv[2] == "--one":     # counters: the grouped build alone, a few launches
    stage = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    gs = GroupSched(rowptr, cols, eps_per, n, 32, S, stage=stage)
    ng = gs.to_group(eps_per)
    for t0 in (0, 32, 64, 96, 128):
        gs.build(ng, t0, 32)
    torch.cuda.synchronize()
    sys.exit(0)

print(json.dumps({"n": n, "nnz": int(cols.numel()), "slices": S, "group_sort_ms": timed(lambda: group(rowptr, cols, eps_per, n, S), 5)}), flush=True)

sc = Sched(rowptr, cols, eps_per, n, 32, S)
nxt = eps_per.clone()
for t0 in (0, 32, 64):
    sc.build(nxt, t0, 32)
snap = nxt.clone()
t_copy = timed(lambda: nxt.copy_(snap))


def rebuild():
    nxt.copy_(snap)
    sc.build(nxt, 96, 32)


print(json.dumps({"kernel": "row-chunk (r03)", "build_ms": timed(rebuild) - t_copy}), flush=True)
want = nxt.clone()
hdr_rc = sc.hdr.clone()
del sc

for stage in (0, 512, 768, 1024, 768 | 4 << 16):
    gs = GroupSched(rowptr, cols, eps_per, n, 32, S, stage=stage)
    ng = gs.to_group(eps_per)
    for t0 in (0, 32, 64):
        gs.build(ng, t0, 32)
    snap_g = ng.clone()

    def rebuild_g():
        ng.copy_(snap_g)
        gs.build(ng, 96, 32)

    ms = timed(rebuild_g) - t_copy
    ok = bool(torch.equal(gs.to_rows(ng), want)) and bool(torch.equal(gs.hdr[:, 1], hdr_rc[:, 1]))
    print(json.dumps({"kernel": "group-ordered (r04)", "stage": stage & 0xffff, "waves_variant": (stage >> 16) & 15, "tc4": stage >> 20, "build_ms": ms, "per_iteration_us": ms / 32 * 1e3,
                      "same_counters_and_records": ok, "list_capacity": int(gs.blk_base[-1])}), flush=True)
    del gs
