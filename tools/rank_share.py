"""One rank's share of a W-rank row-sharded UMAP fit, measured on ONE GPU (torchdr_amd/utils/emulation.py).

    python tools/rank_share.py [--npoints 1000000] [--dim 128] [--worlds 2,4,8] [--ranks first,middle] [--max-iter 1000] [--out FILE]
    python bench.py --emulate-rank 3 --world 8          # the same measurement for one (rank, world), one JSON line

Per (N, D): the single-process fit with its phase split, then for every W the fit of rank r run ALONE -- its pilots, the whole
cluster index, its positions of the pruned scan, bandwidths / symmetrisation / loop layout of its rows, the loop over its N / W rows
against the full replicated embedding through the C loop object -- with the edge exchange served from the other ranks' graphs
(computed beforehand, untimed) and the per-iteration row exchange as a loopback copy of the same bytes.  Link time is NOT in these
numbers: `exchange_bytes` says what would travel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import torchdr_amd
from tests.conftest import gmm
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.utils import phases
from torchdr_amd.utils.emulation import EmulatedRank


def timed_fit(X, k, max_iter, reps=2):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        phases.start()
        t0 = time.perf_counter()
        m = torchdr_amd.UMAP(n_neighbors=k, max_iter=max_iter, random_state=0, backend=None)
        m.fit_transform(X)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ph = phases.stop()
        if best is None or wall < best[0]:
            best = (wall, {k_: round(v, 3) for k_, v in ph.items()}, getattr(m, "row_exchange_", None))
    return best


def measure(n, d, k, max_iter, worlds, which, scale=2.0, emit=print):
    X = gmm(n, d, scale).cuda()
    timed_fit(X, k, min(max_iter, 50), reps=1)   # warm-up (allocator, code objects)
    wall, ph, _ = timed_fit(X, k, max_iter)
    base = {"n": n, "d": d, "k": k, "max_iter": max_iter, "world": 1, "rank": 0, "fit_ms": round(wall, 2), "phases_ms": ph}
    emit(json.dumps(base), flush=True)
    out = [base]
    for W in worlds:
        em = EmulatedRank(W)

        def mk():
            a = UMAPAffinity(n_neighbors=k, max_iter=100)
            a._accept_loop_order = True      # what a row-sharded UMAP tells its affinity (neighbor_embedding/umap.py)
            return a

        t0 = time.perf_counter()
        em.collect(mk, X)
        t_collect = time.perf_counter() - t0
        ranks = sorted({0 if w == "first" else (W // 2 if w == "middle" else (W - 1 if w == "last" else int(w))) for w in which})
        for r in ranks:
            em.prepare(r, n)
            em.enter(r)
            timed_fit(X, k, min(max_iter, 50), reps=1)
            wall, ph, xch = timed_fit(X, k, max_iter)
            rec = {"n": n, "d": d, "k": k, "max_iter": max_iter, "world": W, "rank": r, "fit_ms": round(wall, 2), "phases_ms": ph,
                   "row_exchange": xch, "exchange_bytes": em.exchange_bytes, "edge_exchange_bytes_received": getattr(em, "edge_exchange_bytes", None),
                   "speedup_without_link_time": round(base["fit_ms"] / wall, 3), "collect_other_ranks_sec": round(t_collect, 2)}
            emit(json.dumps(rec), flush=True)
            out.append(rec)
        em.leave()
        del em
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npoints", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--neighbors", type=int, default=30)
    ap.add_argument("--max-iter", type=int, default=1000)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--ranks", default="first,middle")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    f = open(a.out, "a") if a.out else None

    def emit(line, flush=True):
        print(line, flush=True)
        if f:
            f.write(line + "\n")
            f.flush()

    measure(a.npoints, a.dim, a.neighbors, a.max_iter, [int(w) for w in a.worlds.split(",")], a.ranks.split(","), emit=emit)


if __name__ == "__main__":
    main()
