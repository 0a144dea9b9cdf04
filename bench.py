"""Headline benchmark: UMAP fit_transform samples/sec + kNN-graph build seconds, N=1M D=128 k=30
(BASELINE.json `metric`), synthetic Gaussian-mixture data, on N GPUs of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1          # starts its own 8 ranks (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 --steps 3 --warmup 1   # ... or joins the ranks torchrun started

``--gpus N`` with no rank in the environment re-executes this script N times, one process per GPU (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set as torchrun would), backend ``nccl`` (= RCCL) when the node has N devices; with fewer
devices than ranks (a one-GPU development box) the ranks share devices over ``gloo`` and the line says so
(``"devices_shared": true`` -- control flow only, the timing means nothing then).

A "step" = one complete ``UMAP(...).fit_transform(X)`` (dedup + isfinite scan + pack + exact kNN + sigma
search + symmetrisation + PCA init + ``max_iter`` optimisation iterations) over the SAME N points, with X
already resident in HBM when the timed region starts.  Multi-GPU runs shard the rows of the same
N-point problem (strong scaling): every rank holds X, searches its row chunk against the full
database, exchanges transposed edges (all-to-all-v) and all-gathers the updated rows every iteration.  At N > 1 every
rank holds ITS ROW SHARD of X when the timed region starts (``sharded_input=True``; ``--replicated-input`` hands every
rank the full block instead) and the line carries ``phases_ms`` (shard gather, kNN pieces, exchanges, symmetrisation,
init, loop -- HIP events, max over ranks), ``allgather_us`` (one per-iteration row exchange, timed on its own after the
run), ``rccl_context`` and ``hbm_peak_gb`` per rank.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel group of the step, measured with HIP events inside the timed region: the UMAP
                  gradient evaluation (positive pass + L2-sliced negative passes; algorithmic HBM bytes per
                  evaluation vs 8 TB/s) when the loop outweighs the kNN build, else the kNN scan (algorithmic flops
                  2*n_q*N*D per launch vs the matrix peak of the pipe it runs on); the other one is reported as
                  roofline_secondary;
  cpu_baseline -- the CPU oracle (row-chunked torch/MKL restatement of the reference's backend=None
                  path, oracle/ref_torch.py) timed on this box's host cores on a bounded sample.
"""

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 matrix peak (v_mfma_f32_32x32x16_f16)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def gmm(n, d, scale, seed=42):
    """benchmarks/faiss/run_benchmark.py:127-146 shape: min(1000, n//100) clusters, centres*scale, sigma .5"""
    g = torch.Generator().manual_seed(seed)
    nc = max(1, min(1000, n // 100))
    centers = torch.randn(nc, d, generator=g) * scale
    labels = torch.arange(n) % nc
    return (centers[labels] + 0.5 * torch.randn(n, d, generator=g)).contiguous()


def cpu_baseline(X_cpu, k, max_iter, model, budget_s=600.0):
    """CPU oracle on a bounded sample of the same workload (kind "port"): every stage of the fit,
    each on the sample stated in `sample`, each extrapolated by the printed factor.  The kNN and loop samples grow until
    section 8d's size (16 chunks of 4096 query rows; 20 iterations on 200k rows) or the time budget is reached."""
    from oracle import ref_torch as R

    n, d = X_cpu.shape
    threads = torch.get_num_threads()
    notes = []
    # (1) kNN: row chunks of 4096 queries against the full database
    t_used, rows = 0.0, 0
    while rows < min(n, 16 * 4096) and (rows == 0 or t_used < budget_s):
        t0 = time.perf_counter()
        R.knn_chunked(X_cpu, k, "sqeuclidean", True, chunk=4096, rows=(rows, min(rows + 4096, n)))
        t_used += time.perf_counter() - t0
        rows = min(rows + 4096, n)
    f_knn = n / rows
    t_knn = t_used * f_knn
    notes.append(f"kNN: {rows // 4096} chunks x 4096 query rows vs the full database {t_used:.1f}s, x{f_knn:.1f} -> {t_knn:.0f}s")
    # (2) sigma search + symmetrisation on a row sample of the GPU-built kNN distances is not available here (the fit
    #     keeps the graph, not the distances): both run on the oracle's own kNN of a 20k-row subproblem, scaled by rows
    sub = min(n, 20000)
    Cs, Is = R.knn_chunked(X_cpu[:sub].contiguous(), k, "sqeuclidean", True)
    t0 = time.perf_counter()
    _, _, P = R.umap_affinity(Cs, k, 100)
    t_sig_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    R.symmetrize_sparse(P, Is.long(), "sum_minus_prod")
    t_sym_s = time.perf_counter() - t0
    t_sig, t_sym = t_sig_s * n / sub, t_sym_s * n / sub
    notes.append(f"sigma search / symmetrisation: {sub} rows {t_sig_s:.2f}s / {t_sym_s:.2f}s, x{n / sub:.0f} -> {t_sig:.0f}s / {t_sym:.0f}s")
    # (3) PCA initialisation (spectral_embedding/pca.py:151-184: thin SVD of the centred block), 200k rows
    ps = min(n, 200000)
    t0 = time.perf_counter()
    Xc = X_cpu[:ps] - X_cpu[:ps].mean(0, keepdim=True)
    torch.linalg.svd(Xc, full_matrices=False)
    t_pca_s = time.perf_counter() - t0
    t_pca = t_pca_s * n / ps
    notes.append(f"PCA init: SVD of {ps} rows {t_pca_s:.1f}s, x{n / ps:.0f} -> {t_pca:.0f}s")
    # (4) optimisation loop: the oracle's padded-gather step on a row sample of the GPU-built graph
    csr = model["csr"]
    sample = min(csr.n, 200000)
    rp = csr.rowptr[: sample + 1].cpu()
    deg = (rp[1:] - rp[:-1])
    width = int(deg.max())
    cols = csr.cols[: int(rp[-1])].cpu().long()
    vals = csr.vals[: int(rp[-1])].cpu()
    NN = torch.full((sample, width), -1, dtype=torch.long)
    A = torch.zeros((sample, width))
    r_idx = torch.repeat_interleave(torch.arange(sample), deg)
    slot = torch.arange(cols.numel()) - rp[:-1][r_idx]
    NN[r_idx, slot] = cols
    A[r_idx, slot] = vals
    eps_per, nxt = R.umap_prepare(A, max_iter)
    Z = torch.randn(n, 2) * 1e-4
    rows_t = torch.arange(sample)
    iters, t_used = 0, 0.0
    while iters < 20 and (iters < 3 or t_used < budget_s / 2):
        t0 = time.perf_counter()
        neg = R.sample_negatives(n, rows_t, 150)
        R.umap_gradients(Z, NN, eps_per, nxt, neg, iters, model["a"], model["b"], rows=rows_t)
        t_used += time.perf_counter() - t0
        iters += 1
    f_loop = (n / sample) * max_iter / iters
    t_loop = t_used * f_loop
    notes.append(f"loop: {iters} iterations on {sample} rows (padded width {width}, 150 negatives) {t_used:.1f}s, "
                 f"x{n / sample:.0f} rows x{max_iter / iters:.0f} iterations -> {t_loop:.0f}s")
    total = t_knn + t_sig + t_sym + t_pca + t_loop
    return {
        "value": n / total, "unit": "samples/sec", "cores": threads, "kind": "port",
        "sample": "; ".join(notes) + f"; every factor is a linear extrapolation; {threads} torch threads; SURVEY 8d's sample is 16 kNN "
                  f"chunks and 20 loop iterations: {'complete' if rows >= min(n, 16 * 4096) and iters >= 20 else 'cut short by --cpu-budget'} "
                  f"(--cpu-budget = {budget_s:.0f} s is a ceiling on the kNN sample, half of it on the loop sample)",
        "knn_build_sec_est": t_knn, "total_sec_est": total,
    }


def knn_variants(X, args, dbase):
    """Context for `knn_build_sec` (outside the timed region): how much of it is the data.  The pruned search skips the
    tiles whose cluster balls cannot reach a query's threshold -- on the benchmark's 1000-blob mixture that is ~99.9 % of
    them -- so the same exact search is also timed without pruning (two-stage: fp16-split screening on the f16 matrix
    pipe + exact rescoring), as the one-stage fp32-MFMA scan, and on structureless data of the same shape (centre
    scale 0: one Gaussian), where pruning cannot help."""
    from torchdr_amd.distance import pairwise_distances

    def timed(Xd, prune, screen):
        old = dbase.PRUNE_MODE, dbase.SCREEN_MODE
        dbase.PRUNE_MODE, dbase.SCREEN_MODE = prune, screen
        try:
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pairwise_distances(Xd, metric="sqeuclidean", k=args.k, exclude_diag=True, return_indices=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best, dbase.LAST_KNN.get("path") + ("" if not dbase.LAST_KNN.get("flat_terms") else
                                                        " (threshold scan, %d term%s)" % (dbase.LAST_KNN["flat_terms"], "" if dbase.LAST_KNN["flat_terms"] == 1 else "s"))
        finally:
            dbase.PRUNE_MODE, dbase.SCREEN_MODE = old

    flops = 2.0 * args.n * args.n * args.d
    out = {}
    t, path = timed(X, "0", "auto")
    out["knn_unpruned_sec"] = t
    out["knn_unpruned_path"] = path
    out["knn_unpruned_algorithmic_tflops"] = flops / t / 1e12
    out["knn_unpruned_frac_of_f16_peak"] = flops / t / 1e12 / F16_MFMA_PEAK_TFLOPS
    t, path = timed(X, "0", "0")
    out["knn_one_stage_sec"] = t
    out["knn_one_stage_frac_of_fp32_mfma_peak"] = flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS
    X0 = gmm(args.n, args.d, 0.0).to(X.device)
    t, path = timed(X0, "auto", "auto")
    out["knn_structureless_sec"] = t
    out["knn_structureless_path"] = path
    out["note"] = ("wall seconds of pairwise_distances(k=%d) incl. packing and pilots, best of 3, outside the timed region; "
                   "structureless = the same generator with centre scale 0; unpruned searches of this size run as the threshold scan "
                   "(csrc/tdr_knn_flat.hip: seed -> fixed-threshold passes with a select after each -> rescoring; profiles/r05_knn_flat_scan_pmc.json: "
                   "matrix pipe 59 %% busy at the 1.79 GHz the chip sustains under it)" % args.k)
    return out


def knn_uniform(args, dev, dbase):
    """The reference's own "random (uniform)" benchmark set (benchmarks/faiss/run_benchmark.py:143-145: seed 42,
    ``torch.randn(n, d)``, k = 15; published: Faiss Flat 10.16 s on one B200, BENCHMARK_RESULTS.md:22): the exact search
    where no structure helps -- nothing is pruned, the screening kernel scans every tile."""
    from torchdr_amd.distance import pairwise_distances

    torch.manual_seed(42)
    Xu = torch.randn(args.n, args.d).to(dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pairwise_distances(Xu, metric="sqeuclidean", k=15, exclude_diag=True, return_indices=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    flops = 2.0 * args.n * args.n * args.d
    return {"sec": best, "k": 15, "path": dbase.LAST_KNN.get("path"), "tier": dbase.LAST_KNN.get("tier"),
            "threshold_scan_terms": dbase.LAST_KNN.get("flat_terms"), "flagged_rows": dbase.LAST_KNN.get("flagged"),
            "algorithmic_tflops": flops / best / 1e12, "frac_of_f16_peak": flops / best / 1e12 / F16_MFMA_PEAK_TFLOPS,
            "reference_published_sec": 10.16, "reference_hardware": "1x NVIDIA B200, Faiss GpuIndexFlatL2 through torchdr.pairwise_distances",
            "note": "exact kNN of seed-42 randn(N, D), k = 15, wall incl. packing and pilots, best of 3, outside the timed region"}


class _TimedEntry:
    """Wrap one entry point of the ctypes library with HIP events on the current stream (every `every`-th call)."""

    def __init__(self, name, every=1):
        from torchdr_amd import _lib

        self.L, self.name, self.every = _lib.lib(), name, every
        self.fn = getattr(self.L, name)
        self.events, self.calls = [], 0
        setattr(self.L, name, self)

    def __call__(self, *a):
        self.calls += 1
        if self.calls % self.every:
            return self.fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = self.fn(*a)
        e1.record()
        self.events.append((e0, e1))
        return rc

    def close(self):
        setattr(self.L, self.name, self.fn)
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.events]
        return sum(ms) / max(len(ms), 1), len(ms)


def config_c3(dev, width=15, steps=1, iters=500):
    """BASELINE config C3 (outside the timed region): LargeVis N = 1M, D = 128, kNN width 15, 500 iterations.  `roofline` =
    tdr::ne_grad_kernel (one launch per iteration), algorithmic bytes per SURVEY.md section 8d K6 = N k (4 idx + 4 P + 8 z_j
    + 8 far-endpoint update) + N n_neg (8 + 8) + 2 N (8 z + 8 momentum) against the HBM peak.  Both negative samplers are
    timed: the permutation sampler of the one-GPU default (every row the far endpoint of exactly n_neg pairs; pull form, no
    atomics) and the reference's law (independent uniform draws, base.py:628-636; hash sampler + far-endpoint atomics)."""
    import torchdr_amd as t
    from torchdr_amd import config

    n, d, n_neg = 1_000_000, 128, 5
    X = gmm(n, d, 2.0).to(dev)
    perp = width // 3
    nbytes = n * width * (4 + 4 + 8 + 8) + n * n_neg * (8 + 8) + 2 * n * (8 + 8)
    out = {}
    for name, perm, entry in (("permutation", True, "tdr_ne_grad_perm_f32"), ("run-permutation", "runs", "tdr_ne_grad_runs_f32"),
                              ("independent", False, "tdr_ne_grad_f32")):
        with config.options(PERM_NEGATIVES=perm):
            t.LargeVis(perplexity=perp, max_iter=20, random_state=0).fit_transform(X)   # warm-up
            tm = _TimedEntry(entry, every=10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                t.LargeVis(perplexity=perp, max_iter=iters, random_state=0).fit_transform(X)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            ms, cnt = tm.close()
        out[name] = {"ms_per_fit": wall * 1e3, "samples_per_sec": n / wall, "grad_launch_ms": ms, "launches_sampled": cnt,
                     "hbm_gbs": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    from torchdr_amd.neighbor_embedding import base as _nb
    default = {True: "permutation", "runs": "run-permutation", False: "independent"}[_nb.PERM_NEGATIVES]
    best = out[default]
    kern = {"permutation": "tdr::ne_pull4_kernel<2> (4 lanes per row, every negative gathered)",
            "run-permutation": "tdr::ne_pull4_runs_kernel<2> (4 lanes per row, negatives staged into LDS run by run)",
            "independent": "tdr::ne_grad_kernel<2,16> (hash sampler + far-endpoint atomics)"}[default]
    return {
        "workload": f"BASELINE config C3: LargeVis fit_transform N={n} D={d} perplexity={perp} (kNN width {width}) n_negatives={n_neg} "
                    f"max_iter={iters}, Gaussian mixture (1000 clusters, centre scale 2, sigma 0.5, seed 42)",
        "ms": best["ms_per_fit"], "samples_per_sec": best["samples_per_sec"],
        "roofline": {"kernel": kern + " (kind 0: LargeVis attraction + 5 negatives per row), one launch per iteration; the one-GPU default sampler: " + default,
                     "bound": "hbm", "achieved": best["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": best["frac"],
                     "traffic": _committed_traffic("r06_c3_pmc.json", "derived", default + "_hbm_side_traffic_bytes"),
                     "traffic_source": "profiles/r06_c3_pmc.json (2 x FETCH_SIZE + WRITE_SIZE of the default sampler's launch, separate --pmc passes on this round's build; not re-measured here)",
                     "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": best["grad_launch_ms"],
                     "launches_sampled": best["launches_sampled"]},
        "samplers": out,
    }


def _committed_traffic(fname, *path):
    """HBM-side bytes per launch from a PMC summary committed under profiles/ (separate rocprofv3 --pmc passes over the same
    kernel at the same shape in an earlier round; the kernels have not changed since), or None."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", fname)))
        for k in path:
            v = v[k]
        return float(v)
    except Exception:
        return None


def config_c5(dev, steps=12):
    """BASELINE config C5 (outside the timed region): symmetric entropic affinity N = 200k, D = 64, perplexity 30 -- the dual
    iterations of affinity/entropic.py:518-565.  `roofline` = tdr::pair_scan_kernel<8, SeaStats> (one launch per dual
    iteration): 2 N^2 D flop against the fp32 matrix peak; the N^2 exponentials are reported beside it."""
    import torchdr_amd as t

    n, d = 200_000, 64
    X = gmm(n, d, 2.0).to(dev)
    sea = t.SymmetricEntropicAffinity(perplexity=30, lr=1e-1, max_iter=2, zero_diag=False, verbose=False)
    sea.fit_duals(X)    # warm-up
    tm = _TimedEntry("tdr_sea_rowstats_f32")
    sea = t.SymmetricEntropicAffinity(perplexity=30, lr=1e-1, max_iter=steps, tol=0.0, zero_diag=False, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sea.fit_duals(X)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, cnt = tm.close()
    flops = 2.0 * n * n * d
    tf = flops / (ms * 1e-3) / 1e12
    return {
        "workload": f"BASELINE config C5 (input affinity of TSNEkhorn): SymmetricEntropicAffinity.fit_duals N={n} D={d} perplexity=30, "
                    f"Adam lr 0.1, {cnt} dual iterations, matrix-free",
        "ms": wall / max(cnt, 1) * 1e3, "iterations_per_sec": cnt / wall,
        "roofline": {"kernel": "tdr::pair_scan_kernel<KQ=8, SeaStats> (fp32 MFMA distance tiles + streaming row statistics), one launch per dual iteration",
                     "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": _committed_traffic("r05_c5_pmc.json", "derived", "hbm_side_traffic_bytes"),
                     "traffic_source": "profiles/r05_c5_pmc.json (2 x FETCH_SIZE + WRITE_SIZE per dual iteration, separate --pmc passes on the round-5 build; not re-measured here)",
                     "algorithmic_flops_per_launch": flops, "avg_launch_ms": ms, "launches_sampled": cnt,
                     "exp_per_s": n * float(n) / (ms * 1e-3)},
    }


def self_launch(n_ranks):
    """`python bench.py --gpus N` with no rank in the environment: start the N ranks ourselves (what
    `python -m torch.distributed.run --nproc-per-node N` would do), one process per GPU over RCCL.  Rank 0's stdout is this
    process's stdout, so the ONE JSON line comes through unchanged."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    n_dev = torch.cuda.device_count()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n_ranks),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if n_dev < n_ranks:   # fewer devices than ranks: RCCL refuses duplicate devices -> gloo, host-staged collectives
        env["TDR_DIST_BACKEND"] = "gloo"
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", dest="d", type=int, default=128)
    ap.add_argument("--neighbors", dest="k", type=int, default=30)
    ap.add_argument("--max-iter", type=int, default=1000)
    ap.add_argument("--scale", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn-variants", action="store_true",
                    help="skip the untimed kNN context figures (unpruned two-stage, one-stage fp32, structureless data)")
    ap.add_argument("--no-configs", action="store_true", help="skip the untimed BASELINE configs C3 (LargeVis 1M) and C5 (SEA 200k)")
    ap.add_argument("--loop", choices=["auto", "graph", "c", "python"], default="auto",
                    help="UMAP loop driver: replayed HIP graphs (default), plain launches from the C loop object, or one Python iteration per step")
    ap.add_argument("--replicated-input", action="store_true",
                    help="N > 1: every rank is handed the full block (the reference's calling convention) instead of its row shard")
    ap.add_argument("--peer-exchange", action="store_true",
                    help="N > 1: exchange the stepped rows as direct peer writes over HIP IPC (csrc/tdr_peerx.hip) instead of the RCCL "
                         "all-gather; opt-in between distinct devices (never run there), automatic where ranks share a device")
    ap.add_argument("--cpu-budget", type=float, default=600.0,
                    help="ceiling (seconds of CPU work) on the kNN sample of cpu_baseline; SURVEY 8d's 16 chunks need ~215 s and its 20 loop "
                         "iterations ~65 s on the 128-core box, so the default lets both complete")
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="with --world W: run what rank r of a W-rank fit runs, ALONE on this GPU (torchdr_amd/utils/emulation.py; the edge "
                         "exchange served from the other ranks' graphs, the row exchange as a loopback copy) and print its phase split")
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()

    if args.emulate_rank is not None:
        from tools.rank_share import measure

        recs = measure(args.n, args.d, args.k, args.max_iter, [args.world], [str(args.emulate_rank)], scale=args.scale, emit=lambda *a, **k: None)
        print(json.dumps({"emulated_rank_share": recs[-1], "single_process": recs[0]}), flush=True)
        return

    if args.gpus > 1 and "RANK" not in os.environ and "LOCAL_RANK" not in os.environ:
        self_launch(args.gpus)

    from torchdr_amd.distributed import init_from_env

    distributed = init_from_env()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    local_rank = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from torchdr_amd import UMAP
    from torchdr_amd.distance import base as dbase
    from torchdr_amd.neighbor_embedding import umap as umod

    if args.peer_exchange:
        from torchdr_amd.neighbor_embedding import base as nbase

        nbase.PEER_EXCHANGE = True
    if args.loop != "auto":
        umod.LOOP_RUNNER = args.loop != "python"
        umod.LOOP_GRAPH = args.loop == "graph"
    from torchdr_amd.distributed import chunk_bounds
    from torchdr_amd.utils import phases

    X_cpu = gmm(args.n, args.d, args.scale)
    sharded = world > 1 and not args.replicated_input
    if sharded:     # this rank's row shard is what is resident when the clock starts
        s0, s1 = chunk_bounds(args.n, rank, world)
        X = X_cpu[s0:s1].to(dev)
    else:
        X = X_cpu.to(dev)
    devices_shared = world > torch.cuda.device_count()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    keep = {}
    # wall (HIP events) of the whole kNN-graph build = pilots + cluster index + scan + rescoring + any exact fallback
    knn_total_events = []
    _knn_packed = dbase.knn_packed

    def _timed_knn_packed(*a, **kw):
        if dbase.PROFILE is None or not kw.get("_allow_screen", True):
            return _knn_packed(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _knn_packed(*a, **kw)
        e1.record()
        knn_total_events.append((e0, e1))
        return out

    dbase.knn_packed = _timed_knn_packed
    _knn_sharded = dbase.knn_pruned_sharded

    def _timed_knn_sharded(*a, **kw):   # N > 1: pilots + index build + broadcast + pruned scan + row exchange
        if dbase.PROFILE is None:
            return _knn_sharded(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _knn_sharded(*a, **kw)
        e1.record()
        if out is not None:
            knn_total_events.append((e0, e1))
        return out

    dbase.knn_pruned_sharded = _timed_knn_sharded

    def one_step(record):
        dbase.PROFILE = [] if record else None
        umod.PROFILE = [] if record else None
        umod.LOOP_PROFILE = [] if record else None
        m = UMAP(n_neighbors=args.k, max_iter=args.max_iter, random_state=0, backend=None, sharded_input=sharded)
        if record:
            keep["rccl_context"] = None
            _orig = m.clear_memory

            def _keep_then_clear():
                keep["csr"] = m._csr
                keep["a"], keep["b"] = m._a, m._b
                keep["rccl_context"] = getattr(m, "_rccl_ctx", None) is not None
                keep["exchange"] = type(getattr(m, "_rccl_ctx", None)).__name__ if getattr(m, "_rccl_ctx", None) is not None else "torch.distributed"
                keep["loop_order"] = getattr(m, "loop_order_", None) is not None
                _orig()

            m.clear_memory = _keep_then_clear
        Z = m.fit_transform(X)
        return Z

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    knn_events, grad_events, loop_events = [], [], []
    torch.cuda.reset_peak_memory_stats(dev)
    phases.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
        knn_events.extend(dbase.PROFILE)
        grad_events.extend(umod.PROFILE)
        loop_events.extend(umod.LOOP_PROFILE)
    barrier()
    elapsed = time.perf_counter() - t0
    phase_ms = {k_: v / args.steps for k_, v in phases.stop().items()}
    dbase.PROFILE = None
    umod.PROFILE = None
    umod.LOOP_PROFILE = None

    from torchdr_amd.parallel import allreduce_max_

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        allreduce_max_(t)
    elapsed = float(t.item())
    hbm_peak = torch.tensor([0.0] * world, dtype=torch.float64, device=dev)
    hbm_peak[rank] = torch.cuda.max_memory_allocated(dev) / 1e9
    names = sorted(phase_ms)
    ph = torch.tensor([phase_ms[k_] for k_ in names], dtype=torch.float64, device=dev)
    allgather_us = None
    if distributed:
        allreduce_max_(hbm_peak)
        allreduce_max_(ph)      # every rank records the same phase names (same code path)
        # one per-iteration row exchange of the (N, 2) embedding, on its own: the transport the loop used
        from torchdr_amd.parallel import RcclContext, allgather_rows_

        Zt = torch.zeros((args.n, 2), dtype=torch.float32, device=dev)
        c0, c1 = chunk_bounds(args.n, rank, world)
        from torchdr_amd.parallel import PeerExchange

        if keep.get("exchange") == "PeerExchange":      # the transport the loop used
            ctx = PeerExchange.shared(args.n, 2, dev)
        else:
            ctx = RcclContext.shared(args.n, dev) if (keep.get("rccl_context") and dist.get_backend() == "nccl") else None
        reps = 50
        for timed in (False, True):
            barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                if ctx is not None:
                    ctx.allgather_rows_(Zt)
                else:
                    allgather_rows_(Zt, c0, c1 - c0, world)
            torch.cuda.synchronize()
            allgather_us = (time.perf_counter() - t1) / reps * 1e6
        ag = torch.tensor([allgather_us], dtype=torch.float64, device=dev)
        allreduce_max_(ag)
        allgather_us = float(ag.item())
    phase_ms = dict(zip(names, [round(v, 3) for v in ph.tolist()]))

    # ---- kernel-level numbers (HIP events recorded on the launch stream inside the timed region) -------------------
    # (1) kNN build: one event pair around the scan (+ rescoring) launches.  Large searches take the two-stage path
    #     (fp16-split screening on the f16 matrix pipe + exact fp32 rescoring), on clustered data with cluster-bound
    #     pruning, which skips most database tiles: its rate is an EFFECTIVE one (algorithmic flops / time).
    scan_ms = [e[0].elapsed_time(e[1]) for e in knn_events]
    nq = knn_events[0][2] if knn_events else 0
    path = knn_events[0][3] if knn_events else "exact"
    scan_avg_ms = sum(scan_ms) / max(len(scan_ms), 1)
    flops = 2.0 * nq * args.n * args.d
    knn_tflops = flops / (scan_avg_ms * 1e-3) / 1e12 if scan_avg_ms > 0 else 0.0
    tot = [e0.elapsed_time(e1) for (e0, e1) in knn_total_events]
    knn_build_ms = sum(tot) / max(len(tot), 1) if tot else scan_avg_ms
    knn_peak = F16_MFMA_PEAK_TFLOPS if path.startswith("screen") else FP32_MFMA_PEAK_TFLOPS
    # (2) UMAP gradient evaluation (positive pass + negative slice passes), sampled every PROFILE_EVERY iterations.
    #     Algorithmic HBM bytes per evaluation: the CSR stream nnz * (4 col + 4 eps_per +
    #     4 next) + per active edge 4 (next write) + 8 (z_j) + per used negative 8 (z_j) + per row 8 + 8, with the
    #     measured nnz and the expected 8.6 active edges / 43 used negatives per row and iteration.
    #     Scheduled loop: one evaluation = S slice passes of umap_sched_grad_kernel + 1/n_iters of a schedule build
    #     (umap_sched_build_kernel advances the epoch counters 32 iterations at a time); every build is timed.
    #     Loop runner (default): HIP events around every segment of the loop (a segment = the iterations between two
    #     convergence checks, ~check_interval of them, executed as replayed window graphs): one "evaluation" = one whole
    #     iteration = 1/n of its window's schedule build + S gradient passes + the SGD step.
    if loop_events:
        seg = [(e[0].elapsed_time(e[1]), e[2]) for e in loop_events if e[2] > 1]
        grad_avg_ms = sum(t for t, _ in seg) / max(sum(n for _, n in seg), 1)
        grad_only_ms, build_avg_ms, n_sampled = None, None, sum(n for _, n in seg)
        nnz = loop_events[0][3]
    else:
        grad_ms = [e[1].elapsed_time(e[2]) for e in grad_events if e[0] == "grad"]
        build_ms = [e[1].elapsed_time(e[2]) / e[3] for e in grad_events if e[0] == "build"]
        build_avg_ms = sum(build_ms) / max(len(build_ms), 1)      # amortised per iteration
        grad_only_ms = sum(grad_ms) / max(len(grad_ms), 1)
        grad_avg_ms = grad_only_ms + build_avg_ms
        n_sampled = len(grad_ms)
        nnz = next((e[3] for e in grad_events if e[0] == "grad"), 0)
    rows = (args.n + world - 1) // world
    grad_bytes = 12.0 * nnz + rows * (8.6 * 12.0 + 43.0 * 8.0 + 16.0)
    grad_gbs = grad_bytes / (grad_avg_ms * 1e-3) / 1e9 if grad_avg_ms > 0 else 0.0

    def pmc_traffic(name):
        # HBM-side traffic from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs of the same
        # kernels on the same workload; FETCH_SIZE doubled per the gfx950 note of the guide), bytes per launch group
        f = os.path.join(ROOT, "profiles", name)
        if world != 1 or (args.n, args.d, args.k) != (1_000_000, 128, 30) or not os.path.exists(f):
            return None
        try:
            pmc = json.load(open(f))
            return (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        except Exception:
            return None

    knn_pmc = {"screen": "r01_knn_screen_pmc.json", "exact": "r01_knn_exact_scan_pmc.json"}.get(path)
    roof_knn = {
        "kernel": {"exact": "tdr::knn_scan_kernel (fp32 MFMA)"}.get(path, "tdr::scr::knn_screen_kernel (" + path + ") + knn_rescore_kernel"),
        "bound": "mfma", "achieved": knn_tflops, "peak": knn_peak, "unit": "TFLOP/s",
        "frac": None if path.endswith("pruned") else knn_tflops / knn_peak,
        "traffic": pmc_traffic(knn_pmc) if knn_pmc else None,
        "algorithmic_flops_per_launch": flops, "avg_launch_ms": scan_avg_ms,
        "note": ("two-stage exact kNN (fp16-split screening + exact fp32 rescoring; results bit-identical to the one-stage "
                 "fp32-MFMA kernel)" + ("; cluster-bound pruning skips most tiles, so `achieved` is algorithmic flops / time, "
                                        "not matrix-pipe utilisation" if path.endswith("pruned") else "")),
    }
    pool = bool(getattr(umod, "NEGATIVES", "iid") == "pool") and umod.SCHEDULED
    pmc_file = "r06_umap_pool_pmc.json" if pool else ("r04_umap_sched_pmc.json" if umod.SCHEDULED else "r01_umap_grad_pmc.json")
    roof_grad = {
        "kernel": ("one whole UMAP iteration: tdr::umap_pool_grad_kernel<2,512,2,256,8> (fired edges from the per-iteration lists, negatives from a "
                   "per-block LDS pool of 256 uniformly chosen 8-row runs of the embedding; the launch carries torch.optim.SGD's step: two embedding "
                   "buffers) -- HIP events around every 25th iteration's launch -- + 1/32 of tdr::umap_sched_build2_kernel (group-ordered schedule "
                   "build; every build timed)"
                   if pool else
                   "one whole UMAP iteration: tdr::umap_sched_grad_kernel<2,4,false> (ONE launch over both L2 slices of the embedding, "
                   "slices spread over the XCDs) + tdr::umap_sched_combine_sgd_kernel (clamps + SGD step) -- HIP events around "
                   "every 25th iteration's two launches -- + 1/32 of tdr::umap_sched_build2_kernel (group-ordered schedule build; "
                   "every build timed)" if umod.SCHEDULED else
                   "tdr::umap_grad_kernel<2,16,4,true> + 2 x tdr::umap_neg_dense_kernel<2,8,2> (one gradient evaluation)"),
        "bound": "hbm", "achieved": grad_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": grad_gbs / HBM_PEAK_GBS,
        "traffic": pmc_traffic(pmc_file),
        "traffic_source": "profiles/" + pmc_file + ": separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over this bench command on this round's "
                          "build (tools/pmc_bench.sh), per iteration; not re-measured inside this run",
        "algorithmic_bytes_per_launch": grad_bytes, "avg_launch_ms": grad_avg_ms, "evaluations_sampled": n_sampled,
        "grad_passes_ms": grad_only_ms, "schedule_build_ms_per_iteration": build_avg_ms,
        "note": ("algorithmic bytes = SURVEY 8d K5: the per-step edge stream 12 B x nnz + per row 8.6 x 12 + 43 x 8 + 16 B.  The loop reads "
                 "per-iteration firing lists instead of the edge stream and, since round 6, serves the 43 negatives per row from LDS (one "
                 "128-byte line per 16 staged rows instead of one L2 request per negative): it moves a fraction of those bytes (`traffic`), so "
                 "`frac` can exceed 1 -- it is SURVEY 8d's figure, not an HBM pin; what binds the launch is `binding_resource`"
                 if pool else
                 "algorithmic bytes = SURVEY 8d's per-step edge stream (12 B x nnz + gathers); the scheduled loop reads "
                 "per-iteration firing lists instead (~8.6 x 4 B per row), so what bounds the passes is the L2 gather "
                 "request rate of the ~52 random 8-byte reads of Z per row and iteration; see DESIGN.md section 6"),
    }
    if pool and world == 1 and (args.n, args.d, args.k) == (1_000_000, 128, 30):
        try:
            gp = json.load(open(os.path.join(ROOT, "profiles", "r06_pool_pmc.json")))
            # vector ALU busy share of the gradient launch: SQ_ACTIVE_INST_VALU counts 4-cycle quads per SIMD, summed over the chip
            busy = 4.0 * float(gp["SQ_ACTIVE_INST_VALU"]) / (1024.0 * float(gp["GRBM_GUI_ACTIVE"]) / 8.0)
            roof_grad["binding_resource"] = {
                "name": "vector instruction issue of the gradient launch (one lane per row; 5 negatives per fired edge, each ~16 vector "
                        "instructions of which 3 run at quarter rate)",
                "valu_busy_frac": busy, "vector_instructions_per_wavefront": float(gp["SQ_INSTS_VALU"]) / float(gp["SQ_WAVES"]),
                "l2_read_requests_per_launch": float(gp.get("TCP_TCC_READ_REQ_sum", 0.0)),
                "source": "profiles/r06_pool_pmc.json (tools/pmc_pool.sh: separate --pmc passes over the launch at this shape on this round's build)",
            }
        except Exception:
            pass
    # what actually bounds the gradient launch: the scheduled loop gathers ~52 random 8-byte rows of Z per
    # row and iteration out of the XCD's L2; the request count per launch comes from the committed TCP_TCC_READ_REQ pass
    # (same kernel, same shape), the time from this run's HIP events, the ceiling from tools/gather_bench.hip (random 8-byte
    # gathers from an L2-resident 4 MB table, every CU busy: 266-273 G requests/s on this chip, profiles/r04_grad_pmc.json)
    if umod.SCHEDULED and not pool and world == 1 and (args.n, args.d, args.k) == (1_000_000, 128, 30):
        try:
            gp = json.load(open(os.path.join(ROOT, "profiles", "r04_grad_pmc.json")))
            req = float(gp["counters"]["TCP_TCC_READ_REQ_sum"])
            t_grad = (grad_only_ms if grad_only_ms else grad_avg_ms) * 1e-3
            roof_grad["binding_resource"] = {
                "name": "L2 gather requests (TCP->TCC reads) of the gradient launch",
                "requests_per_launch": req, "gather_req_per_s": req / t_grad, "ceiling_req_per_s": 268e9,
                "frac": req / t_grad / 268e9,
                "source": "requests: profiles/r04_grad_pmc.json (TCP_TCC_READ_REQ_sum, separate --pmc pass, same kernel and shape); time: this "
                          "run's HIP events over the iteration (the combine + schedule-build share included, so the rate is a lower "
                          "bound); ceiling: tools/gather_bench.hip, 266-273 G/s measured",
            }
        except Exception:
            pass
    loop_ms = grad_avg_ms * args.max_iter
    dominant, secondary = (roof_grad, roof_knn) if loop_ms >= scan_avg_ms else (roof_knn, roof_grad)

    if rank == 0:
        # The driver keeps the LAST ~2000 characters of the line: everything that is context goes first (and, in full, into the side
        # file named by `detail_file`), the contract keys, both halves of the metric, `roofline` and `cpu_baseline` come last.
        detail = {
            "phases_ms": phase_ms,
            "hbm_peak_gb": [round(v, 2) for v in hbm_peak.tolist()],
            "knn_scan_sec": scan_avg_ms * 1e-3,
            "knn_path": path,
            "roofline_detail": dominant,
            "roofline_secondary": secondary,
        }
        if world > 1:
            detail["backend"] = dist.get_backend()
            detail["devices_shared"] = bool(devices_shared)
            detail["rccl_context"] = bool(keep.get("rccl_context"))
            detail["row_exchange"] = keep.get("exchange")      # PeerExchange (direct peer writes), RcclContext (ring all-gather) or torch.distributed
            detail["loop_in_cluster_order"] = bool(keep.get("loop_order"))
            detail["allgather_us"] = allgather_us
            detail["allgather_ms_per_fit"] = None if allgather_us is None else allgather_us * args.max_iter * 1e-3
        if not args.no_knn_variants and world == 1:
            detail["knn_context"] = knn_variants(X, args, dbase)
            detail["knn_uniform"] = knn_uniform(args, dev, dbase)
        if not args.no_configs and world == 1:
            del X
            torch.cuda.empty_cache()
            detail["configs"] = {"c3": config_c3(dev), "c5": config_c5(dev),
                                 "note": "BASELINE.json configs[2] and configs[4], run once each after the timed region; per-kernel roofline "
                                         "by HIP events around the C-ABI entry point on the launch stream"}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(X_cpu, args.k, args.max_iter, keep, budget_s=args.cpu_budget)
            detail["cpu_baseline_detail"] = dict(cpu)
        detail_file = None
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            detail_file = os.path.join("gpurun_out", "bench_detail.json")
            json.dump(detail, open(os.path.join(ROOT, detail_file), "w"), indent=1)
        except OSError:
            detail_file = None
        out = dict(detail)
        out["detail_file"] = detail_file
        out.update({
            "metric": "samples/sec (fit_transform) + kNN-graph build sec, UMAP N=1M D=128 k=30",
            "value": args.n * args.steps / elapsed,
            "unit": "samples/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"UMAP fit_transform N={args.n} D={args.d} k={args.k} max_iter={args.max_iter} "
                            f"n_components=2 init=pca, Gaussian mixture (min(1000,N/100) clusters, centre scale "
                            f"{args.scale}, sigma 0.5, seed 42)",
                "parallelism": f"rows sharded over {world} GPU(s)" + ("" if world == 1 else
                               (", row-sharded input" if sharded else ", replicated input")),
            },
            "knn_build_sec": knn_build_ms * 1e-3,
            "loop_ms_per_iteration": grad_avg_ms,
            # the contract's compact objects (the long forms are `roofline_detail` / `cpu_baseline_detail` above)
            "roofline": {k_: dominant.get(k_) for k_ in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")}
                        | {"kernel": dominant["kernel"].split(" (")[0][:90],
                           "binding": ({"what": "vector issue", "valu_busy_frac": round(dominant["binding_resource"].get("valu_busy_frac", 0.0), 3)}
                                       if "valu_busy_frac" in dominant.get("binding_resource", {}) else
                                       ({"what": "L2 gather requests", "frac": round(dominant["binding_resource"]["frac"], 3)}
                                        if "binding_resource" in dominant else None))},
        })
        if cpu is not None:
            out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                   "sample": cpu["sample"][:420], "knn_build_sec_est": cpu["knn_build_sec_est"]}
        print(json.dumps(out), flush=True)
    if distributed:
        from torchdr_amd.parallel import RcclContext

        dist.barrier()
        RcclContext.destroy_shared()
        from torchdr_amd.parallel import PeerExchange

        PeerExchange.destroy_shared()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
