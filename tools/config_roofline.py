"""Bench lines for BASELINE configs C3 and C5 in bench.py's JSON shape (one line each), with a `roofline` object for the
dominant kernel measured by HIP events on the launch stream inside the run:

    python tools/config_roofline.py c3     # LargeVis N = 1M, D = 128, kNN width 15 (perplexity 5), 500 iterations
    python tools/config_roofline.py c5     # symmetric entropic affinity N = 200k, D = 64, perplexity 30: dual iterations

C3, `tdr::ne_grad_kernel` through `tdr_ne_grad_perm_f32` (one launch per iteration): algorithmic bytes per SURVEY.md section 8d K6 =
N k (4 idx + 4 P + 8 z_j + 8 far-endpoint update) + N n_neg (8 + 8) + 2 N (8 z + 8 momentum) = 0.472 GB at k = 15,
against the 8 TB/s HBM peak.  C5, `tdr::pair_scan_kernel<.., SeaStats>` (one launch per dual iteration): 2 N^2 D flop
(5.12e12) against the fp32 matrix peak 157.3 TFLOP/s; the N^2 = 4e10 exponentials are reported beside it (`exp_per_s`).
The kernels are found by wrapping the C-ABI entry points of the loaded library with an event pair (nothing else changes).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.conftest import gmm  # noqa: E402
import torchdr_amd as t  # noqa: E402
from torchdr_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3


class Timed:
    """Wrap one entry point of the ctypes library with HIP events on the current stream (every `every`-th call)."""

    def __init__(self, name, every=1):
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


def line(metric, value, unit, ms, steps, workload, roof, extra=None):
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": workload, "parallelism": "1 GPU"}, "roofline": roof}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def c3(width=15, steps=2):
    n, d, iters, n_neg = 1_000_000, 128, 500, 5
    X = gmm(n, d, 2.0).cuda()
    perp = width // 3
    t.LargeVis(perplexity=perp, max_iter=20, random_state=0).fit_transform(X)   # warm-up
    tm = Timed("tdr_ne_grad_perm_f32", every=10)     # one GPU: permutation sampler, both shares of every negative pair pulled
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.LargeVis(perplexity=perp, max_iter=iters, random_state=0).fit_transform(X)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms, cnt = tm.close()
    nbytes = n * width * (4 + 4 + 8 + 8) + n * n_neg * (8 + 8) + 2 * n * (8 + 8)
    gbs = nbytes / (ms * 1e-3) / 1e9
    line(f"samples/sec (fit_transform), LargeVis N=1M D=128 kNN width {width}, {iters} iterations", n / wall, "samples/sec", wall * 1e3, steps,
         f"BASELINE config C3: LargeVis fit_transform N={n} D={d} perplexity={perp} (kNN width {width}) n_negatives={n_neg} "
         f"max_iter={iters}, Gaussian mixture (1000 clusters, centre scale 2, sigma 0.5, seed 42)",
         {"kernel": "tdr::ne_grad_kernel<2,16> (kind 0: LargeVis attraction and the 5 negatives per row, both endpoints' shares pulled: permutation sampler, no far-endpoint atomics), one launch per iteration",
          "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
          "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "launches_sampled": cnt,
          "note": "SURVEY 8d K6 bytes; traffic: see profiles/r03_c3_pmc.json (separate rocprofv3 --pmc passes)"},
         {"loop_ms_per_fit": ms * iters})


def c5(steps=12):
    n, d = 200_000, 64
    X = gmm(n, d, 2.0).cuda()
    sea = t.SymmetricEntropicAffinity(perplexity=30, lr=1e-1, max_iter=2, zero_diag=False, verbose=False)
    sea.fit_duals(X)    # warm-up
    tm = Timed("tdr_sea_rowstats_f32")
    sea = t.SymmetricEntropicAffinity(perplexity=30, lr=1e-1, max_iter=steps, tol=0.0, zero_diag=False, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sea.fit_duals(X)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, cnt = tm.close()
    flops = 2.0 * n * n * d
    tf = flops / (ms * 1e-3) / 1e12
    line("dual iterations/sec, symmetric entropic affinity N=200k D=64 perplexity 30", cnt / wall, "iterations/sec", wall / max(cnt, 1) * 1e3, cnt,
         f"BASELINE config C5 (input affinity of TSNEkhorn): SymmetricEntropicAffinity.fit_duals N={n} D={d} perplexity=30, Adam lr 0.1, "
         f"{cnt} dual iterations, matrix-free (nothing of size N^2 exists)",
         {"kernel": "tdr::pair_scan_kernel<KQ=8, SeaStats> (fp32 MFMA distance tiles + streaming row statistics: row sum and entropy), one launch per dual iteration",
          "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
          "algorithmic_flops_per_launch": flops, "avg_launch_ms": ms, "launches_sampled": cnt, "exp_per_s": n * float(n) / (ms * 1e-3),
          "note": "SURVEY 8d K7: 2 N^2 D flop on the fp32 matrix pipe + N^2 exponentials; traffic: profiles/r03_c5_pmc.json"})


if __name__ == "__main__":
    for w in sys.argv[1:] or ["c3", "c5"]:
        if w == "c3":
            c3()
        elif w == "c3b":
            c3(width=45, steps=1)
        elif w == "c5":
            c5()
