"""The threshold scan's exactness argument as an executable model (no device): seed -> select -> passes against fixed
thresholds -> select -> rescoring, run in numpy on screening values that differ from the exact distances by at most E, with the
pass plan the C library itself computes (tdr_knn_screen_flat_plan).  What csrc/tdr_knn_flat.hip / tdr_knn_screen.hip claim
(DESIGN section 3, K1f): a query that is neither lost (a pass met more candidates than its region holds) nor flagged (final list full
inside the band) gets exactly the k nearest rows -- whatever the noise, the visiting order and the plan."""

import ctypes

import numpy as np
import pytest


def _plan(n, d, k, terms, LL):
    from torchdr_amd import _lib

    L = _lib.lib()
    bounds = (ctypes.c_int32 * 40)()
    stride = ctypes.c_int32(0)
    nb = L.tdr_knn_screen_flat_plan(n, n, d, k, terms, LL, bounds, 40, ctypes.byref(stride))
    assert nb >= 2
    return [int(bounds[i]) for i in range(nb)], int(stride.value)


def _model(D, A, k, LL, E, bounds, stride, cap=256):
    """D exact / A screening values (nq, n), |A - D| <= E.  Returns (indices of the k smallest D among the rescored candidates,
    flagged) per query, following the device pipeline step by step."""
    nq, n = D.shape
    n_tiles = (n + 31) // 32
    order = (np.arange(n_tiles, dtype=np.int64) * stride) % n_tiles          # position -> tile
    assert len(np.unique(order)) == n_tiles

    def rows_of(p0, p1):
        t = order[p0:p1]
        r = (t[:, None] * 32 + np.arange(32)[None, :]).reshape(-1)
        return r[r < n]

    out = np.full((nq, k), -1, dtype=np.int64)
    flagged = np.zeros(nq, dtype=bool)
    for q in range(nq):
        seen = rows_of(0, bounds[0])
        lst = seen[np.argsort(A[q, seen], kind="stable")][:LL]               # select after the seed: the L smallest, ascending
        lost = False
        for p0, p1 in zip(bounds, bounds[1:]):
            a_k = A[q, lst[k - 1]]
            tau = a_k + 2 * E
            if len(lst) == LL:
                tau = min(tau, A[q, lst[-1]])
            r = rows_of(p0, p1)
            surv = r[A[q, r] <= tau]
            if len(surv) > cap:
                lost = True
                surv = surv[:cap]
            cand = np.concatenate([lst, surv])
            lst = cand[np.argsort(A[q, cand], kind="stable")][:LL]
        a_k = A[q, lst[k - 1]]
        full_in_band = len(lst) == LL and A[q, lst[-1]] <= a_k + 2 * E
        flagged[q] = lost or full_in_band
        band = lst[A[q, lst] <= a_k + 2 * E]                                   # rescoring: exact distances of the band's candidates
        out[q] = band[np.argsort(D[q, band], kind="stable")][:k]
    return out, flagged


@pytest.mark.parametrize("k,LL,E_rel,terms", [(15, 64, 0.002, 1), (30, 128, 0.01, 1), (30, 128, 0.0005, 3), (100, 128, 0.001, 2)])
def test_threshold_scan_model_returns_the_exact_neighbours(k, LL, E_rel, terms):
    rng = np.random.default_rng(7 + k)
    n, d, nq = 131_072 + 17, 24, 48
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[: n // 2] += 3.0 * rng.standard_normal((1, d)).astype(np.float32)       # two populations: thresholds differ between queries
    Q = X[rng.choice(n, nq, replace=False)]
    D = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    E = E_rel * float(np.median(np.sort(D, axis=1)[:, k]))
    A = D + rng.uniform(-E, E, size=D.shape)                                   # screening values: anywhere within E of the exact ones
    bounds, stride = _plan(n, 128, k, terms, LL)
    got, flagged = _model(D, A, k, LL, E, bounds, stride)
    exact = np.argsort(D, axis=1, kind="stable")[:, :k]
    ok = ~flagged
    assert ok.sum() >= nq // 2, (int(ok.sum()), "the model flags most queries: the case proves nothing")
    assert np.array_equal(got[ok], exact[ok])
    # and the flag is what protects the rest: with a band wider than the lists hold, wrong answers appear ONLY among flagged queries
    wrong = (got != exact).any(1)
    assert not (wrong & ok).any()


def test_threshold_scan_model_flags_what_it_cannot_answer():
    """A band that holds more candidates than the lists do: those queries are flagged (the device recomputes them exactly), the
    others are still exact."""
    rng = np.random.default_rng(3)
    n, d, nq, k, LL = 131_072, 16, 32, 20, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = X[rng.choice(n, nq, replace=False)]
    D = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    E = 0.05 * float(np.median(np.sort(D, axis=1)[:, k]))                      # dozens of candidates inside 2E of the k-th
    A = D + rng.uniform(-E, E, size=D.shape)
    bounds, stride = _plan(n, 128, k, 1, LL)
    got, flagged = _model(D, A, k, LL, E, bounds, stride)
    exact = np.argsort(D, axis=1, kind="stable")[:, :k]
    assert flagged.any()
    assert np.array_equal(got[~flagged], exact[~flagged])
