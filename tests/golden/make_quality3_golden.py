"""Embedding-quality bars from the REFERENCE itself for the LargeVis sampler gate of round 6 (the run-permutation sampler of
tdr_ne_grad_runs_f32): TorchDR's LargeVis (backend=None, CPU, perplexity 10, 500 iterations, two seeds) on four data regimes x two
sizes (tests.conftest.regime_data), scored with the
reference's own `neighborhood_preservation` (K = 15), a 10-NN label accuracy and -- the reference's own integration check,
tests/test_neighbor_embedding.py:42-74 -- the silhouette score of the embedding under the labels (sklearn).  Run in the build
container:

    PYTHONPATH=/root/reference python tests/golden/make_quality3_golden.py

Writes tests/golden/quality3.json (numbers only; the data is regenerated from the seed by the tests)."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.conftest import regime_data  # noqa: E402


def label_accuracy(Z, lab, k=10):
    acc = []
    for r0 in range(0, Z.shape[0], 4096):
        D = torch.cdist(Z[r0:r0 + 4096], Z)
        D[torch.arange(D.shape[0]), torch.arange(r0, r0 + D.shape[0])] = float("inf")
        idx = D.topk(k, largest=False).indices
        acc.append(torch.mode(lab[idx], dim=1).values == lab[r0:r0 + 4096])
    return float(torch.cat(acc).float().mean())


def main():
    import torchdr
    from sklearn.metrics import silhouette_score
    from torchdr.eval import neighborhood_preservation

    out = {"perplexity": 10, "max_iter": 500, "cases": []}
    for name in ("gmm2", "overlap", "swiss", "heavytail"):
        for n in (5000, 20000):
            X, lab = regime_data(name, n)
            for seed in (0, 1):
                t0 = time.time()
                Z = torch.as_tensor(torchdr.LargeVis(perplexity=10, backend=None, device="cpu", random_state=seed, max_iter=500).fit_transform(X)).detach()
                rec = {"regime": name, "n": n, "random_state": seed,
                       "neighborhood_preservation_K15": float(neighborhood_preservation(X, Z, K=15, backend=None)),
                       "knn_label_accuracy_k10": label_accuracy(Z, lab, 10),
                       "silhouette": float(silhouette_score(Z.numpy(), lab.numpy(), sample_size=5000, random_state=0)),
                       "sec": time.time() - t0}
                out["cases"].append(rec)
                print(rec, flush=True)
                json.dump(out, open(os.path.join(HERE, "quality3.json"), "w"), indent=1)
    out["torchdr_version"] = getattr(torchdr, "__version__", "?")
    json.dump(out, open(os.path.join(HERE, "quality3.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
