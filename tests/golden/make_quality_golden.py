"""Quality bar of the C2 test from the REFERENCE itself: TorchDR's UMAP (backend=None, CPU) on a 20k-point
instance of the benchmark generator (the 100k-point config cannot run through the reference's dense CPU path: 3 x 40 GB), scored
with the reference's own `neighborhood_preservation` and a kNN label accuracy.  Run in the build container:

    PYTHONPATH=/root/reference python tests/golden/make_quality_golden.py

Writes tests/golden/quality.json (numbers only; the data is regenerated from the seed by the test)."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.conftest import gmm  # noqa: E402


def label_accuracy(Z, lab, k=10):
    D = torch.cdist(Z, Z)
    D.fill_diagonal_(float("inf"))
    idx = D.topk(k, largest=False).indices
    votes = lab[idx]
    pred = torch.mode(votes, dim=1).values
    return float((pred == lab).float().mean())


def main():
    import torchdr
    from torchdr.eval import neighborhood_preservation

    n, d, k = 20000, 128, 30
    X = gmm(n, d, 2.0, seed=42)
    lab = torch.arange(n) % max(1, min(1000, n // 100))
    out = {"n": n, "d": d, "n_neighbors": k, "generator": "tests.conftest.gmm(n, d, 2.0, seed=42)", "runs": []}
    for seed in (0, 1):
        t0 = time.time()
        m = torchdr.UMAP(n_neighbors=k, backend=None, device="cpu", random_state=seed, max_iter=500)
        Z = m.fit_transform(X)
        Z = torch.as_tensor(Z)
        np_ = float(neighborhood_preservation(X, Z, K=15, backend=None))
        acc = label_accuracy(Z, lab, 10)
        out["runs"].append({"random_state": seed, "max_iter": 500, "neighborhood_preservation_K15": np_, "knn_label_accuracy_k10": acc,
                            "sec": time.time() - t0})
        print(out["runs"][-1], flush=True)
    out["torchdr_version"] = getattr(torchdr, "__version__", "?")
    json.dump(out, open(os.path.join(HERE, "quality.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
