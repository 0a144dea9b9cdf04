"""The reference's own integration tests, re-run against this implementation: ``torchdr/tests/test_neighbor_embedding.py``
:42-74 (every neighbour-embedding method on 100 two-moons points, 100 iterations of Adam(lr=1), silhouette > 0.15) and
:98-129 (numpy vs torch initial embeddings give the same result)."""

import numpy as np
import pytest
import torch
from sklearn.datasets import make_moons
from sklearn.metrics import silhouette_score

pytestmark = pytest.mark.gpu

PARAM_OPTIM = {"lr": 1.0, "optimizer": "Adam", "optimizer_kwargs": None}
SEA = {"lr_affinity_in": 1e-1, "max_iter_affinity_in": 1000}


def toy_dataset(n=100, dtype="float32"):
    X, y = make_moons(n_samples=n, noise=0.05, random_state=0)  # torchdr/tests/utils.py:5-9
    return X.astype(dtype), y


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize(
    "name,kwargs",
    [("SNE", {}), ("TSNE", {}), ("TSNEkhorn", {**SEA, "unrolling": False}), ("LargeVis", {}), ("InfoTSNE", {}),
     ("UMAP", {"optimizer": "SGD"}), ("PACMAP", {})],
)
def test_NE(name, kwargs, dtype):
    import warnings

    import torchdr_amd

    n = 100
    X, y = toy_dataset(n, dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # float64 is processed in float32 (one warning per process)
        model = getattr(torchdr_amd, name)(n_components=2, init="normal", max_iter=100, random_state=0,
                                           min_grad_norm=1e-10, **{**PARAM_OPTIM, **kwargs})
        Z = model.fit_transform(X)
    assert isinstance(Z, np.ndarray) and Z.shape == (n, 2) and Z.dtype == np.dtype(dtype)
    assert silhouette_score(Z, y) > 0.15, "Silhouette score should not be too low."


def test_array_init():
    import torchdr_amd

    n = 100
    X, y = toy_dataset(n)
    Z_init_np = np.random.RandomState(0).randn(n, 2).astype("float32")
    outs = []
    for Z_init in (Z_init_np, torch.from_numpy(Z_init_np)):
        model = torchdr_amd.SNE(n_components=2, init=Z_init, max_iter=100, random_state=0, **PARAM_OPTIM)
        Z = model.fit_transform(X)
        assert Z.shape == (n, 2)
        assert silhouette_score(Z, y) > 0.2
        outs.append(Z)
    assert ((outs[0] - outs[1]) ** 2).mean() < 1e-5, "The two inits should yield similar results."


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_COSNE(dtype):
    """torchdr/tests/test_neighbor_embedding.py:77-94, same hyper-parameters: Iris, 2000 Riemannian-Adam steps."""
    import warnings

    from sklearn.datasets import load_iris

    import torchdr_amd

    iris = load_iris()
    X, y = iris.data.astype(dtype), iris.target
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = torchdr_amd.COSNE(lr=5e-2, n_components=2, max_iter=2000, random_state=0, gamma=1,
                                  learning_rate_for_h_loss=0.01, init_scaling=0.01)
        Z = model.fit_transform(X)
    assert isinstance(Z, np.ndarray) and Z.shape == (X.shape[0], 2)
    assert not np.isnan(Z).any(), "COSNE embedding has NaNs."
    assert silhouette_score(Z, y) > 0.15, "Silhouette score should not be too low."
