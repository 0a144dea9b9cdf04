import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as g

    # the oracle is plain C; the HIP library is cross-compiled here and shipped to the GPU box
    g.build_oracle()
    if not os.path.exists(g.LIB):
        g.build_hip()
    torch.manual_seed(42)


def gmm(n, d, scale, seed=42):
    """Gaussian-mixture generator of the reference's benchmark (benchmarks/faiss/run_benchmark.py:127-146
    shape: n_clusters = min(1000, n // 100), centres * scale, sigma 0.5)."""
    g = torch.Generator().manual_seed(seed)
    nc = max(1, min(1000, n // 100))
    centers = torch.randn(nc, d, generator=g) * scale
    labels = torch.arange(n) % nc
    return (centers[labels] + 0.5 * torch.randn(n, d, generator=g)).contiguous()
