from .base import (  # noqa: F401
    LIST_METRICS,
    PackedPoints,
    knn_packed,
    dense_packed,
    pairwise_distances,
    pairwise_distances_indexed,
)
from .faiss import FaissConfig  # noqa: F401
from .backends import (  # noqa: F401,E402
    LIST_METRICS_FAISS,
    LIST_METRICS_TORCH,
    pairwise_distances_faiss,
    pairwise_distances_faiss_from_dataloader,
    pairwise_distances_torch,
)
