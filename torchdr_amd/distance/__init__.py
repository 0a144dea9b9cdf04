from .base import (  # noqa: F401
    LIST_METRICS,
    PackedPoints,
    knn_packed,
    dense_packed,
    pairwise_distances,
    pairwise_distances_indexed,
)
from .faiss import FaissConfig  # noqa: F401
