"""torchdr_amd -- MI355X-native neighbor-embedding hot path behind TorchDR's plugin API."""

__version__ = "0.1.0"

from torchdr_amd.distance import pairwise_distances, pairwise_distances_indexed  # noqa: F401
from torchdr_amd.distributed import DistributedContext  # noqa: F401
from torchdr_amd.affinity import (  # noqa: F401,E402
    Affinity, LogAffinity, SparseAffinity, SparseLogAffinity, EntropicAffinity, UMAPAffinity,
    SymmetricEntropicAffinity, SinkhornAffinity, PACMAPAffinity,
)
from torchdr_amd.neighbor_embedding import UMAP, LargeVis, TSNE, TSNEkhorn, SNE, InfoTSNE, PACMAP, COSNE  # noqa: F401,E402
from torchdr_amd.base import DRModule  # noqa: F401,E402
from torchdr_amd.affinity_matcher import AffinityMatcher  # noqa: F401,E402
from torchdr_amd.neighbor_embedding.base import NeighborEmbedding, NegativeSamplingNeighborEmbedding  # noqa: F401,E402
from torchdr_amd import eval  # noqa: F401,E402,A004
from torchdr_amd.eval import knn_label_accuracy, neighborhood_preservation  # noqa: F401,E402
from torchdr_amd.utils import binary_search, false_position  # noqa: F401,E402
