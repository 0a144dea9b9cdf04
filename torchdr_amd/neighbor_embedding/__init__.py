from .base import NeighborEmbedding, NegativeSamplingNeighborEmbedding  # noqa: F401
from .umap import UMAP, find_ab_params  # noqa: F401
from .largevis import LargeVis  # noqa: F401
from .tsne import TSNE  # noqa: F401
from .tsnekhorn import TSNEkhorn  # noqa: F401
from .sne import SNE  # noqa: F401
from .infotsne import InfoTSNE  # noqa: F401
from .pacmap import PACMAP  # noqa: F401
from .cosne import COSNE  # noqa: F401
