from .base import Affinity, LogAffinity, SparseAffinity, SparseLogAffinity  # noqa: F401
from .entropic import EntropicAffinity, SinkhornAffinity, SymmetricEntropicAffinity  # noqa: F401
from .knn_normalized import PACMAPAffinity, UMAPAffinity  # noqa: F401
