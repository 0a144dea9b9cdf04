"""Evaluation metrics that consume the exact kNN kernel -- mirror of ``torchdr/eval`` (reference
``eval/neighborhood_preservation.py:15-200`` and ``eval/knn_labels.py:17-190``)."""

from .neighborhood_preservation import neighborhood_preservation  # noqa: F401
from .knn_labels import knn_label_accuracy  # noqa: F401
