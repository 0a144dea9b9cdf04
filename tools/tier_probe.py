"""GPU probe: which screening tier the pruned kNN build picks at the headline size and how many candidates the one-term / two-term
error bands would hold (share of pilot queries whose band holds >= L candidates): python tools/tier_probe.py"""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import base as B
from torchdr_amd.distance import pairwise_distances
X = gmm(1_000_000, 128, 2.0).cuda()
pairwise_distances(X, metric="sqeuclidean", k=30, exclude_diag=True)
print(json.dumps({k: (v if not torch.is_tensor(v) else None) for k, v in B.LAST_KNN.items()}, default=str))
Y = B.PackedPoints(X)
ops = B._screen_operands(Y, Y)
for tier, L, terms in ((0, 62, 0), (0, 128, 0), (1, 62, 2), (1, 48, 2), (1, 128, 2), (1, 62, 0)):
    try:
        f = B._flat_pilot(Y, Y, ops, 0, 30, "sqeuclidean", True, 0, tier, L, terms)
        print(json.dumps({"tier": tier, "pred_L": L, "pred_terms": terms, "share_of_queries_with_band_ge_L": f}))
    except Exception as e:
        print("err", tier, L, terms, repr(e)[:200])
