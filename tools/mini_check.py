"""Ten-second GPU sanity check: three estimators, five iterations each (python tools/mini_check.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torchdr_amd, warnings
warnings.filterwarnings("ignore")
from tests.conftest import gmm
X = gmm(3000, 16, 2.0, seed=1).cuda()
for cls, kw in ((torchdr_amd.TSNE, dict(perplexity=10)), (torchdr_amd.LargeVis, dict(perplexity=10)), (torchdr_amd.UMAP, dict(n_neighbors=10))):
    Z = cls(max_iter=5, random_state=0, **kw).fit_transform(X)
    assert Z.shape == (3000, 2) and bool(torch.isfinite(Z).all())
print("mini ok")
