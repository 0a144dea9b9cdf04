"""GPU probe: wall time of the phases of UMAP.fit_transform at the headline size (what is left outside the two
dominant kernels -- the part that does not shrink when rows are sharded over more GPUs)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
import torchdr_amd
from torchdr_amd.utils.validation import validate_tensor  # noqa: F401

X = gmm(1_000_000, 128, 2.0).cuda()
T = {}


def tic():
    torch.cuda.synchronize()
    return time.perf_counter()


class Probe(torchdr_amd.UMAP):
    def on_affinity_computation_start(self):
        T["pre (dedup / validation)"] = tic() - self._t0
        self._t1 = tic()
        super().on_affinity_computation_start()

    def on_affinity_computation_end(self):
        T["affinity (kNN + sigma search + symmetrisation)"] = tic() - self._t1
        t = tic()
        super().on_affinity_computation_end()
        T["epoch counters / exclusion tables"] = tic() - t
        self._t2 = tic()

    def on_training_step_start(self):
        if int(self.n_iter_) == 0:
            T["init embedding (PCA) + optimizer setup"] = tic() - self._t2
            self._t3 = tic()
        super().on_training_step_start()


for rep in range(2):
    m = Probe(n_neighbors=30, random_state=0)
    m._t0 = tic()
    Z = m.fit_transform(X)
    total = tic() - m._t0
    T["loop (1000 iterations)"] = tic() - m._t3
T["total"] = total
print(json.dumps({k: round(v, 4) for k, v in T.items()}, indent=1))
