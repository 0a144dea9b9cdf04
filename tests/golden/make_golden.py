"""Generate the golden vectors in tests/golden/*.npz by IMPORTING THE REAL REFERENCE (TorchDR at
/root/reference, CPU ``backend=None``) in the build container.  The reference never travels to the
GPU box; these small fixtures (inputs + expected outputs) do.

    python tests/golden/make_golden.py

Environment used: python 3.10, torch 2.10.0+rocm7.0 (CPU, MKL 2024.2), 8 threads.
"""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import torchdr  # noqa: E402
from torchdr.affinity import EntropicAffinity, UMAPAffinity  # noqa: E402
from torchdr.distance import pairwise_distances, pairwise_distances_indexed  # noqa: E402
from torchdr.distributed import DistributedContext  # noqa: E402
from torchdr.utils.sparse import symmetrize_sparse  # noqa: E402

from tests.conftest import gmm  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def knn_fixtures():
    cases = []
    for (n, d, s, k, metric, excl) in [
        (1024, 128, 2.0, 30, "sqeuclidean", True),
        (1024, 128, 10.0, 30, "sqeuclidean", True),
        (1024, 32, 0.0, 15, "sqeuclidean", True),
        (1000, 50, 2.0, 90, "euclidean", True),
        (1024, 32, 2.0, 15, "angular", True),
        (777, 128, 2.0, 30, "sqeuclidean", False),
    ]:
        X = gmm(n, d, s, seed=11)
        C, I = pairwise_distances(X, metric=metric, backend=None, exclude_diag=excl, k=k, return_indices=True)
        # the reference's top-(k+8) distances: lets the harness classify boundary ties
        Cw, _ = pairwise_distances(X, metric=metric, backend=None, exclude_diag=excl, k=k + 8, return_indices=True)
        cases.append(dict(n=n, d=d, s=s, k=k, metric=metric, excl=excl, C=C, I=I, Cw=Cw))
    flat = {}
    for i, c in enumerate(cases):
        for key, v in c.items():
            flat[f"c{i}_{key}"] = v if not isinstance(v, str) else np.array(v)
    flat["n_cases"] = len(cases)
    # cross (X != Y) and k >= N
    X = gmm(300, 40, 2.0, seed=12)
    Y = gmm(200, 40, 2.0, seed=13)
    Cx, Ix = pairwise_distances(X, Y, metric="sqeuclidean", backend=None, k=10, return_indices=True)
    flat["cross_C"], flat["cross_I"] = Cx, Ix
    Cd, Id = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=300, return_indices=True)
    assert Id is None
    flat["dense_excl"] = Cd
    save("knn", **flat)


def knn_wide_fixture():
    """D = 784 (MNIST-shaped): beyond K ~ 380 MKL's sgemm splits the contraction, so the reference's distances are no longer ONE
    k-ordered fma chain (oracle/knn_oracle.c header; distance/torch.py:82-91) -- the fixture pins what the real reference returns
    there; the points are regenerated from the seed by the tests (gmm(2048, 784, 2.0, seed=14))."""
    X = gmm(2048, 784, 2.0, seed=14)
    k = 15
    C, I = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=k, return_indices=True)
    Cw, _ = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=k + 8, return_indices=True)
    save("knn_wide", n=2048, d=784, s=2.0, seed=14, k=k, C=C, I=I.to(torch.int32), Cw=Cw)


def indexed_fixture():
    g = torch.Generator().manual_seed(5)
    Z = torch.randn(50, 2, generator=g)
    q = torch.arange(10, 30)
    keys = torch.randint(0, 50, (20, 7), generator=g)
    keys[3, 2] = -1
    D = pairwise_distances_indexed(Z, query_indices=q, key_indices=keys, metric="sqeuclidean")
    save("indexed", Z=Z, q=q, keys=keys, D=D)


def affinity_fixtures():
    X = gmm(600, 20, 2.0, seed=21)
    out = {"X": X}
    for nn in (10, 30):
        aff = UMAPAffinity(n_neighbors=nn, symmetrize=False, backend=None, max_iter=100)
        P, I = aff(X)
        C, I2 = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=nn, return_indices=True)
        assert torch.equal(I, I2)
        out[f"umap{nn}_C"], out[f"umap{nn}_I"], out[f"umap{nn}_P"] = C, I, P
        out[f"umap{nn}_rho"], out[f"umap{nn}_eps"] = aff.rho_, aff.eps_
        affs = UMAPAffinity(n_neighbors=nn, symmetrize=True, backend=None, max_iter=100)
        Ps, Is = affs(X)
        out[f"umap{nn}_Psym"], out[f"umap{nn}_Isym"] = Ps, Is
    for perp in (5, 30):
        aff = EntropicAffinity(perplexity=perp, backend=None, max_iter=100)
        logP, I = aff(X, log=True)
        k = I.shape[1]
        C, I2 = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=k, return_indices=True)
        assert torch.equal(I, I2)
        out[f"ent{perp}_C"], out[f"ent{perp}_I"], out[f"ent{perp}_logP"] = C, I, logP
        out[f"ent{perp}_eps"], out[f"ent{perp}_lognorm"] = aff.eps_, aff.log_normalization_
    save("affinity", **out)


def symmetrize_fixture():
    g = torch.Generator().manual_seed(31)
    n, k = 40, 6
    vals = torch.rand(n, k, generator=g)
    idx = torch.randint(0, n, (n, k), generator=g)
    idx[0, 1] = idx[0, 0]          # duplicate column in a row
    idx[0, 2] = idx[0, 0]          # triple
    idx[5, 0] = 5                  # self loop
    for mode in ("sum_minus_prod", "sum"):
        V, J = symmetrize_sparse(vals, idx, mode=mode)
        if mode == "sum_minus_prod":
            out = dict(vals=vals, idx=idx, V=V, J=J)
        else:
            out.update(V_sum=V, J_sum=J)
    save("symmetrize", **out)


def umap_step_fixture():
    """State before / after three optimisation steps of the real UMAP (hooks capture everything)."""
    from torchdr import UMAP

    torch.manual_seed(0)
    X = gmm(500, 16, 2.0, seed=41)
    rec = {}

    class Probe(UMAP):
        def on_affinity_computation_end(self):
            super().on_affinity_computation_end()
            rec["A_padded_eps_per"] = self.epochs_per_sample.clone()
            rec["NN"] = self.NN_indices_.clone()

        def _training_step(self):
            t = int(self.n_iter_)
            if t < 3:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"neg_{t}"] = self.neg_indices_.clone()
                rec[f"next_{t}"] = self.epoch_of_next_sample.clone()
                rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]))
            out = super()._training_step()
            if t < 3:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                rec[f"nextafter_{t}"] = self.epoch_of_next_sample.clone()
            return out

    m = Probe(n_neighbors=10, max_iter=20, backend=None, init="normal", random_state=0)
    m.fit_transform(X)
    # affinity recomputed for the harness (the estimator overwrote its buffer with epochs_per_sample)
    Ps, Is = UMAPAffinity(n_neighbors=10, backend=None, max_iter=100)(X)
    rec.update(X=X, Psym=Ps, Isym=Is, a=torch.tensor(m._a), b=torch.tensor(m._b), max_iter=torch.tensor(20))
    # full LR sequence of the default LinearLR(1 -> 0)
    p = torch.zeros(1, requires_grad=True)
    opt = torch.optim.SGD([p], lr=1.0)
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=torch.tensor(1.0), end_factor=torch.tensor(0), total_iters=20)
    lrs = []
    for _ in range(20):
        lrs.append(float(opt.param_groups[0]["lr"]))
        opt.step()
        sch.step()
    rec["lr_seq"] = torch.tensor(lrs, dtype=torch.float64)
    save("umap_step", **rec)


def ne_step_fixture():
    """LargeVis / TSNE: (Z, NN, P, neg) -> loss gradient via autograd, momentum step."""
    from torchdr import TSNE, LargeVis

    X = gmm(400, 16, 2.0, seed=51)
    out = {"X": X}
    for name, cls, kw in (("largevis", LargeVis, dict(perplexity=5)), ("tsne", TSNE, dict(perplexity=8))):
        rec = {}

        class Probe(cls):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                    if hasattr(self, "neg_indices_"):
                        rec[f"neg_{t}"] = self.neg_indices_.clone()
                    rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]))
                    rec[f"mom_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["momentum"]))
                    rec[f"exag_{t}"] = torch.tensor(float(self.early_exaggeration_coeff_))
                    if t == 0:
                        rec["P"] = self.affinity_in_.clone()
                        rec["NN"] = self.NN_indices_.clone()
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                    rec[f"loss_{t}"] = loss.detach().clone()
                return loss

        torch.manual_seed(1)
        m = Probe(max_iter=4, backend=None, init="normal", random_state=1, **kw)
        m.fit_transform(X)
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("ne_step", **out)


def ne_step64_fixture():
    """float64 inputs: the reference computes everything in its input's dtype (tests/test_neighbor_embedding.py:34,55-74 run
    every method in float32 and float64) -- kNN, affinity, the epoch counters of UMAP, the embedding and the optimizer state
    are float64.  Three UMAP steps and two LargeVis / TSNE steps of the real reference on float64 data."""
    from torchdr import TSNE, UMAP, LargeVis

    out = {}
    X = gmm(500, 16, 2.0, seed=41).double()
    rec = {}

    class ProbeU(UMAP):
        def on_affinity_computation_end(self):
            super().on_affinity_computation_end()
            rec["eps_per"] = self.epochs_per_sample.clone()
            rec["NN"] = self.NN_indices_.clone()

        def _training_step(self):
            t = int(self.n_iter_)
            if t < 3:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"neg_{t}"] = self.neg_indices_.clone()
                rec[f"next_{t}"] = self.epoch_of_next_sample.clone()
                rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]), dtype=torch.float64)
            o = super()._training_step()
            if t < 3:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                rec[f"nextafter_{t}"] = self.epoch_of_next_sample.clone()
            return o

    torch.manual_seed(0)
    m = ProbeU(n_neighbors=10, max_iter=20, backend=None, init="normal", random_state=0)
    m.fit_transform(X)
    assert rec["Z_0"].dtype == torch.float64 and rec["eps_per"].dtype == torch.float64
    Ps, Is = UMAPAffinity(n_neighbors=10, backend=None, max_iter=100)(X)
    out.update({f"umap_{k_}": v for k_, v in rec.items()})
    out.update(umap_X=X, umap_Psym=Ps, umap_Isym=Is, umap_a=torch.tensor(m._a, dtype=torch.float64),
               umap_b=torch.tensor(m._b, dtype=torch.float64))
    X2 = gmm(400, 16, 2.0, seed=51).double()
    out["ne_X"] = X2
    for name, cls, kw in (("largevis", LargeVis, dict(perplexity=5)), ("tsne", TSNE, dict(perplexity=8))):
        rec = {}

        class Probe(cls):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                    if hasattr(self, "neg_indices_"):
                        rec[f"neg_{t}"] = self.neg_indices_.clone()
                    rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]), dtype=torch.float64)
                    rec[f"mom_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["momentum"]), dtype=torch.float64)
                    rec[f"exag_{t}"] = torch.tensor(float(self.early_exaggeration_coeff_), dtype=torch.float64)
                    if t == 0:
                        rec["P"] = self.affinity_in_.clone()
                        rec["NN"] = self.NN_indices_.clone()
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                return loss

        torch.manual_seed(1)
        Probe(max_iter=4, backend=None, init="normal", random_state=1, **kw).fit_transform(X2)
        assert rec["Z_0"].dtype == torch.float64 and rec["P"].dtype == torch.float64
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("ne_step64", **out)


def ne2_step_fixture():
    """SNE / InfoTSNE (the SURVEY section 8f "next" estimators): autograd gradient + momentum step."""
    from torchdr import SNE, InfoTSNE

    X = gmm(400, 16, 2.0, seed=53)
    out = {"X": X}
    for name, cls, kw in (("sne", SNE, dict(perplexity=6)), ("infotsne", InfoTSNE, dict(perplexity=7, n_negatives=40))):
        rec = {}

        class Probe(cls):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                    if hasattr(self, "neg_indices_"):
                        rec[f"neg_{t}"] = self.neg_indices_.clone()
                    rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]))
                    rec[f"mom_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["momentum"]))
                    rec[f"exag_{t}"] = torch.tensor(float(self.early_exaggeration_coeff_))
                    if t == 0:
                        rec["P"] = self.affinity_in_.clone()
                        rec["NN"] = self.NN_indices_.clone()
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                    rec[f"loss_{t}"] = loss.detach().clone()
                return loss

        torch.manual_seed(2)
        # init scaling 1.0 so that the Gaussian / Student kernels are away from their flat d ~ 0 regime
        m = Probe(max_iter=4, backend=None, init="normal", init_scaling=1.0, random_state=2, **kw)
        m.fit_transform(X)
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("ne2_step", **out)


def distributed_fixture():
    out = {}
    for n in (97, 100, 103):
        for w in (3, 4, 7, 8):
            ctx = DistributedContext(force_enable=True)
            ctx.world_size = w
            bounds = []
            for r in range(w):
                ctx.rank = r
                bounds.append(ctx.compute_chunk_bounds(n))
            out[f"bounds_{n}_{w}"] = np.array(bounds)
            out[f"owner_{n}_{w}"] = DistributedContext.get_rank_for_indices(torch.arange(n), n, w)
    save("distributed", **out)


def tsnekhorn_fixture():
    from torchdr import TSNEkhorn
    from torchdr.affinity import SinkhornAffinity, SymmetricEntropicAffinity

    X = gmm(256, 16, 2.0, seed=61)
    out = {"X": X}
    sea = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False, backend=None)
    logP = sea(X, log=True)
    out.update(sea_logP=logP, sea_eps=sea.eps_.detach(), sea_mu=sea.mu_.detach(), sea_n_iter=torch.tensor(int(sea.n_iter_)))
    sea_z = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=8, tol=1e-3, zero_diag=True, backend=None)
    out["sea_zd_logP"] = sea_z(X, log=True)
    g = torch.Generator().manual_seed(62)
    Z = torch.randn(256, 2, generator=g) * 3
    init = torch.randn(256, generator=g) * 0.1
    sk = SinkhornAffinity(base_kernel="student", max_iter=5, backend=None)
    logQ = sk(Z, log=True, init_dual=init.clone())
    out.update(sk_Z=Z, sk_init=init, sk_dual=sk.dual_.detach(), sk_logQ=logQ)
    rec = {}

    class Probe(TSNEkhorn):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 2:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
            loss = super()._training_step()
            if t < 2:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"dual_{t}"] = self.dual_sinkhorn_.detach().clone()
                rec[f"loss_{t}"] = loss.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                if t == 0:
                    rec["logP"] = self.affinity_in_.log().detach().clone() if False else None
            return loss

    torch.manual_seed(3)
    m = Probe(perplexity=10, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0,
              min_grad_norm=1e-12, lr=1.0, optimizer="SGD", optimizer_kwargs=None, backend=None, random_state=3)
    m.fit_transform(X)
    for k_, v in rec.items():
        if v is not None:
            out[f"tk_{k_}"] = v
    save("tsnekhorn", **out)


def dense_affinity_fixture():
    X = gmm(300, 12, 2.0, seed=71)
    out = {"X": X}
    aff = EntropicAffinity(perplexity=10, sparsity=False, backend=None, max_iter=100)
    out["ent_logP"] = aff(X, log=True, return_indices=False)
    out["ent_eps"] = aff.eps_
    au = UMAPAffinity(n_neighbors=10, sparsity=False, backend=None, max_iter=100)
    out["umap_P"] = au(X, return_indices=False)
    out["umap_eps"], out["umap_rho"] = au.eps_, au.rho_
    save("affinity_dense", **out)


def eval_fixture():
    """eval metrics of the reference on a reference UMAP embedding: value parity of the metrics themselves and
    the end-to-end quality bar for our estimators (same data, same hyper-parameters)."""
    from torchdr import UMAP
    from torchdr.eval import knn_label_accuracy, neighborhood_preservation

    n, nc = 3000, 30
    X = gmm(n, 32, 3.0, seed=81)
    labels = torch.arange(n) % nc
    torch.manual_seed(0)
    Z = UMAP(n_neighbors=15, max_iter=300, random_state=0, backend=None, device="cpu").fit_transform(X)
    out = {"X": X, "labels": labels, "Z_ref": Z}
    for K in (10, 30):
        out[f"np_K{K}"] = neighborhood_preservation(X, Z, K=K, backend=None, device="cpu")
        out[f"np_per_sample_K{K}"] = neighborhood_preservation(X, Z, K=K, backend=None, device="cpu", return_per_sample=True)
    out["acc_k10"] = knn_label_accuracy(Z, labels, k=10, backend=None, device="cpu")
    out["acc_X_k10"] = knn_label_accuracy(X, labels, k=10, backend=None, device="cpu")
    out["acc_per_sample_k10"] = knn_label_accuracy(Z, labels, k=10, backend=None, device="cpu", return_per_sample=True)
    save("eval", **out)


def pacmap_fixture():
    """PACMAPAffinity indices / rho and four PaCMAP steps (one per weight phase incl. the one-step lag of the
    schedule): embedding, sampled mid-near / further tables, weights, autograd gradient, Adam-updated embedding."""
    from torchdr import PACMAP
    from torchdr.affinity import PACMAPAffinity

    X = gmm(400, 16, 2.0, seed=91)
    out = {"X": X}
    aff = PACMAPAffinity(n_neighbors=10, backend=None)
    _, idx = aff(X)
    out["aff_idx"], out["aff_rho"] = idx, aff.rho_
    rec = {}

    class Probe(PACMAP):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 4:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"neg_{t}"] = self.neg_indices_.clone()
                rec[f"w_{t}"] = torch.tensor([float(self.w_NB), float(self.w_MN), float(self.w_FP)])
                if t == 0:
                    rec["NN"] = self.NN_indices_.clone()
            loss = super()._training_step()
            if t < 4:
                rec[f"mid_{t}"] = self.mid_near_indices.clone()
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
            return loss

    torch.manual_seed(4)
    m = Probe(n_neighbors=10, max_iter=5, iter_per_phase=1, backend=None, init="normal", init_scaling=1.0,
              random_state=4, device="cpu")
    m.fit_transform(X)
    for k_, v in rec.items():
        out[f"pm_{k_}"] = v
    save("pacmap", **out)


def manhattan_fixture():
    """metric='manhattan' (distance/torch.py:96-98, distance/base.py:368, 388): kNN incl. a cascade-length feature
    dimension and an integer-valued (heavily tied) set, cross / dense forms, the gathered forms, one affinity."""
    out = {}
    cases = [(600, 128, 2.0, 15, True), (500, 37, 2.0, 10, True), (300, 5, 0.0, 7, False), (250, 600, 2.0, 5, True)]
    for i, (n, d, s, k, excl) in enumerate(cases):
        X = gmm(n, d, s, seed=31 + i)
        C, I = pairwise_distances(X, metric="manhattan", backend=None, exclude_diag=excl, k=k, return_indices=True)
        Cw, _ = pairwise_distances(X, metric="manhattan", backend=None, exclude_diag=excl, k=k + 8, return_indices=True)
        out.update({f"c{i}_n": n, f"c{i}_d": d, f"c{i}_s": s, f"c{i}_k": k, f"c{i}_excl": excl, f"c{i}_C": C, f"c{i}_I": I,
                    f"c{i}_Cw": Cw})
    out["n_cases"] = len(cases)
    Xt = torch.randint(0, 3, (400, 16), generator=torch.Generator().manual_seed(35)).float()   # ties everywhere
    Ct, It = pairwise_distances(Xt, metric="manhattan", backend=None, exclude_diag=True, k=6, return_indices=True)
    Ctw, _ = pairwise_distances(Xt, metric="manhattan", backend=None, exclude_diag=True, k=14, return_indices=True)
    out.update(ties_X=Xt, ties_C=Ct, ties_I=It, ties_Cw=Ctw)
    X = gmm(300, 40, 2.0, seed=12)
    Y = gmm(200, 40, 2.0, seed=13)
    out["cross_C"], out["cross_I"] = pairwise_distances(X, Y, metric="manhattan", backend=None, k=10, return_indices=True)
    out["cross_dense"] = pairwise_distances(X, Y, metric="manhattan", backend=None)
    out["dense_excl"] = pairwise_distances(X, metric="manhattan", backend=None, exclude_diag=True)
    g = torch.Generator().manual_seed(5)
    keys = torch.randint(0, 300, (40, 7), generator=g)
    keys[3, 2] = -1
    q = torch.arange(100, 140)
    out["keys"], out["q"] = keys, q
    out["indexed_l1"] = pairwise_distances_indexed(X, query_indices=q, key_indices=keys, metric="manhattan")
    out["indexed_ang"] = pairwise_distances_indexed(X, query_indices=q, key_indices=keys, metric="angular")
    out["block_l1"] = pairwise_distances_indexed(X, query_indices=q, key_indices=torch.arange(5, 90), metric="manhattan")
    Xa = gmm(500, 37, 2.0, seed=32)
    aff = UMAPAffinity(n_neighbors=10, metric="manhattan", symmetrize=False, backend=None, max_iter=100)
    P, I = aff(Xa)
    out["umap_P"], out["umap_I"], out["umap_rho"], out["umap_eps"] = P, I, aff.rho_, aff.eps_
    save("manhattan", **out)


def ball_points(n, d, seed, radius=0.9):
    """Points of the open unit ball (inputs of the sqhyperbolic metric)."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(n, d, generator=g)
    r = radius * torch.rand(n, 1, generator=g) ** (1.0 / d)
    return (v / v.norm(dim=1, keepdim=True) * r).contiguous()


def hyperbolic_fixture():
    """metric='sqhyperbolic' (distance/torch.py:101-107, distance/base.py:372-377, 392-398) on points of the unit ball."""
    out = {}
    X, Y = ball_points(400, 5, 51), ball_points(150, 5, 52)
    out["knn_C"], out["knn_I"] = pairwise_distances(X, metric="sqhyperbolic", backend=None, exclude_diag=True, k=9,
                                                   return_indices=True)
    out["knn_Cw"], _ = pairwise_distances(X, metric="sqhyperbolic", backend=None, exclude_diag=True, k=17, return_indices=True)
    out["cross_C"], out["cross_I"] = pairwise_distances(X, Y, metric="sqhyperbolic", backend=None, k=6, return_indices=True)
    out["cross_dense"] = pairwise_distances(X, Y, metric="sqhyperbolic", backend=None)
    out["dense_excl"] = pairwise_distances(X[:120], metric="sqhyperbolic", backend=None, exclude_diag=True)
    g = torch.Generator().manual_seed(5)
    keys = torch.randint(0, 400, (40, 7), generator=g)
    keys[3, 2] = -1
    q = torch.arange(100, 140)
    out["keys"], out["q"] = keys, q
    out["indexed"] = pairwise_distances_indexed(X, query_indices=q, key_indices=keys, metric="sqhyperbolic")
    out["block"] = pairwise_distances_indexed(X, query_indices=q, key_indices=torch.arange(5, 90), metric="sqhyperbolic")
    save("hyperbolic", X=X, Y=Y, **out)


def cosne_fixture():
    """COSNE (neighbor_embedding/cosne.py:162-193 + utils/radam.py + utils/manifold.py): float64 Poincare-ball
    trajectories of the reference.  Per recorded iteration: the state the step starts from (embedding, Adam moments,
    step counter), the Riemannian gradient the optimizer saw (egrad2rgrad rescales ``.grad`` in place) and the state
    after the step.  Two runs: the default ``lr='auto'`` (= N/4: every point is thrown onto the boundary of the ball in
    one step, where the ball arithmetic is ill-conditioned) and ``lr=0.05`` (points stay interior: tight parity)."""
    from torchdr import COSNE

    X = gmm(300, 10, 2.0, seed=41)
    keep = (0, 1, 2, 5, 19)
    out = {"X": X}

    def run(prefix, **kw):
        rec = {}

        class Rec(COSNE):
            def _opt_state(self, tag, it):
                st = self.optimizer_.state.get(self.embedding_, {})
                zero = torch.zeros_like(self.embedding_.detach())
                rec[f"EA{tag}{it}"] = st["exp_avg"].clone() if st else zero
                rec[f"ES{tag}{it}"] = st["exp_avg_sq"].clone() if st else zero
                rec[f"step{tag}{it}"] = int(self.optimizer_.param_groups[0].get("step", 0))

            def on_training_step_start(self):
                super().on_training_step_start()
                it = int(self.n_iter_)
                if it == 0:
                    rec["P"], rec["NN"] = self.affinity_in_.clone(), self.NN_indices_.clone()
                    rec["lr"] = float(self.lr_)
                if it in keep:   # state the step starts from
                    rec[f"Zb{it}"] = self.embedding_.detach().clone()
                    self._opt_state("b", it)

            def on_training_step_end(self):
                it = int(self.n_iter_)
                if it in keep:   # state the step produced; .grad holds the Riemannian gradient (rescaled in place)
                    rec[f"Za{it}"] = self.embedding_.detach().clone()
                    rec[f"R{it}"] = self.embedding_.grad.detach().clone()
                    self._opt_state("a", it)
                return super().on_training_step_end()

        m = Rec(perplexity=10, max_iter=20, random_state=0, learning_rate_for_h_loss=0.1, gamma=2, **kw)
        Z = m.fit_transform(X)
        assert Z.dtype == torch.float64 and torch.equal(Z, rec["Za19"])
        out.update({prefix + k: v for k, v in rec.items()})

    run("auto_")
    run("small_", lr=0.05)
    save("cosne", **out)


def affinity64_fixture():
    """float64 inputs (the reference computes in the input's dtype, tests/test_affinity.py:54-60): kNN, both root-search
    affinities, the symmetrised UMAP graph and gathered distances, all in float64."""
    X = gmm(600, 16, 2.0, seed=64).double()
    out = {"X": X}
    C, I = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=True, k=15, return_indices=True)
    out["knn_C"], out["knn_I"] = C, I
    Ce = pairwise_distances(X[:50], X[100:300], metric="euclidean", backend=None)
    out["cross_euclid"] = Ce
    ea = EntropicAffinity(perplexity=5, backend=None)
    lp, idx = ea(X, log=True, return_indices=True)
    out["ent_logP"], out["ent_idx"], out["ent_eps"] = lp, idx, ea.eps_
    ua = UMAPAffinity(n_neighbors=10, backend=None)
    P, J = ua(X, return_indices=True)
    out["umap_P"], out["umap_J"], out["umap_rho"], out["umap_eps"] = P, J, ua.rho_, ua.eps_
    q = torch.arange(0, 600, 7)
    keys = torch.randint(0, 600, (q.numel(), 5), generator=torch.Generator().manual_seed(1))
    keys[3, 2] = -1
    out["idx_q"], out["idx_keys"] = q, keys
    out["idx_D"] = pairwise_distances_indexed(X, query_indices=q, key_indices=keys)
    for v in (C, lp, P, out["idx_D"], Ce):
        assert v.dtype == torch.float64
    save("affinity64", **out)


def sinkhorn_fixture():
    """SinkhornAffinity on INPUT points (entropic.py:693-755): Gaussian base kernel (the class default) and the student
    base kernel on an 8-d input, cold and warm started: duals, iteration counts, a few rows of log P."""
    from torchdr.affinity import SinkhornAffinity

    X = gmm(300, 8, 1.5, seed=77)
    out = {"X": X}
    for name, kw in (("gauss", dict(eps=5.0, base_kernel="gaussian")), ("gauss_nozd", dict(eps=20.0, base_kernel="gaussian", zero_diag=False)),
                     ("student", dict(eps=2.0, base_kernel="student"))):
        aff = SinkhornAffinity(tol=1e-5, max_iter=300, backend=None, **kw)
        logP = aff(X, log=True)
        out[f"{name}_dual"], out[f"{name}_n_iter"], out[f"{name}_logP_rows"] = aff.dual_.clone(), torch.tensor(aff.n_iter_), logP[:6].clone()
    aff = SinkhornAffinity(eps=5.0, tol=1e-5, max_iter=3, backend=None)
    init = torch.linspace(-1, 1, 300)
    aff._compute_log_affinity(X, init_dual=init.clone())
    out["warm_init"], out["warm_dual"] = init, aff.dual_.clone()
    save("sinkhorn", **out)


def sea_lbfgs_fixture():
    """The objective SymmetricEntropicAffinity(optimizer="LBFGS") hands to torch.optim.LBFGS (entropic.py:483-491): loss
    and autograd gradients of the reference's own closure expression at given duals, with and without the squared
    parameterisation of eps, with and without the diagonal.  (The reference's LBFGS RUN is not recorded: with its
    defaults it ends in NaN duals or overflows inside the line search on this kind of data; the objective is what can be
    pinned.)"""
    from torchdr.affinity.entropic import _log_Pse
    from torchdr.utils import entropy

    X = gmm(256, 16, 2.0, seed=61)
    out = {"X": X}
    gen = torch.Generator().manual_seed(7)
    target = torch.log(torch.tensor(10.0)) + 1
    for name, eps_square, zero_diag in (("sq", True, True), ("lin", False, False)):
        C = pairwise_distances(X, metric="sqeuclidean", backend=None, exclude_diag=zero_diag)
        if isinstance(C, tuple):
            C = C[0]
        eps = (0.8 + 0.4 * torch.rand(256, generator=gen)).requires_grad_(True)
        mu = (0.5 * torch.randn(256, generator=gen)).requires_grad_(True)
        _eps = eps**2 if eps_square else eps
        log_P = _log_Pse(C, _eps, mu, eps_square=False)
        H = entropy(log_P, log=True, dim=1)
        loss = -(log_P.exp() * C).sum(0).sum() - torch.inner(_eps, target - H) + torch.inner(mu, log_P.logsumexp(1).squeeze().expm1())
        loss.backward()
        out.update({f"{name}_eps": eps.detach(), f"{name}_mu": mu.detach(), f"{name}_loss": loss.detach(), f"{name}_grad_eps": eps.grad,
                    f"{name}_grad_mu": mu.grad, f"{name}_H": H.detach(), f"{name}_rowsum": log_P.logsumexp(1).exp().detach()})
    save("sea_lbfgs", **out)


def dense_ne_fixture():
    """sparsity=False: TSNE / SNE / LargeVis / InfoTSNE on the DENSE (N, N) entropic affinity (NN_indices_ is None, the
    attraction runs over all pairs: tsne.py:162-170, sne.py:163-172, largevis.py:192-201, infotsne.py:179-188); two
    optimisation steps each, negatives recorded for the two samplers."""
    from torchdr import SNE, TSNE, InfoTSNE, LargeVis

    X = gmm(300, 16, 2.0, seed=57)
    out = {"X": X}
    for name, cls, kw in (("tsne", TSNE, dict(perplexity=8)), ("sne", SNE, dict(perplexity=6)),
                          ("largevis", LargeVis, dict(perplexity=5)),
                          ("infotsne", InfoTSNE, dict(perplexity=7, n_negatives=40))):
        rec = {}

        class Probe(cls):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                    if hasattr(self, "neg_indices_"):
                        rec[f"neg_{t}"] = self.neg_indices_.clone()
                    if t == 0:
                        assert self.NN_indices_ is None and self.affinity_in_.shape == (300, 300)
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                return loss

        torch.manual_seed(4)
        Probe(max_iter=4, backend=None, init="normal", random_state=4, sparsity=False, **kw).fit_transform(X)
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("dense_ne", **out)


def tsnekhorn3_fixture():
    """TSNEkhorn with n_components = 3 (tsnekhorn.py:210-230): two optimisation steps of the reference on the data of
    the `tsnekhorn` fixture -- embedding before, Sinkhorn dual, autograd gradient, embedding after."""
    from torchdr import TSNEkhorn

    X = gmm(256, 16, 2.0, seed=61)
    rec = {}

    class Probe(TSNEkhorn):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 2:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
            loss = super()._training_step()
            if t < 2:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"dual_{t}"] = self.dual_sinkhorn_.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
            return loss

    torch.manual_seed(3)
    Probe(perplexity=10, n_components=3, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0,
          min_grad_norm=1e-12, lr=1.0, optimizer="SGD", optimizer_kwargs=None, backend=None, random_state=3).fit_transform(X)
    save("tsnekhorn3", **rec)


def tsnekhorn_unrolled_fixture():
    """TSNEkhorn(unrolling=True) (tsnekhorn.py:134, 224-227: autograd through the 5 Sinkhorn updates) and TSNEkhorn with
    n_components = 4: two optimisation steps of the reference each, on the data of the `tsnekhorn` fixture -- embedding
    before, Sinkhorn dual, autograd gradient, embedding after; plus the reference's log P for the oracle check."""
    from torchdr import TSNEkhorn

    X = gmm(256, 16, 2.0, seed=61)
    out = {"X": X}
    for name, kw in (("u2", dict(unrolling=True)), ("u3", dict(unrolling=True, n_components=3)), ("n4", dict(n_components=4)),
                     ("u5", dict(unrolling=True, n_components=5))):
        rec = {}

        class Probe(TSNEkhorn):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"dual_{t}"] = self.dual_sinkhorn_.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                if t == 0 and name == "u2":
                    out["log_P"] = self.affinity_in_.detach().log().clone()
                return loss

        torch.manual_seed(3)
        Probe(perplexity=10, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0, min_grad_norm=1e-12, lr=1.0,
              optimizer="SGD", optimizer_kwargs=None, backend=None, random_state=3, **kw).fit_transform(X)
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("tsnekhorn_unrolled", **out)


def signatures_fixture():
    """Constructor / function signatures of the in-scope public surface of the reference: parameter names in order and the
    repr of every default (tests/golden/signatures.json) -- what `from torchdr import X; X(**kwargs)` code relies on."""
    import inspect
    import json

    import torchdr.affinity as RA
    import torchdr.distance as RD
    import torchdr.eval as RE

    def sig(obj):
        ps = inspect.signature(obj).parameters
        return [[k, None if v.default is inspect._empty else repr(v.default)] for k, v in ps.items()
                if k not in ("self", "kwargs")]

    out = {}
    for n in ("UMAP", "TSNE", "LargeVis", "SNE", "InfoTSNE", "PACMAP", "COSNE", "TSNEkhorn", "AffinityMatcher", "NeighborEmbedding",
              "NegativeSamplingNeighborEmbedding"):
        out[n] = sig(getattr(torchdr, n).__init__)
    for n in ("EntropicAffinity", "UMAPAffinity", "SymmetricEntropicAffinity", "SinkhornAffinity", "PACMAPAffinity"):
        out["affinity." + n] = sig(getattr(RA, n).__init__)
    for n in ("pairwise_distances", "pairwise_distances_indexed", "pairwise_distances_torch", "pairwise_distances_faiss",
              "pairwise_distances_faiss_from_dataloader"):
        out["distance." + n] = sig(getattr(RD, n))
    out["distance.FaissConfig"] = sig(RD.FaissConfig.__init__)
    for n in ("neighborhood_preservation", "knn_label_accuracy"):
        out["eval." + n] = sig(getattr(RE, n))
    path = os.path.join(HERE, "signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("signatures:", len(out), "entries")


def manifold_fixture():
    """utils/manifold.py: every operation of PoincareBallManifold (and the non-trivial ones of EuclideanManifold) on random
    float64 inputs at curvatures 1 and 0.7 -- points inside the ball, a few near its boundary, tangent vectors, a matrix."""
    from torchdr.utils.manifold import EuclideanManifold, PoincareBallManifold

    gen = torch.Generator().manual_seed(11)
    out = {}
    for tag, c in (("c1", 1.0), ("c07", 0.7)):
        B = PoincareBallManifold()
        x = torch.randn(40, 3, generator=gen, dtype=torch.float64)
        x = x / x.norm(dim=1, keepdim=True) * torch.rand(40, 1, generator=gen, dtype=torch.float64) * 0.95 / c ** 0.5
        x[:4] = x[:4] / x[:4].norm(dim=1, keepdim=True) * (1 - 1e-7) / c ** 0.5       # near the boundary
        y = torch.randn(40, 3, generator=gen, dtype=torch.float64)
        y = y / y.norm(dim=1, keepdim=True) * torch.rand(40, 1, generator=gen, dtype=torch.float64) * 0.9 / c ** 0.5
        u = torch.randn(40, 3, generator=gen, dtype=torch.float64) * 0.3
        v = torch.randn(40, 3, generator=gen, dtype=torch.float64) * 0.3
        m = torch.randn(3, 3, generator=gen, dtype=torch.float64)
        far = x * 1.7
        out.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_u": u, f"{tag}_v": v, f"{tag}_m": m, f"{tag}_far": far})
        out.update({
            f"{tag}_sqdist": B.sqdist(x, y, c), f"{tag}_egrad2rgrad": B.egrad2rgrad(x, u.clone(), c), f"{tag}_proj": B.proj(far, c),
            f"{tag}_proj32": B.proj(far.float(), c), f"{tag}_expmap": B.expmap(u, y, c), f"{tag}_logmap": B.logmap(y, x, c),
            f"{tag}_expmap0": B.expmap0(u * 20, c), f"{tag}_logmap0": B.logmap0(x, c), f"{tag}_mobius_add": B.mobius_add(x, y, c),
            f"{tag}_mobius_matvec": B.mobius_matvec(m, y, c), f"{tag}_inner": B.inner(y, c, u, v), f"{tag}_inner_self": B.inner(y, c, u, keepdim=True),
            f"{tag}_ptransp": B.ptransp(y, x, u, c), f"{tag}_ptransp0": B.ptransp0(y, u, c), f"{tag}_lambda": B._lambda_x(x, c),
            f"{tag}_hyperboloid": B.to_hyperboloid(y, c),
        })
        z = y.clone().requires_grad_(True)
        B.sqdist(x, z, c).sum().backward()
        out[f"{tag}_sqdist_grad"] = z.grad
    E = EuclideanManifold()
    out["euc_sqdist"] = E.sqdist(x, y, 1.0)
    out["euc_ptransp0"] = E.ptransp0(x, u, 1.0)
    out["euc_normalize"] = E.normalize((x * 3).clone())
    save("manifold", **out)


def radam_fixture():
    """utils/radam.py: trajectories of the reference's RiemannianAdam (a torch.optim.Adam subclass) -- a ManifoldParameter on
    the Poincare ball (c = 0.8) and a plain tensor with amsgrad + weight decay -- five steps each on fixed quadratic losses:
    points, both moments and the group's step counter after every step."""
    from torchdr.utils import ManifoldParameter, PoincareBallManifold, RiemannianAdam

    gen = torch.Generator().manual_seed(21)
    out = {}
    x0 = torch.randn(12, 3, generator=gen, dtype=torch.float64) * 0.2
    tgt = torch.randn(12, 3, generator=gen, dtype=torch.float64) * 0.3
    out.update(ball_init=x0, ball_target=tgt)
    p = ManifoldParameter(x0.clone(), True, PoincareBallManifold(), 0.8)
    opt = RiemannianAdam([p], lr=0.05, stabilize=4)
    for t in range(5):
        opt.zero_grad()
        ((p - tgt) ** 2).sum().backward()
        opt.step()
        st = opt.state[p]
        out.update({f"ball_x{t}": p.detach().clone(), f"ball_m1_{t}": st["exp_avg"].clone(), f"ball_m2_{t}": st["exp_avg_sq"].clone(),
                    f"ball_step{t}": torch.tensor(opt.param_groups[0]["step"])})
    y0 = torch.randn(6, 4, generator=gen, dtype=torch.float64)
    out["flat_init"] = y0
    q = y0.clone().requires_grad_(True)
    opt = RiemannianAdam([q], lr=0.01, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2, amsgrad=True)
    for t in range(5):
        opt.zero_grad()
        (q ** 4).sum().backward()
        opt.step()
        st = opt.state[q]
        out.update({f"flat_x{t}": q.detach().clone(), f"flat_m1_{t}": st["exp_avg"].clone(), f"flat_m2_{t}": st["exp_avg_sq"].clone(),
                    f"flat_max{t}": st["max_exp_avg_sq"].clone()})
    save("radam", **out)


def c1_tsne_fixture():
    """BASELINE config C1 at full size: TSNE on the 5000 x 50 Gaussian mixture, perplexity 30, backend=None (CPU):
    the reference's first two optimisation steps (embedding before / gradient / after, lr, momentum, exaggeration) and
    the entropic bandwidths.  X itself is regenerated by the test (tests.conftest.gmm, same seed)."""
    from torchdr import TSNE

    X = gmm(5000, 50, 2.0, seed=42)
    rec = {}

    class Probe(TSNE):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 2:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]))
                rec[f"mom_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["momentum"]))
                rec[f"exag_{t}"] = torch.tensor(float(self.early_exaggeration_coeff_))
                if t == 0:
                    rec["eps"] = self.affinity_in.eps_.detach().clone()
                    rec["NN_head"] = self.NN_indices_[:, :8].to(torch.int32).clone()
            loss = super()._training_step()
            if t < 2:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                rec[f"loss_{t}"] = loss.detach().clone()
            return loss

    torch.manual_seed(3)
    Probe(perplexity=30, max_iter=3, backend=None, random_state=3).fit_transform(X)
    save("c1_tsne", **rec)


def grad_in_float64(est):
    """The reference's OWN loss at the estimator's current state, evaluated in float64: a deep copy of the estimator with
    every floating tensor (embedding, affinities, Sinkhorn duals ...) cast up, `_compute_loss()` and autograd.  Same
    inputs as the float32 step the fixture records, none of its rounding: what separates the float32 autograd noise of
    the reference from the error of a float32 kernel.  The random generator is restored afterwards (PaCMAP samples inside
    its loss), the estimator itself is left untouched."""
    import copy
    from collections import OrderedDict

    def up(v):
        if torch.is_tensor(v) and v.is_floating_point():
            return v.detach().double()
        return v

    def shallow_double(obj):
        c = copy.copy(obj)
        c.__dict__ = {k: up(v) for k, v in obj.__dict__.items()}
        if isinstance(obj, torch.nn.Module):
            c._parameters = OrderedDict((k, None if v is None else torch.nn.Parameter(up(v))) for k, v in obj._parameters.items())
            c._buffers = OrderedDict((k, up(v)) for k, v in obj._buffers.items())
            c._modules = OrderedDict(obj._modules)
        return c

    with torch.random.fork_rng():
        m = shallow_double(est)
        for name in ("affinity_out", "affinity_in"):
            obj = getattr(est, name, None)
            if obj is not None:
                if name in m.__dict__:
                    m.__dict__[name] = shallow_double(obj)
                else:
                    m._modules[name] = shallow_double(obj)
        emb = m.embedding_.detach().double().requires_grad_(True)
        if "embedding_" in m._parameters:
            m._parameters["embedding_"] = torch.nn.Parameter(emb)
        else:
            m.__dict__["embedding_"] = emb
        loss = m._compute_loss()
        loss.backward()
        assert m.embedding_.grad.dtype == torch.float64
        return m.embedding_.grad.detach().clone()


def grad64_fixture():
    """float64 twins of the gradients held by ne_step / ne2_step / dense_ne / tsnekhorn / tsnekhorn3 / tsnekhorn_unrolled /
    c1_tsne / pacmap: the same estimators run again (same data, seeds and arguments as the functions above), and before
    every recorded step `grad_in_float64` evaluates the reference's loss in float64 AT THE FLOAT32 STATE of that step.
    Every run checks that its float32 gradient is the one the original fixture holds (same state, bit for bit)."""
    from torchdr import PACMAP, SNE, TSNE, InfoTSNE, LargeVis, TSNEkhorn

    out = {}

    def run(fixture, prefix, cls, n_steps, seed, X, kw, key="grad", exact=True):
        have = np.load(os.path.join(HERE, fixture + ".npz"))
        rec = {}

        class Probe(cls):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < n_steps:
                    rec[t] = grad_in_float64(self)
                loss = super()._training_step()
                if t < n_steps:
                    g32 = self.embedding_.grad.detach().numpy()
                    want = have[f"{prefix}{key}_{t}"]
                    if exact:
                        assert np.array_equal(g32, want), (fixture, prefix, t)
                    else:   # 5000 points: the threaded reductions of the initialisation do not repeat bit for bit
                        dev = float(np.abs(g32 - want).max() / np.abs(want).max())
                        print(f"  {fixture}/{prefix}{t}: this run's float32 gradient vs the stored one: {dev:.2e} of max |g|")
                        assert dev < 1e-6, (fixture, prefix, t, dev)
                return loss

        torch.manual_seed(seed)
        Probe(**kw).fit_transform(X)
        for t, g in rec.items():
            out[f"{fixture}/{prefix}grad64_{t}"] = g
            g32 = torch.from_numpy(have[f"{prefix}{key}_{t}"]).double()
            print(f"  {fixture}/{prefix}{t}: reference float32 autograd vs float64: {float((g32 - g).abs().max() / g.abs().max()):.2e} of max |g|")

    X = gmm(400, 16, 2.0, seed=51)
    for name, cls, kw in (("largevis", LargeVis, dict(perplexity=5)), ("tsne", TSNE, dict(perplexity=8))):
        run("ne_step", name + "_", cls, 2, 1, X, dict(max_iter=4, backend=None, init="normal", random_state=1, **kw))
    X = gmm(400, 16, 2.0, seed=53)
    for name, cls, kw in (("sne", SNE, dict(perplexity=6)), ("infotsne", InfoTSNE, dict(perplexity=7, n_negatives=40))):
        run("ne2_step", name + "_", cls, 2, 2, X, dict(max_iter=4, backend=None, init="normal", init_scaling=1.0, random_state=2, **kw))
    X = gmm(300, 16, 2.0, seed=57)
    for name, cls, kw in (("tsne", TSNE, dict(perplexity=8)), ("sne", SNE, dict(perplexity=6)), ("largevis", LargeVis, dict(perplexity=5)),
                          ("infotsne", InfoTSNE, dict(perplexity=7, n_negatives=40))):
        run("dense_ne", name + "_", cls, 2, 4, X, dict(max_iter=4, backend=None, init="normal", random_state=4, sparsity=False, **kw))
    X = gmm(256, 16, 2.0, seed=61)
    tk = dict(perplexity=10, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0, min_grad_norm=1e-12, lr=1.0,
              optimizer="SGD", optimizer_kwargs=None, backend=None, random_state=3)
    run("tsnekhorn", "tk_", TSNEkhorn, 2, 3, X, tk)
    run("tsnekhorn3", "", TSNEkhorn, 2, 3, X, dict(tk, n_components=3))
    for name, kw in (("u2", dict(unrolling=True)), ("u3", dict(unrolling=True, n_components=3)), ("n4", dict(n_components=4)),
                     ("u5", dict(unrolling=True, n_components=5))):
        run("tsnekhorn_unrolled", name + "_", TSNEkhorn, 2, 3, X, dict(tk, **kw))
    X = gmm(400, 16, 2.0, seed=91)
    run("pacmap", "pm_", PACMAP, 4, 4, X, dict(n_neighbors=10, max_iter=5, iter_per_phase=1, backend=None, init="normal", init_scaling=1.0,
                                               random_state=4, device="cpu"))
    X = gmm(5000, 50, 2.0, seed=42)
    run("c1_tsne", "", TSNE, 2, 3, X, dict(perplexity=30, max_iter=3, backend=None, random_state=3), exact=False)
    # C1 end to end in float64: the affinity (kNN, entropic bisection) computed in float64 as well, the embedding of each
    # recorded step set to the float32 fixture's -- how far the reference's OWN float32 pipeline (bisection stopped at its
    # tolerance, float32 P) is from the float64 one at the gradient
    have = np.load(os.path.join(HERE, "c1_tsne.npz"))
    rec = {}

    class Probe64(TSNE):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 2:
                with torch.no_grad():
                    self.embedding_.copy_(torch.from_numpy(have[f"Z_{t}"]).double())
            loss = super()._training_step()
            if t < 2:
                rec[t] = self.embedding_.grad.detach().clone()
            return loss

    torch.manual_seed(3)
    Probe64(perplexity=30, max_iter=3, backend=None, random_state=3).fit_transform(X.double())
    for t, g in rec.items():
        assert g.dtype == torch.float64
        out[f"c1_tsne/grad64_pipeline_{t}"] = g
        g32 = torch.from_numpy(have[f"grad_{t}"]).double()
        print(f"  c1_tsne/{t}: reference float32 PIPELINE (affinity + gradient) vs the float64 pipeline: "
              f"{float((g32 - g).abs().max() / g.abs().max()):.2e} of max |g|")
    # the symmetric entropic affinity of the `tsnekhorn` fixture on the same points in float64 (30 Adam steps on the duals)
    from torchdr.affinity import SymmetricEntropicAffinity

    have = np.load(os.path.join(HERE, "tsnekhorn.npz"))
    X = gmm(256, 16, 2.0, seed=61)
    sea = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False, backend=None)
    logP = sea(X.double(), log=True)
    assert logP.dtype == torch.float64
    out.update({"tsnekhorn/sea64_logP": logP, "tsnekhorn/sea64_eps": sea.eps_.detach(), "tsnekhorn/sea64_mu": sea.mu_.detach(),
                "tsnekhorn/sea64_n_iter": torch.tensor(int(sea.n_iter_))})
    d32 = torch.from_numpy(have["sea_logP"]).double()
    print(f"  tsnekhorn/sea: reference float32 log P vs float64: max abs {float((d32 - logP).abs().max()):.2e}, "
          f"P relative to max P {float((d32.exp() - logP.exp()).abs().max() / logP.exp().max()):.2e}; "
          f"iterations {int(have['sea_n_iter'])} / {int(sea.n_iter_)}; eps rel {float((torch.from_numpy(have['sea_eps']).double() - sea.eps_).abs().max() / sea.eps_.abs().max()):.2e}")
    save("grad64", **out)



def sampler_quality_fixture():
    """End-to-end quality reference for the negative samplers: the REFERENCE's LargeVis and InfoTSNE (independent
    uniform negatives, base.py:617-649) on a 20 000-point mixture, three seeds each -- embeddings are not kept, only what the
    test compares: neighbourhood preservation (K = 15) and kNN label accuracy (k = 10) per seed.  X is regenerated by the
    test (tests.conftest.gmm(20000, 32, 2.0, seed=3), labels = index mod 200)."""
    from torchdr import InfoTSNE, LargeVis
    from torchdr.eval import knn_label_accuracy, neighborhood_preservation

    n = 20000
    X = gmm(n, 32, 2.0, seed=3)
    labels = torch.arange(n) % (n // 100)
    out = {}
    for name, cls, kw in (("largevis", LargeVis, dict(perplexity=10, max_iter=300)), ("infotsne", InfoTSNE, dict(perplexity=10, max_iter=300))):
        nps, accs = [], []
        for seed in (0, 1, 2):
            torch.manual_seed(seed)
            Z = cls(random_state=seed, backend=None, device="cpu", **kw).fit_transform(X)
            nps.append(float(neighborhood_preservation(X, Z, K=15, backend=None, device="cpu")))
            accs.append(float(knn_label_accuracy(Z, labels, k=10, backend=None, device="cpu")))
            print(f"  {name} seed {seed}: neighbourhood preservation {nps[-1]:.4f}, label accuracy {accs[-1]:.4f}", flush=True)
        out[f"{name}_np_K15"] = torch.tensor(nps, dtype=torch.float64)
        out[f"{name}_acc_k10"] = torch.tensor(accs, dtype=torch.float64)
    save("sampler_quality", **out)


def ne2_step64_fixture():
    """float64 inputs for the round-4 float64 twins: SNE (two steps: P, NN, embedding, autograd gradient, embedding after) and
    PaCMAP (four steps, one per weight phase: near / mid-near / further tables, weights, gradient, Adam-updated embedding) of
    the real reference on float64 data -- everything float64, as tests/test_neighbor_embedding.py:34,55-74 run them."""
    from torchdr import PACMAP, SNE

    out = {}
    X = gmm(400, 16, 2.0, seed=53).double()
    out["sne_X"] = X
    rec = {}

    class ProbeS(SNE):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 2:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"lr_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["lr"]), dtype=torch.float64)
                rec[f"mom_{t}"] = torch.tensor(float(self.optimizer_.param_groups[0]["momentum"]), dtype=torch.float64)
                rec[f"exag_{t}"] = torch.tensor(float(self.early_exaggeration_coeff_), dtype=torch.float64)
                if t == 0:
                    rec["P"] = self.affinity_in_.clone()
                    rec["NN"] = self.NN_indices_.clone()
            loss = super()._training_step()
            if t < 2:
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
            return loss

    torch.manual_seed(2)
    ProbeS(perplexity=6, max_iter=4, backend=None, init="normal", init_scaling=1.0, random_state=2).fit_transform(X)
    assert rec["Z_0"].dtype == torch.float64 and rec["P"].dtype == torch.float64 and rec["grad_0"].dtype == torch.float64
    for k_, v in rec.items():
        out[f"sne_{k_}"] = v
    X = gmm(400, 16, 2.0, seed=91).double()
    out["pm_X"] = X
    rec = {}

    class ProbeP(PACMAP):
        def _training_step(self):
            t = int(self.n_iter_)
            if t < 4:
                rec[f"Z_{t}"] = self.embedding_.detach().clone()
                rec[f"neg_{t}"] = self.neg_indices_.clone()
                rec[f"w_{t}"] = torch.tensor([float(self.w_NB), float(self.w_MN), float(self.w_FP)], dtype=torch.float64)
                if t == 0:
                    rec["NN"] = self.NN_indices_.clone()
            loss = super()._training_step()
            if t < 4:
                rec[f"mid_{t}"] = self.mid_near_indices.clone()
                rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
            return loss

    torch.manual_seed(4)
    ProbeP(n_neighbors=10, max_iter=5, iter_per_phase=1, backend=None, init="normal", init_scaling=1.0, random_state=4,
           device="cpu").fit_transform(X)
    assert rec["Z_0"].dtype == torch.float64 and rec["grad_0"].dtype == torch.float64
    for k_, v in rec.items():
        out[f"pm_{k_}"] = v
    save("ne2_step64", **out)


def tsnekhorn64_fixture():
    """float64 twins of the tsnekhorn / tsnekhorn3 / tsnekhorn_unrolled fixtures: the real reference on the float64 copy of the
    same points -- symmetric entropic affinity (30 Adam steps on the duals: log P, eps, mu), and two optimisation steps of
    TSNEkhorn (2 / 3 / 4 components, with and without unrolling): embedding before, Sinkhorn dual, autograd gradient, embedding
    after.  Everything float64, as tests/test_neighbor_embedding.py:34,55-74 run the method."""
    from torchdr import TSNEkhorn
    from torchdr.affinity import SymmetricEntropicAffinity

    X = gmm(256, 16, 2.0, seed=61).double()
    out = {"X": X}
    sea = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=30, tol=1e-3, zero_diag=False, backend=None)
    logP = sea(X, log=True)
    assert logP.dtype == torch.float64
    out.update(sea_logP=logP, sea_eps=sea.eps_.detach(), sea_mu=sea.mu_.detach(), sea_n_iter=torch.tensor(int(sea.n_iter_)))
    sea_z = SymmetricEntropicAffinity(perplexity=10, lr=1e-1, max_iter=8, tol=1e-3, zero_diag=True, backend=None)
    out["sea_zd_logP"] = sea_z(X, log=True)
    for name, kw in (("n2", dict()), ("n3", dict(n_components=3)), ("n4", dict(n_components=4)), ("u2", dict(unrolling=True)),
                     ("u3", dict(unrolling=True, n_components=3))):
        rec = {}

        class Probe(TSNEkhorn):
            def _training_step(self):
                t = int(self.n_iter_)
                if t < 2:
                    rec[f"Z_{t}"] = self.embedding_.detach().clone()
                loss = super()._training_step()
                if t < 2:
                    rec[f"grad_{t}"] = self.embedding_.grad.detach().clone()
                    rec[f"dual_{t}"] = self.dual_sinkhorn_.detach().clone()
                    rec[f"Zafter_{t}"] = self.embedding_.detach().clone()
                return loss

        torch.manual_seed(3)
        Probe(perplexity=10, max_iter=3, max_iter_affinity_in=30, init="normal", init_scaling=1.0, min_grad_norm=1e-12, lr=1.0,
              optimizer="SGD", optimizer_kwargs=None, backend=None, random_state=3, **kw).fit_transform(X)
        assert rec["Z_0"].dtype == torch.float64 and rec["grad_0"].dtype == torch.float64
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = v
    save("tsnekhorn64", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ALL = dict(knn=knn_fixtures, knn_wide=knn_wide_fixture, indexed=indexed_fixture, affinity=affinity_fixtures, symmetrize=symmetrize_fixture,
               umap_step=umap_step_fixture, ne_step=ne_step_fixture, ne_step64=ne_step64_fixture, ne2_step=ne2_step_fixture,
               distributed=distributed_fixture, tsnekhorn=tsnekhorn_fixture, affinity_dense=dense_affinity_fixture,
               eval=eval_fixture, pacmap=pacmap_fixture, manhattan=manhattan_fixture,
               cosne=cosne_fixture, hyperbolic=hyperbolic_fixture, c1_tsne=c1_tsne_fixture, sinkhorn=sinkhorn_fixture, affinity64=affinity64_fixture,
               sea_lbfgs=sea_lbfgs_fixture, dense_ne=dense_ne_fixture, tsnekhorn3=tsnekhorn3_fixture, tsnekhorn_unrolled=tsnekhorn_unrolled_fixture, signatures=signatures_fixture, manifold=manifold_fixture, radam=radam_fixture, grad64=grad64_fixture, sampler_quality=sampler_quality_fixture, ne2_step64=ne2_step64_fixture, tsnekhorn64=tsnekhorn64_fixture)
    for name in (sys.argv[1:] or list(ALL)):  # optional: names of the fixtures to regenerate
        ALL[name]()
    print("reference version:", torchdr.__version__)
