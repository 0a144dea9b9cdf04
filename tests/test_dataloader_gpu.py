"""DataLoader inputs on the HIP path -- mirrors the reference's tests/test_dataloader.py (same results as the tensor
input for every batch size, the argument errors, estimators fed by a DataLoader)."""

import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

pytestmark = pytest.mark.gpu


@pytest.fixture
def sample_data():
    return torch.randn(1000, 32, generator=torch.Generator().manual_seed(42))


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "angular", "manhattan"])
@pytest.mark.parametrize("exclude_diag", [False, True])
def test_dataloader_equals_tensor_input(sample_data, metric, exclude_diag):
    from torchdr_amd.distance import pairwise_distances

    k = 10
    Ct, It = pairwise_distances(sample_data.cuda(), k=k, metric=metric, exclude_diag=exclude_diag, return_indices=True)
    for bs in (50, 256, 1000):
        dl = DataLoader(TensorDataset(sample_data), batch_size=bs, shuffle=False)
        Cd, Id = pairwise_distances(dl, k=k, metric=metric, exclude_diag=exclude_diag, return_indices=True)
        assert Cd.is_cuda and torch.equal(Cd, Ct) and torch.equal(Id, It)
    assert isinstance(pairwise_distances(dl, k=k, metric=metric), torch.Tensor)       # return_indices=False


def test_dataloader_argument_errors(sample_data):
    from torchdr_amd.distance import pairwise_distances

    dl = DataLoader(TensorDataset(sample_data), batch_size=100)
    with pytest.raises(ValueError, match="k cannot be None"):
        pairwise_distances(dl, k=None)
    with pytest.raises(ValueError, match="Y must be None"):
        pairwise_distances(dl, Y=torch.randn(100, 32), k=10)
    with pytest.raises(ValueError, match="only supports FAISS backend"):
        pairwise_distances(dl, k=10, backend="keops")
    with pytest.raises(ValueError, match="DataLoader is empty"):
        pairwise_distances(DataLoader(TensorDataset(sample_data[:0]), batch_size=10), k=3)


def test_estimators_and_affinities_accept_a_dataloader(sample_data):
    import torchdr_amd
    from torchdr_amd.affinity import UMAPAffinity

    # own generator: iterating a DataLoader otherwise draws its base seed from the global RNG, which the estimators
    # seed in their constructor (reference base.py:75-79) -- the embedding would start from another state
    dl = DataLoader(TensorDataset(sample_data), batch_size=128, shuffle=False, generator=torch.Generator().manual_seed(1))
    P, I = UMAPAffinity(n_neighbors=10, symmetrize=False)(dl)
    Pt, It = UMAPAffinity(n_neighbors=10, symmetrize=False)(sample_data.cuda())
    assert torch.equal(I, It) and torch.equal(P, Pt)
    Z = torchdr_amd.UMAP(n_neighbors=10, max_iter=40, random_state=0).fit_transform(dl)
    Zt = torchdr_amd.UMAP(n_neighbors=10, max_iter=40, random_state=0).fit_transform(sample_data.cuda())
    # outputs for DataLoader inputs are CPU tensors, as in the reference (utils/wrappers.py:50-54, 76-85)
    assert isinstance(Z, torch.Tensor) and Z.device.type == "cpu" and Z.shape == (1000, 2)
    assert torch.allclose(Z, Zt.cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("bs,drop_last", [(50, False), (77, False), (1000, False), (96, True), (2048, False)])
def test_batches_are_packed_as_they_arrive(sample_data, bs, drop_last):
    """`pairwise_distances(dataloader)` streams every batch into its rows of the resident block and packs the MFMA tiles
    it completes at once (`PackedPoints.from_batches`): the images and norms equal those of the concatenated block bit for
    bit, whatever the batch size (tiles that straddle two batches, a ragged last tile, `drop_last`)."""
    from torchdr_amd.distance.base import PackedPoints
    from torchdr_amd.utils.dataloader import stream_dataloader_packed

    dl = DataLoader(TensorDataset(sample_data), batch_size=bs, shuffle=False, drop_last=drop_last)
    X, packed = stream_dataloader_packed(dl, "cuda", "sqeuclidean")
    n = (len(sample_data) // bs) * bs if drop_last else len(sample_data)
    assert packed is not None and packed.n == n and packed.X is X and X.shape == (n, 32)
    ref = PackedPoints(sample_data[:n].cuda())
    assert torch.equal(X, sample_data[:n].cuda()) and torch.equal(packed.norms, ref.norms)
    # a tile image = fragment blocks + 32 norms + 32 floats of padding the kernels never write or read
    stride = ref.data.numel() // ((n + 31) // 32)
    got = packed.data[: ref.data.numel()].view(-1, stride)[:, : stride - 32]
    assert torch.equal(got, ref.data.view(-1, stride)[:, : stride - 32])
    # metrics / shapes without fp32 tile images fall back to the resident block alone
    X2, p2 = stream_dataloader_packed(dl, "cuda", "manhattan")
    assert p2 is None and torch.equal(X2, X)
