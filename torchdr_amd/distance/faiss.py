"""``FaissConfig`` -- the reference's kNN backend configuration object (``torchdr/distance/faiss.py:113-221``), kept for
call-site compatibility.  There is no Faiss here: ``index_type="Flat"`` is the exact HIP search (what every other
``backend`` value selects as well), ``"IVF"`` the approximate search on the package's own cluster index
(``tdr_knn_ivf_f32``: ``nlist`` clusters, ``nprobe`` scanned per query block), and ``"IVFPQ"`` is served by the same
uncompressed IVF search (no product quantisation: 288 GB of HBM hold the rows themselves; results are at least as
accurate as the quantised index's).  ``temp_memory``, ``device``, ``M``, ``nbits`` and extra keywords are accepted and
ignored."""

from typing import Union

_INDEX_TYPES = ("Flat", "IVF", "IVFPQ")


class FaissConfig:
    def __init__(self, temp_memory: Union[str, float] = "auto", device: int = 0, index_type: str = "Flat", nprobe: int = 1,
                 nlist: int = 100, M: int = 16, nbits: int = 8, **kwargs):
        self.temp_memory, self.device = temp_memory, device
        self.index_type, self.nprobe, self.nlist = index_type, int(nprobe), int(nlist)
        self.M, self.nbits = M, nbits
        self.faiss_kwargs = dict(kwargs)

    def check_index_type(self):
        """The reference accepts any string at construction and reports it when the index is built
        (``distance/faiss.py:350-354``); the search entry points call this."""
        if self.index_type not in _INDEX_TYPES:
            raise ValueError(
                f"[TorchDR] ERROR : Index type '{self.index_type}' is not supported. "
                "Supported types are 'Flat', 'IVF', and 'IVFPQ'."
            )

    @property
    def approximate(self) -> bool:
        return self.index_type in ("IVF", "IVFPQ")

    def __repr__(self):
        extra = f", M={self.M}, nbits={self.nbits}" if self.index_type == "IVFPQ" else ""
        return (f"FaissConfig(temp_memory={self.temp_memory!r}, device={self.device}, index_type={self.index_type!r}, "
                f"nprobe={self.nprobe}, nlist={self.nlist}{extra})")


def get_dataloader_metadata(dataloader):
    """Reference ``distance/faiss.py:27-41`` (the name lives in this module there)."""
    from torchdr_amd.utils.dataloader import get_dataloader_metadata as _get

    return _get(dataloader)
