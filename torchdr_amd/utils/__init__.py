"""Host-side numeric glue mirroring the hot subset of ``torchdr/utils`` (reference
``utils/utils.py``, ``utils/wrappers.py``, ``utils/validation.py``, ``utils/sparse.py``)."""

from .misc import as_float32  # noqa: F401
from .misc import (  # noqa: F401
    bool_arg,
    seed_everything,
    set_logger,
    compute_device,
)
from .validation import (  # noqa: F401
    check_NaNs,
    check_neighbor_param,
    check_nonnegativity,
    validate_tensor,
)
from .wrappers import compile_if_requested, handle_input_output, restore_original_format, to_torch  # noqa: F401
from .dataloader import dataloader_metadata, get_dataloader_metadata, is_dataloader, materialize_dataloader  # noqa: F401
from .sparse import CSRAffinity, distributed_symmetrize_sparse, symmetrize_sparse, symmetrize_to_csr  # noqa: F401
from .numeric import (  # noqa: F401
    binary_search, cross_entropy_loss, entropy, false_position, init_bounds, kmax, kmin, logsumexp_red, matrix_transpose, square_loss,
    sum_matrix_vector, sum_red,
)
from .radam import PoincareAdamKernel, RiemannianAdam  # noqa: F401
from .manifold import EuclideanManifold, Manifold, ManifoldParameter, PoincareBallManifold  # noqa: F401
from torchdr_amd.distributed import DistributedContext  # noqa: F401,E402
