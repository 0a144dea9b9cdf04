"""Affinity-matching optimisation loop -- mirror of ``torchdr/affinity_matcher.py``
(``AffinityMatcher``: ``_fit_transform`` :201-352, ``_training_step`` :354-430, ``_init_embedding``
:493-573, optimizer / scheduler configuration :577-657, hooks :475-489).

MI355X-first differences:
  * gradients always come from the closed-form HIP kernels (K5/K6/K9); there is no autograd graph;
  * with ``optimizer="SGD"`` the update is the fused ``tdr_sgd_step_f32`` kernel; any other
    ``torch.optim`` class steps on the kernel-produced gradient (generic path);
  * the learning-rate sequence is produced by the real ``torch.optim.lr_scheduler`` object driving a
    host-side dummy parameter, so arbitrary schedulers keep their exact torch semantics;
  * the per-iteration NaN guard (reference :315) is a device-side flag checked every
    ``check_interval`` iterations and after the loop (no host sync per step).
"""

from typing import Any, Dict, Optional, Type, Union

import numpy as np
import torch

from torchdr_amd import _lib
from torchdr_amd.affinity import Affinity, SparseAffinity
from torchdr_amd.base import DRModule
from torchdr_amd.utils import check_nonnegativity, compute_device, cross_entropy_loss, square_loss, to_torch


LOSS_DICT = {"square_loss": square_loss, "cross_entropy_loss": cross_entropy_loss}


_LR_TABLE_CACHE = {}


def lr_schedule_table(scheduler_class, scheduler_kwargs, lr0, lr_as_tensor, n_steps):
    """Learning rate seen by optimisation step 0 .. n_steps-1, produced by the REAL torch scheduler object
    stepping a throw-away SGD optimizer (exact torch semantics for any scheduler class, including the fp32
    recursion LinearLR runs when its factors are tensors).  Pure function of its arguments -> memoised, so
    the ~40 us/step of Python scheduler overhead is paid once per configuration, not once per iteration."""
    def _key(v):
        return (float(v), str(v.dtype)) if isinstance(v, torch.Tensor) else v

    key = (scheduler_class, tuple(sorted((k, _key(v)) for k, v in (scheduler_kwargs or {}).items())), float(lr0),
           bool(lr_as_tensor), int(n_steps))
    try:
        hit = _LR_TABLE_CACHE.get(key)
    except TypeError:  # unhashable kwargs: no memoisation
        hit, key = None, None
    if hit is not None:
        return hit
    p = torch.zeros(1, requires_grad=True)
    opt = torch.optim.SGD([p], lr=torch.tensor(float(lr0)) if lr_as_tensor else float(lr0))
    if scheduler_class is None:
        table = [float(lr0)] * n_steps
    else:
        sch = scheduler_class(opt, **(scheduler_kwargs or {}))
        table = []
        for _ in range(n_steps):
            table.append(float(opt.param_groups[0]["lr"]))
            opt.step()
            sch.step()
    if key is not None:
        if len(_LR_TABLE_CACHE) > 64:
            _LR_TABLE_CACHE.clear()
        _LR_TABLE_CACHE[key] = table
    return table


class AffinityMatcher(DRModule):
    # the reference hands torch.optim a TENSOR learning rate here (affinity_matcher.py:621) but a plain
    # float in NeighborEmbedding (neighbor_embedding/base.py:342); scheduler arithmetic follows suit.
    _lr_as_tensor = True

    def __init__(self, affinity_in: Affinity, affinity_out: Optional[Affinity] = None,
                 kwargs_affinity_out: Optional[Dict] = None, n_components: int = 2,
                 loss_fn: str = "square_loss", kwargs_loss: Optional[Dict] = None,
                 optimizer: Union[str, Type[torch.optim.Optimizer]] = "Adam",
                 optimizer_kwargs: Optional[Dict] = None, lr: Union[float, str] = 1e0,
                 scheduler: Optional[Union[str, Type[torch.optim.lr_scheduler.LRScheduler]]] = None,
                 scheduler_kwargs: Optional[Dict] = None, min_grad_norm: float = 1e-7, max_iter: int = 1000,
                 init: Union[str, torch.Tensor, np.ndarray] = "pca", init_scaling: float = 1e-4,
                 device: str = "auto", backend=None, verbose: bool = False,
                 random_state: Optional[float] = None, check_interval: int = 50, compile: bool = False,
                 encoder=None, **kwargs):
        super().__init__(n_components=n_components, device=device, backend=backend, verbose=verbose,
                         random_state=random_state, compile=compile, **kwargs)
        if encoder is not None:
            raise NotImplementedError("[torchdr_amd] parametric (encoder) embeddings are out of scope.")
        self.optimizer = optimizer
        self.optimizer_kwargs = optimizer_kwargs
        self.lr = lr
        self.min_grad_norm = min_grad_norm
        self.check_interval = check_interval
        self.max_iter = max_iter
        self.scheduler = scheduler
        self.scheduler_kwargs = scheduler_kwargs
        if loss_fn not in LOSS_DICT:
            raise ValueError(f"[TorchDR] ERROR : Loss function {loss_fn} not supported.")
        self.loss_fn = loss_fn
        self.kwargs_loss = kwargs_loss
        self.init = init
        self.init_scaling = init_scaling
        if not isinstance(affinity_in, Affinity) and not affinity_in == "precomputed":
            raise ValueError('[TorchDR] affinity_in must be an Affinity instance or "precomputed".')
        self.affinity_in = affinity_in
        if isinstance(self.affinity_in, Affinity):
            self.affinity_in._pre_processed = True
            self.affinity_in.compile = self.compile
        if affinity_out is not None:
            if not isinstance(affinity_out, Affinity):
                raise ValueError("[TorchDR] ERROR : affinity_out must be an Affinity instance when not None.")
            affinity_out._pre_processed = True
        self.affinity_out = affinity_out
        self.kwargs_affinity_out = kwargs_affinity_out
        self.encoder = None
        self.n_iter_ = torch.tensor(-1, dtype=torch.long)

    # ------------------------------------------------------------------------------------------
    def _compute_affinity_in(self, X):
        """Reference :269-286.  Subclasses may override to ask the affinity for a device-native
        layout (UMAP asks for CSR)."""
        if isinstance(self.affinity_in, SparseAffinity):
            affinity_matrix, nn_indices = self.affinity_in(X, return_indices=True)
            self.register_buffer("NN_indices_", nn_indices, persistent=False)
        else:
            affinity_matrix = self.affinity_in(X)
        self.register_buffer("affinity_in_", affinity_matrix, persistent=False)

    def _fit_transform(self, X: torch.Tensor, y: Optional[Any] = None) -> torch.Tensor:
        self.n_samples_in_, self.n_features_in_ = X.shape
        self.device_ = compute_device(X, self.device)
        X = X.to(self.device_)
        if X.dtype != torch.float32 and not (X.dtype == torch.float64 and getattr(self, "_float64_loop", False)):
            raise NotImplementedError(
                f"[torchdr_amd] only float32 inputs are supported by the HIP path of {type(self).__name__} (got {X.dtype})."
            )
        self._dtype = X.dtype

        from torchdr_amd.utils.phases import phase

        self._start_pca_prefetch(X)
        self.on_affinity_computation_start()
        if self.affinity_in == "precomputed":   # reference :259-271
            if self.verbose:
                self.logger.info("----- Using precomputed affinity matrix -----")
            if self.n_features_in_ != self.n_samples_in_:
                raise ValueError(
                    '[TorchDR] ERROR : When affinity_in="precomputed" the input '
                    "X in fit must be a tensor of lazy tensor of shape "
                    "(n_samples, n_samples)."
                )
            check_nonnegativity(X)
            self.register_buffer("affinity_in_", X, persistent=False)
        else:
            if self.verbose:
                self.logger.info(
                    f"----- Computing the input affinity matrix with {self.affinity_in.__class__.__name__} -----"
                )
            self._compute_affinity_in(X)
        with phase("loop layout (epoch counters, exclusion tables, RCCL context)"):
            self.on_affinity_computation_end()

        if self.verbose:
            self.logger.info("----- Optimizing the embedding -----")
        with phase("init embedding"):
            self._init_embedding(X)
            self._set_learning_rate()
            self._configure_optimizer()
            self._configure_scheduler()
        del X

        self._nan_flag = torch.zeros(1, dtype=torch.int32, device=self.device_)
        with phase("loop"):
            self._run_training_loop()
            self._raise_if_nan()
        self.clear_memory()
        return self.embedding_

    def _start_pca_prefetch(self, X):
        """``init="pca"`` depends on X alone: its kernels (column means, Gram matrix, Jacobi eigensolver, projection -- none
        reads anything back) are enqueued on a side stream now and run under the kNN search; `_init_embedding` waits for
        that stream.  3.5 ms of the N = 1M fit."""
        self._pca_prefetch = None
        if not (_opt("PCA_PREFETCH") and _opt("PCA_EIGH") in ("top", "jacobi") and isinstance(self.init, str) and self.init == "pca"):
            return
        if not X.is_cuda or X.dtype != torch.float32 or X.shape[1] > 256 or self.n_components > 4:
            return
        main = torch.cuda.current_stream(X.device)
        side = _PREFETCH_STREAMS.get(X.device)
        if side is None:
            side = _PREFETCH_STREAMS[X.device] = torch.cuda.Stream(device=X.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            emb = pca_scores(X, self.n_components)
        self._pca_deterministic = pca_scores.deterministic
        self._pca_prefetch = (side, emb)

    def _run_training_loop(self):
        """Reference :288-352: step, hooks, and every ``check_interval`` iterations the NaN / convergence checks."""
        for step in range(self.max_iter):
            self.n_iter_.fill_(step)
            self.on_training_step_start()
            self._training_step()
            self.on_training_step_end()
            if step % self.check_interval == 0:
                self._raise_if_nan()
                if self._converged(step, self._grad_norm()):
                    break

    def _converged(self, step: int, grad_norm: float) -> bool:
        if self.verbose:
            self.logger.info(f"[{step}/{self.max_iter}] Grad norm: {grad_norm:.2e} | LR: {self._current_lr():.2e}")
        if grad_norm < self.min_grad_norm:
            if self.verbose:
                self.logger.info(f"Convergence reached at iter {step} with grad norm: {grad_norm:.2e}.")
            return True
        return False

    def _raise_if_nan(self):
        if not getattr(self, "_fused_sgd", True):
            # torch.optim path: nothing wrote the flag -- scan the embedding here (every check_interval iterations)
            if bool(torch.isnan(self.embedding_.detach()).any()):
                self._nan_flag.fill_(int(self.n_iter_) + 1)
        if getattr(self, "world_size", 1) > 1:
            # a rank steps only its own rows: agree on the flag so that all ranks raise together instead of one raising
            # while the others wait in the next collective
            from torchdr_amd.parallel import allreduce_max_

            allreduce_max_(self._nan_flag)
        it = int(self._nan_flag.item())
        if it != 0:
            raise ValueError(f"[TorchDR] ERROR AffinityMatcher : NaNs in the embeddings at iter {it - 1}.")
        ctx = getattr(self, "_rccl_ctx", None)
        if ctx is not None and hasattr(ctx, "failed"):
            # peer exchange (csrc/tdr_peerx.hip): a bounded wait for a peer's rows ran out -- rows of an older generation may
            # have been used since.  Agreed between the ranks like the NaN flag, so that all of them raise.
            bad = torch.tensor([1 if ctx.failed() else 0], dtype=torch.int32, device=self.embedding_.device)
            from torchdr_amd.parallel import allreduce_max_

            allreduce_max_(bad)
            if int(bad.item()) != 0:
                # every rank is here (the flag was agreed): drop the exchange everywhere -- its error word, flags and
                # generation counters die with it -- so that later fits of this process do not inherit the failure
                from torchdr_amd.parallel import PeerExchange

                PeerExchange.retire_shared()
                self._rccl_ctx = None
                raise RuntimeError("[torchdr_amd] peer exchange: a rank's rows did not arrive within the wait limit "
                                   "(a rank fell behind by more than parallel.PeerExchange.WAIT_LIMIT spins, or was lost); the exchange has "
                                   "been retired for this process group: later fits use the RCCL all-gather.")

    # ------------------------------------------------------------------------------------------
    def _training_step(self):
        """Reference :354-430 with closed-form gradients (optimizer step, then scheduler step = advance
        in the learning-rate table).  ``_compute_gradients`` returns either the chunk's rows
        (``rows_only=True``: UMAP, only row i moves) or a full (N, c) buffer that other ranks also scatter
        into (LargeVis / TSNE).

        Multi-GPU, rows-only + fused SGD: each rank steps ITS rows and the updated rows are all-gathered
        (1/W of the reference's zero-padded gradient all-reduce, :395-413, and no full-size optimizer pass);
        otherwise the gradient is all-gathered / all-reduced (:425) and every rank steps the full embedding."""
        # closed-form gradients (reference :381-413) when the class provides them -- an own `_compute_gradients`, or
        # `_use_closed_form_gradients = True` with the base class's combination of the attractive / repulsive hooks --
        # else autograd of the loss (:414-425)
        own = not getattr(type(self)._compute_gradients, "_is_base", False)
        if not (own or getattr(self, "_use_closed_form_gradients", False)):
            return self._autograd_training_step()
        out = self._compute_gradients()
        if isinstance(out, tuple):
            grad, rows_only = out
        else:   # the reference's contract: a bare tensor holding the gradient rows of this rank's chunk (:384-413)
            grad, rows_only = out, True
            if grad is None:        # :383 `if gradients is not None`
                self._lr_pos += 1
                return None
            if grad.shape[0] != getattr(self, "chunk_size_", grad.shape[0]):
                raise RuntimeError(
                    f"Gradient size mismatch in distributed mode: expected {self.chunk_size_} gradients for chunk but "
                    f"_compute_gradients() returned {grad.shape[0]}"
                )
            grad = grad.detach().to(getattr(self, "_dtype", torch.float32)).contiguous()
        world = getattr(self, "world_size", 1)
        if world > 1 and rows_only and self._fused_sgd:
            from torchdr_amd.parallel import allgather_rows_

            c0, c1 = self.chunk_start_, self.chunk_start_ + self.chunk_size_
            rows = self.embedding_[c0:c1]
            self._last_grad = grad
            self._last_grad_is_chunk = True
            self._sgd_kernel(rows, grad, chunk=True)
            ctx = getattr(self, "_rccl_ctx", None)
            if ctx is not None:
                # on-stream RCCL through the C library; the context moves rows of floats, a float64 embedding goes through
                # as twice as many float columns (a byte copy either way)
                Z = self.embedding_
                ctx.allgather_rows_(Z.view(torch.float32) if Z.dtype == torch.float64 else Z)
            else:
                allgather_rows_(self.embedding_, c0, self.chunk_size_, world)
            self._lr_pos += 1
            return None
        if world > 1:
            from torchdr_amd.parallel import allgather_rows, allreduce_

            if rows_only:
                grad = allgather_rows(grad, self.n_samples_in_, world)
            else:
                allreduce_(grad)  # :425
        self._last_grad = grad
        self._last_grad_is_chunk = False
        self._optimizer_step(grad)
        self._lr_pos += 1
        return None

    def _grad_norm(self) -> float:
        """2-norm of the full gradient (reference :331-342), reduced over ranks when only chunks are held."""
        g = self._last_grad
        if getattr(self, "_last_grad_is_chunk", False):
            from torchdr_amd.parallel import allreduce_

            sq = (g * g).sum().reshape(1)
            allreduce_(sq)
            return float(sq.sqrt().item())
        return float(g.norm(2).item())

    def _compute_gradients(self):
        raise NotImplementedError("[TorchDR] ERROR : _compute_gradients method must be implemented.")

    _compute_gradients._is_base = True

    # ---- autograd mode (reference :418-425) ---------------------------------------------------------------------------
    def _autograd_training_step(self):
        """A subclass that defines a LOSS instead of closed-form gradients (``_compute_loss``, or for neighbour
        embeddings ``_compute_attractive_loss`` / ``_compute_repulsive_loss``) is differentiated by PyTorch autograd on
        the device tensors, exactly as the reference does; the optimizer step then runs as for the kernel-produced
        gradients.  The estimators shipped here never take this path (they all carry closed forms)."""
        emb = self.embedding_
        if not emb.requires_grad:
            emb.requires_grad_(True)
        emb.grad = None
        loss = self._compute_loss()
        loss.backward()
        grad = emb.grad.detach()
        if getattr(self, "world_size", 1) > 1:
            from torchdr_amd.parallel import allreduce_

            allreduce_(grad)  # :425
        self._last_grad, self._last_grad_is_chunk = grad, False
        if self._fused_sgd:
            emb.requires_grad_(False)
            self._sgd_kernel(emb, grad.contiguous())
        else:
            lr = self._current_lr()
            for g in self.optimizer_.param_groups:
                g["lr"] = lr
            self.optimizer_.step()
            self.optimizer_.zero_grad(set_to_none=True)
        self._lr_pos += 1
        return loss

    def _compute_loss(self):
        """Reference :433-459: ``loss_fn(affinity_in_, affinity_out(embedding_))``; ``affinity_out`` has to be
        differentiable (an Affinity written with torch ops -- the HIP affinities of this package have no autograd)."""
        if self.affinity_out is None:
            raise ValueError("[TorchDR] ERROR : affinity_out is not set. Set it or implement _compute_loss method.")
        from torchdr_amd.affinity import LogAffinity

        kwargs_affinity_out = dict(self.kwargs_affinity_out or {})
        kwargs_loss = dict(self.kwargs_loss or {})
        if self.loss_fn == "cross_entropy_loss" and isinstance(self.affinity_out, LogAffinity):
            kwargs_affinity_out.setdefault("log", True)
            kwargs_loss.setdefault("log", True)
        Q = self.affinity_out(self.embedding_, **kwargs_affinity_out)
        return LOSS_DICT[self.loss_fn](self.affinity_in_, Q, **kwargs_loss)

    def _current_lr(self) -> float:
        return self._lr_table[min(self._lr_pos, len(self._lr_table) - 1)]

    def _sgd_kernel(self, Z, grad, chunk=False):
        """Fused torch.optim.SGD(momentum) step on the rows ``Z`` (a contiguous view of the embedding)."""
        mom = float(self._sgd_momentum)
        if mom != 0.0 and self._momentum_buf is None:
            self._momentum_buf = torch.empty_like(Z)
            first = 1
        else:
            first = 0
        holder = getattr(self, "optimizer_", None)
        if isinstance(holder, torch.optim.Optimizer):
            holder.param_groups[0]["lr"] = self._current_lr()
        _lib.check(
            _lib.fn("tdr_sgd_step", Z.dtype)(_lib.ptr(Z), _lib.ptr(grad), _lib.ptr(self._momentum_buf), Z.numel(),
                                             self._current_lr(), mom, first, _lib.ptr(self._nan_flag), int(self.n_iter_),
                                             _lib.stream_ptr()),
            "tdr_sgd_step",
        )

    def _optimizer_step(self, grad):
        if self._fused_sgd:
            self._sgd_kernel(self.embedding_, grad)
        else:
            lr = self._current_lr()
            for g in self.optimizer_.param_groups:
                g["lr"] = lr
            self.embedding_.grad = grad
            self.optimizer_.step()
            self.optimizer_.zero_grad(set_to_none=True)

    # ------------------------------------------------------------------------------------------
    def on_affinity_computation_start(self):
        pass

    def on_affinity_computation_end(self):
        pass

    def on_training_step_start(self):
        pass

    def on_training_step_end(self):
        pass

    # ------------------------------------------------------------------------------------------
    def _init_embedding(self, X):
        """Reference :493-573 (A.5): Z0 = init_scaling * E / std(E[:, 0])."""
        n = X.shape[0]
        dev = getattr(self, "device_", None) or X.device    # callable on its own, as in the reference's unit tests
        if isinstance(self.init, (torch.Tensor, np.ndarray)):
            emb = to_torch(self.init).to(device=dev, dtype=X.dtype)
        elif self.init in ("normal", "random"):
            emb = torch.randn((n, self.n_components), device=dev, dtype=X.dtype)
        elif self.init == "pca":
            pre, self._pca_prefetch = getattr(self, "_pca_prefetch", None), None
            if pre is not None:
                main = torch.cuda.current_stream(pre[1].device)
                main.wait_stream(pre[0])
                emb = pre[1]
                emb.record_stream(main)
            else:
                emb = pca_scores(X, self.n_components)
                self._pca_deterministic = pca_scores.deterministic
        else:
            raise ValueError(f"[TorchDR] ERROR : init {self.init} not supported in {self.__class__.__name__}.")
        self.embedding_ = (self.init_scaling * emb / emb[:, 0].std()).contiguous()
        return self.embedding_

    def _set_params(self):
        """Reference :577-583 (there is no encoder on this path: the embedding itself is what is optimised)."""
        self.params_ = [{"params": self.embedding_}]
        return self.params_

    def _set_learning_rate(self):
        if self.lr == "auto":
            if self.verbose:
                self.logger.warning("lr set to 'auto' without any implemented rule. Setting lr=1.0 by default.")
            self.lr_ = 1.0
        else:
            self.lr_ = self.lr

    def _resolve_optimizer_kwargs(self):
        return self.optimizer_kwargs or {}

    def _configure_optimizer(self):
        kwargs = dict(self._resolve_optimizer_kwargs())
        if isinstance(self.optimizer, str):
            try:
                optimizer_class = getattr(torch.optim, self.optimizer)
            except AttributeError:
                raise ValueError(f"[TorchDR] ERROR: Optimizer '{self.optimizer}' not found in torch.optim.")
        else:
            if not issubclass(self.optimizer, torch.optim.Optimizer):
                raise ValueError(
                    "[TorchDR] ERROR: optimizer must be a string (name of an optimizer in "
                    "torch.optim) or a subclass of torch.optim.Optimizer."
                )
            optimizer_class = self.optimizer
        self._fused_sgd = optimizer_class is torch.optim.SGD and set(kwargs) <= {"momentum"}
        self._momentum_buf = None
        if self._fused_sgd:
            # the step itself is the fused kernel (`_sgd_kernel`); `optimizer_` stays a genuine torch.optim.SGD over the
            # embedding -- hyper-parameters and the current learning rate can be read from it as from the reference's --
            # whose `step` is never called
            self._sgd_momentum = kwargs.get("momentum", 0.0)
            self.optimizer_ = optimizer_class([self.embedding_], lr=float(self.lr_), **kwargs)
        else:
            self.embedding_.requires_grad_(True)
            self.optimizer_ = optimizer_class([self.embedding_], lr=float(self.lr_), **kwargs)
        return self.optimizer_

    def _configure_scheduler(self, n_iter: Optional[int] = None):
        """Resolve the scheduler class (reference :625-657) and tabulate the learning rates it produces."""
        n_iter = n_iter or self.max_iter
        if not hasattr(self, "optimizer_"):
            raise ValueError(
                "[TorchDR] ERROR : optimizer not set. Please call _configure_optimizer before _configure_scheduler."
            )
        scheduler_class = None
        if self.scheduler is not None:
            if isinstance(self.scheduler, str):
                try:
                    scheduler_class = getattr(torch.optim.lr_scheduler, self.scheduler)
                except AttributeError:
                    raise ValueError(
                        f"[TorchDR] ERROR: Scheduler '{self.scheduler}' not found in torch.optim.lr_scheduler."
                    )
            else:
                if not issubclass(self.scheduler, torch.optim.lr_scheduler.LRScheduler):
                    raise ValueError(
                        "[TorchDR] ERROR: scheduler must be a string (name of a scheduler in "
                        "torch.optim.lr_scheduler) or a subclass of torch.optim.lr_scheduler.LRScheduler."
                    )
                scheduler_class = self.scheduler
        # the learning rates come from the table below (the same scheduler class stepping a throw-away optimizer, memoised);
        # `scheduler_` is an instance bound to `optimizer_`, as in the reference, and is never stepped
        opt = getattr(self, "optimizer_", None)
        if scheduler_class is not None and isinstance(opt, torch.optim.Optimizer):
            self.scheduler_ = scheduler_class(opt, **(self.scheduler_kwargs or {}))
        else:
            self.scheduler_ = None if scheduler_class is None else scheduler_class
        self._lr_table = lr_schedule_table(scheduler_class, self.scheduler_kwargs, self.lr_, self._lr_as_tensor,
                                           int(self.max_iter))
        self._lr_pos = 0
        return self.scheduler_

    def clear_memory(self):
        super().clear_memory()
        if isinstance(self.affinity_in, Affinity):
            self.affinity_in.clear_memory()
        if isinstance(self.affinity_out, Affinity):
            self.affinity_out.clear_memory()
        for attr in ["optimizer_", "scheduler_", "lr_", "_lr_table", "_momentum_buf", "_last_grad", "_nan_flag"]:
            if hasattr(self, attr):
                delattr(self, attr)
        if isinstance(self.embedding_, torch.Tensor) and self.embedding_.requires_grad:
            self.embedding_ = self.embedding_.detach()


# D x D eigenproblem of the PCA initialisation: "top" = tdr_eigh_top_f64 (the n_components leading pairs by Householder +
# multisection + inverse iteration: one workgroup, no host read, ~0.5 ms at D = 128), "jacobi" = tdr_eigh_jacobi_f64 (the whole
# decomposition, one workgroup, no host read: 12.5 ms at D = 128, 127 ms at D = 256), "library" = torch.linalg.eigh (rocSOLVER;
# reads its status word back, i.e. synchronises the host with the stream)
PCA_EIGH = "top"
# enqueue the PCA initialisation on a side stream at the start of the fit (it runs under the kNN search)
PCA_PREFETCH = True
_PREFETCH_STREAMS = {}

def _opt(name):
    """A behaviour switch of this module: the scoped override (torchdr_amd.config.options) or the module attribute."""
    from torchdr_amd import config

    return config.get(name, globals())



def pca_scores(X: torch.Tensor, n_components: int) -> torch.Tensor:
    """PCA scores U*S of the centred data with the reference's sign convention
    (spectral_embedding/pca.py:169-178 + utils/utils.py:292-298 ``svd_flip``, u-based).

    Computed from the D x D covariance eigen-decomposition instead of a thin SVD of the N x D block: column means and
    the Gram matrix of the centred block by ``tdr_pca_gram_f32`` (fp32 matrix pipe, deterministic fp64 combination), the
    leading pairs of the D x D eigenproblem by ``tdr_eigh_top_f64`` (one workgroup, no host read), and the projection by
    ``tdr_pca_project_f32``.  Same subspace and signs, O(N D^2) on the GPU.  Initialisation only -- the scores are
    rescaled to std 1e-4 right after (A.5).  D > 256 or more than 4 components use torch ops."""
    n, d = X.shape
    pca_scores.deterministic = False
    if d > 256 or n_components > 4 or not X.is_cuda or X.dtype != torch.float32:
        mean = X.mean(0, keepdim=True)
        Xc = X - mean
        cov = Xc.T @ Xc
        evals, evecs = torch.linalg.eigh(cov.double())
        V = evecs[:, -n_components:].flip(1).to(X.dtype)
        E = Xc @ V
    else:
        if X.stride(1) != 1:
            X = X.contiguous()
        L = _lib.lib()
        mean = torch.empty(d, dtype=torch.float32, device=X.device)
        G = torch.empty((d, d), dtype=torch.float64, device=X.device)
        ws_floats = int(L.tdr_pca_gram_workspace_floats(n, d))
        ws = torch.empty(ws_floats, dtype=torch.float32, device=X.device)
        _lib.check(L.tdr_pca_gram_f32(_lib.ptr(X), n, d, X.stride(0), _lib.ptr(mean), _lib.ptr(G), _lib.ptr(ws), ws_floats,
                                      _lib.stream_ptr()), "tdr_pca_gram_f32")
        # same bits on every rank of a row-sharded fit only on this branch (ordered fp64 combination of the Gram tiles, one-
        # workgroup eigensolver): NeighborEmbedding._init_embedding skips the reference's broadcast (:421) when it was taken
        pca_scores.deterministic = _opt("PCA_EIGH") in ("top", "jacobi")
        if _opt("PCA_EIGH") == "top" and n_components <= d:    # leading pairs only, one workgroup, no host read (csrc/tdr_prep.hip)
            evals = torch.empty(n_components, dtype=torch.float64, device=X.device)
            evecs = torch.empty((d, n_components), dtype=torch.float64, device=X.device)
            ews = torch.empty(d * d, dtype=torch.float64, device=X.device)
            _lib.check(L.tdr_eigh_top_f64(_lib.ptr(G), d, n_components, _lib.ptr(evals), _lib.ptr(evecs), _lib.ptr(ews),
                                          _lib.stream_ptr()), "tdr_eigh_top_f64")
            V = evecs.to(torch.float32).contiguous()
        elif _opt("PCA_EIGH") in ("top", "jacobi"):    # the whole decomposition, one workgroup, no host read
            evals = torch.empty(d, dtype=torch.float64, device=X.device)
            evecs = torch.empty((d, d), dtype=torch.float64, device=X.device)
            ews = torch.empty(2 * d * d, dtype=torch.float64, device=X.device)
            _lib.check(L.tdr_eigh_jacobi_f64(_lib.ptr(G), d, _lib.ptr(evals), _lib.ptr(evecs), _lib.ptr(ews), _lib.stream_ptr()),
                       "tdr_eigh_jacobi_f64")
            V = evecs[:, :n_components].to(torch.float32).contiguous()       # top components, descending
        else:
            evals, evecs = torch.linalg.eigh(G)
            V = evecs[:, -n_components:].flip(1).to(torch.float32).contiguous()  # top components, descending
        E = torch.empty((n, n_components), dtype=torch.float32, device=X.device)
        _lib.check(L.tdr_pca_project_f32(_lib.ptr(X), n, d, X.stride(0), _lib.ptr(mean), _lib.ptr(V), n_components, _lib.ptr(E),
                                         _lib.stream_ptr()), "tdr_pca_project_f32")
    idx = E.abs().argmax(0)
    signs = torch.sign(E[idx, torch.arange(E.shape[1], device=E.device)])
    signs = torch.where(signs == 0, torch.ones_like(signs), signs)
    return E * signs[None, :]


pca_scores.deterministic = False   # set by every call: True when the deterministic kernel path produced the scores
