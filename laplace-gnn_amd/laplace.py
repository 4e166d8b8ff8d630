"""Thin ``Laplace`` front for the curvature path: the caller of the backend.

Mirrors what the fit path of the reference does and nothing else:
``Laplace`` factory (laplace/laplace.py:13-47), ``BaseLaplace.__init__`` parameter filter and lazy
backend (laplace/baselaplace.py:94-190), ``ParametricLaplace.fit`` (:778-854), ``KronLaplace``
(:1556-1610 incl. the ``override=False`` discounting), ``DiagLaplace`` (:1848-1857), and a
last-layer ``FullLLLaplace`` (laplace/lllaplace.py:369-378 intent).  Attributes after ``fit``:
``H``, ``H_facs`` (kron), ``loss``, ``n_data``, ``n_outputs``, ``n_params``, ``mean``,
``model.output_size``.

Data parallelism (new; the reference is single process).  The accumulated factors + loss are summed
with ONE all-reduce of a flat fp32 buffer (RCCL over xGMI on GPUs, gloo in the CPU tests).  What is
dealt to the ranks of the ``torch.distributed`` group depends on the structure:
* kron: a batch's SAMPLES are never split -- the B factors have cross-sample terms inside a batch
  (SURVEY.md 0.5), so the reference loader's boundaries are part of the result -- but
  ``B = sum_batches sum_classes g_c^T g_c``, so the unit of work is (batch, class column) and the T*C
  units are dealt in balanced contiguous runs (``data.units_of_rank``); backends without class-range
  support get whole batches round-robin (batch t -> rank t mod world);
* diag / last-layer full: plain sums over samples -> every rank takes its slice of every batch.
"""
from __future__ import annotations

from math import log, pi
from typing import Any

import os

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.utils import parameters_to_vector

from .curvature import HipGGN
from .data import TensorBatchLoader
from .matrix import Kron, KronDecomposed


def _dist_info(process_group):
    if process_group is None and not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(process_group), dist.get_world_size(process_group)


def all_reduce_flat_(tensors: list[torch.Tensor], process_group=None):
    """Sum a list of tensors over the group with one collective on a flat buffer (in place)."""
    rank, world = _dist_info(process_group)
    if world == 1:
        return
    big = [t for t in tensors if t.numel() >= (1 << 24)]
    if len(tensors) == 1 or big:
        # a large contiguous tensor (the 2.3 GB last-layer H at the products shape) is reduced where it lives --
        # no second buffer, no copy back; the small rest (loss scalars) still travels as one flat message
        for t in big if big else tensors:
            if not t.is_contiguous():
                raise ValueError("all_reduce_flat_ needs contiguous tensors")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        tensors = [t for t in tensors if t.numel() < (1 << 24)] if big else []
        if not tensors:
            return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class BaseLaplace:
    def __init__(self, model: nn.Module, likelihood: str, sigma_noise: float = 1.0, prior_precision: float = 1.0,
                 prior_mean: float = 0.0, temperature: float = 1.0, enable_backprop: bool = False,
                 dict_key_x: str = "input_ids", dict_key_y: str = "labels", backend: type | None = None,
                 backend_kwargs: dict[str, Any] | None = None, asdl_fisher_kwargs: dict[str, Any] | None = None):
        if likelihood not in ("classification", "regression", "reward_modeling"):
            raise ValueError(f"Invalid likelihood type {likelihood}")
        self.model, self.likelihood = model, likelihood
        # only do Laplace on params that require grad; the fork also drops 'adj' / 'norms' (baselaplace.py:118-122)
        self.params, self.is_subset_params = [], False
        for k, p in model.named_parameters():
            if p.requires_grad and "adj" not in k and "norms" not in k:
                self.params.append(p)
            else:
                self.is_subset_params = True
        self.n_params = sum(p.numel() for p in self.params)
        self.n_layers = len(self.params)
        if sigma_noise != 1 and likelihood != "regression":
            raise ValueError("Sigma noise != 1 only available for regression.")
        self.sigma_noise, self.temperature = sigma_noise, temperature
        self.prior_precision, self.prior_mean = prior_precision, prior_mean
        self.enable_backprop = enable_backprop
        self.dict_key_x, self.dict_key_y = dict_key_x, dict_key_y
        self._backend = None
        self._backend_cls = HipGGN if backend is None else backend
        self._backend_kwargs = dict() if backend_kwargs is None else backend_kwargs
        self._asdl_fisher_kwargs = dict() if asdl_fisher_kwargs is None else asdl_fisher_kwargs
        self.loss, self.n_outputs, self.n_data = 0.0, 0, 0

    @property
    def _device(self):
        # (walking named_parameters() costs 7 us per call on the host: a Cora-shaped fit asks twice and takes 0.2 ms in all)
        p = self.params[0] if self.params else next(self.model.parameters())
        return p.device

    @property
    def backend(self):
        if self._backend is None:
            lik = "classification" if self.likelihood == "reward_modeling" else self.likelihood
            self._backend = self._backend_cls(self.model, lik, dict_key_x=self.dict_key_x,
                                              dict_key_y=self.dict_key_y, **self._backend_kwargs)
        return self._backend

    @property
    def _H_factor(self):
        return 1 / self.sigma_noise ** 2 / self.temperature

    @property
    def prior_precision_diag(self) -> torch.Tensor:
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float32, device=self._device).reshape(-1)
        if pp.numel() == 1:
            return pp.expand(self.n_params).clone()
        if pp.numel() == self.n_params:
            return pp
        if pp.numel() == self.n_layers:
            return torch.cat([d.expand(p.numel()) for d, p in zip(pp, self.params)])
        raise ValueError("Mismatch of prior and model. Diagonal, scalar, or per-layer prior.")


def _is_zero_number(v) -> bool:
    return isinstance(v, (int, float)) and v == 0


def _adjacency_candidates(eng, candidates, sym: bool):
    """Candidate pairs (i, j) of ``neg_marglik_adj_grad`` -> (a, b, accumulator) in the propagation matrix's coordinates, or
    None.  The device kernels index rows with these pairs: validated before the cast to int32 (one host round trip; this is not
    the per-batch path).  Stored pairs are not candidates: their gradient comes back with the stored entries (a GCN stores its
    diagonal, gnn/models/models.py:23; GraphSAGE's diagonal entries are genuine non-edges of the reference's dense adj.grad,
    gnn/models/models.py:47)."""
    if candidates is None:
        return None
    ci, cj = candidates[0].to(eng.device).to(torch.int64), candidates[1].to(eng.device).to(torch.int64)
    Nn = eng.num_nodes
    if ci.numel():
        if bool(((ci < 0) | (ci >= Nn) | (cj < 0) | (cj >= Nn)).any()):
            raise ValueError(f"candidate pairs must index nodes in [0, {Nn})")
        sr, sc = eng.export_adj()  # row-major sorted
        skey, ckey = sr * Nn + sc, ci * Nn + cj
        pos = torch.searchsorted(skey, ckey).clamp(max=max(skey.numel() - 1, 0))
        if skey.numel() and bool((skey[pos] == ckey).any()):
            raise ValueError("candidate pairs must not be stored entries of the adjacency")
    if sym:  # (adj + adj^T) / 2 feeds the model: both orientations are needed
        ci, cj = torch.cat([ci, cj]), torch.cat([cj, ci])
    # entry (i, j) of the adjacency is entry (a = j, b = i) of the GCN propagation matrix D A^T D and entry
    # (a = i, b = j) of GraphSAGE's A / rowsum
    ca, cb = (cj, ci) if eng.kind == "gcn" else (ci, cj)
    return (ca.to(torch.int32).contiguous(), cb.to(torch.int32).contiguous(),
            torch.zeros(ci.shape[0], dtype=torch.float32, device=eng.device))


class ParametricLaplace(BaseLaplace):
    def _init_H(self):
        raise NotImplementedError

    def _curv_closure(self, X, y, N):
        raise NotImplementedError

    def _reduce_tensors(self) -> list[torch.Tensor]:
        raise NotImplementedError

    def fit(self, train_loader, override: bool = True, progress_bar: bool = False, process_group=None) -> None:
        """laplace/baselaplace.py:778-854, plus the round-robin batch sharding described above."""
        if override:
            self._init_H()
            self.loss = 0
            self.n_data = 0
        if self.model.training:
            self.model.eval()
        # the flattened parameters, recomputed on every fit exactly as laplace/baselaplace.py:800 does (one concatenation
        # kernel).  Version counters cannot stand in for it: ``p.data.add_()`` / ``p.data.copy_()`` do not bump
        # ``p._version``, and ``load_state_dict`` / ``la.mean = ...`` replace the vector behind any key
        self.mean = parameters_to_vector(self.params).detach()
        dev = self._device
        # the reference finds the output width with a forward pass of one sample (laplace/baselaplace.py:806-816); a model
        # that states it (laplace_gnn_amd.models.BaseGNN.n_outputs) saves that pass' launches -- a Cora-shaped fit is
        # launch bound
        self.n_outputs = getattr(self.model, "n_outputs", None)
        if self.n_outputs is None:
            X0 = next(iter(train_loader))[0]
            with torch.no_grad():
                try:
                    out = self.model(X0[:1].to(dev))
                except (TypeError, AttributeError):
                    out = self.model(X0.to(dev))
            self.n_outputs = out.shape[-1]
        setattr(self.model, "output_size", self.n_outputs)
        N = len(train_loader.dataset)
        rank, world = _dist_info(process_group)
        # override=False inside a distributed job: self.H already holds the previous fits' ALL-REDUCED curvature; it must
        # not take part in this fit's all-reduce (it would be counted once per rank): accumulate into a fresh buffer
        H_prev = None
        if not override and world > 1 and torch.is_tensor(getattr(self, "H", None)):
            H_prev, self.H = self.H, torch.zeros_like(self.H)
        loss = None  # stays None for in-place backends (they add the raw loss to a buffer of their own)
        plan = self._shard_plan(train_loader, rank, world)
        for t, (X, y) in enumerate(train_loader):
            todo = plan(t, X.shape[0])
            if todo is None:
                continue
            X, y = X.to(dev), y.to(dev)
            if isinstance(todo, slice):  # sample-additive structures: this rank's slice of the batch
                X, y = X[todo], y[todo]
                loss_batch, H_batch = self._curv_closure(X, y, N=N)
            elif todo is True:  # the whole batch
                loss_batch, H_batch = self._curv_closure(X, y, N=N)
            else:  # (class_begin, class_end): an exact additive share of the batch's KFAC factors
                loss_batch, H_batch = self._curv_closure(X, y, N=N, classes=todo)
            if H_batch is not None or torch.is_tensor(loss_batch):  # (in-place backends return (0.0, None): nothing to add)
                loss = loss_batch if loss is None else loss + loss_batch
            self._accumulate(H_batch)
        self._finish_accumulate()
        if getattr(self, "_on_accumulated", None) is not None:
            self._on_accumulated()  # measurement hook (bench.py): end of the accumulate loop, before reduce / decompose
        # invalid node ids / labels are flagged on the device; one sync per fit.  (Deferring this check until the
        # decomposition is queued was measured: 94.9 -> 95.9 ms per arxiv-shaped fit, the host running ahead costs more
        # than the ~1 ms of launches it hides.)
        if hasattr(self.backend, "check_async_errors") and not getattr(self, "_capturing", False):
            self.backend.check_async_errors()  # (a graph capture cannot hold the synchronisation: _fit_graph_try checks after the replay)
        if world > 1:
            phase = getattr(self, "_on_phase", None)  # measurement hook (bench.py): brackets the factor all-reduce
            if phase is not None:
                phase("reduce_begin")
            loss = torch.as_tensor(0.0 if loss is None else loss, dtype=torch.float32, device=dev)
            all_reduce_flat_(self._reduce_tensors() + [loss], process_group)
            if phase is not None:
                phase("reduce_end")
            self._after_reduce()
        self._post_accumulate()  # e.g. the last-layer pair accumulators are placed into H here, once per fit
        if H_prev is not None:
            H_prev += self.H
            self.H = H_prev
        if loss is not None:
            self.loss = loss if _is_zero_number(self.loss) else self.loss + loss
        self.n_data += N

    # how the work of one fit is dealt to the ranks of the process group
    _sample_additive = False  # True: H is a plain sum over samples (diag, last-layer full) -> slice batches

    def _shard_plan(self, train_loader, rank: int, world: int):
        """Returns plan(t, M) -> None (skip) | True (whole batch) | slice (samples) | (c0, c1) (class range)."""
        if world == 1:
            return lambda t, M: True
        if self._sample_additive:
            def plan(t, M):
                lo, hi = M * rank // world, M * (rank + 1) // world
                return slice(lo, hi) if hi > lo else None
            return plan
        return lambda t, M: True if t % world == rank else None  # whole batches round-robin

    # hooks so that subclasses can keep H in a flat buffer
    def _accumulate(self, H_batch):
        self.H += H_batch

    def _finish_accumulate(self):
        pass

    def _after_reduce(self):
        pass

    def _post_accumulate(self):
        pass

    # ---- posterior samples and the sampling ("nn", link_approx="mc") predictive the GNN driver evaluates with
    # (gnn/marglik_training.py:338-352 -> laplace/baselaplace.py:1183-1199): theta ~ N(mean, P^-1), one full-graph
    # forward per sample through the HIP engine (the in-place parameter writes invalidate its cache), softmax, mean.
    def sample(self, n_samples: int = 100, generator: torch.Generator | None = None, eps: torch.Tensor | None = None):
        """``eps`` (n_samples x n_params standard normal draws) may be passed instead of a generator so that
        results are reproducible across devices."""
        if eps is None:
            eps = torch.randn(n_samples, self.n_params, device=self._device, generator=generator)
        return self.mean.reshape(1, self.n_params) + self._scale_samples(eps.to(self._device))

    def _scale_samples(self, eps):
        raise NotImplementedError

    @torch.no_grad()
    def __call__(self, x, pred_type: str = "glm", link_approx: str = "probit", n_samples: int = 100,
                 diagonal_output: bool = False, generator: torch.Generator | None = None,
                 eps: torch.Tensor | None = None, **kwargs):
        """Posterior predictive (laplace/baselaplace.py:975-1072), classification.

        ``pred_type="glm"`` (default, with ``link_approx`` in probit / mc / bridge / bridge_norm): linearised model,
        ``f ~ N(f_mu, J P^-1 J^T)`` from explicit Jacobians (:1123-1158, :570-665).  ``pred_type="nn"`` with
        ``link_approx="mc"``: parameter samples, one forward each (:1183-1199).  ``eps`` replaces the random draws
        (standard normal, [n_samples, n_params] for "nn", [n_outputs, n_samples] for "glm"/"mc")."""
        if pred_type not in ("glm", "nn"):
            raise ValueError("Only glm and nn supported as prediction types.")
        if link_approx not in ("mc", "probit", "bridge", "bridge_norm"):
            raise ValueError(f"Unsupported link approximation {link_approx}.")
        if pred_type == "nn" and link_approx != "mc":
            raise ValueError("Only mc link approximation is supported for nn prediction type.")
        x = x.to(self._device)
        regression = self.likelihood == "regression"
        if pred_type == "glm":
            if regression:  # (f_mu [M, C], f_var [M, C, C]) of the linearised model (:620-624)
                f_mu, f_var = self._glm_predictive_distribution(x)
                return f_mu, (torch.diagonal(f_var, dim1=-2, dim2=-1) if diagonal_output else f_var)
            return self._glm_forward_call(x, link_approx, n_samples, diagonal_output, generator, eps)
        from torch.nn.utils import vector_to_parameters
        outs = []
        samples = self.sample(n_samples, generator=generator, eps=eps)
        try:
            for theta in samples:
                vector_to_parameters(theta, self.params)
                f = self.model(x)
                outs.append(f if regression else torch.softmax(f, dim=-1))
        finally:
            vector_to_parameters(self.mean.clone(), self.params)  # the parameters must not alias self.mean afterwards
        outs = torch.stack(outs)
        if regression:  # mean and variance over the parameter samples (:1060-1066)
            return outs.mean(dim=0), outs.var(dim=0)
        return outs.mean(dim=0)

    # ---- GLM predictive ("next" row 8(f)-3) ---------------------------------------------------------------------
    def functional_variance(self, Js: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _matrix_free_operands(self, out_map=None):
        """Operands of ``GraphEngine.glm_variance`` for this posterior (mapped by ``out_map`` = E [Cm, C] where they depend on
        the outputs), or None when the matrix-free route does not apply."""
        return None

    def _glm_variance_matrix_free(self, x, out_map=None):
        """(f_mu [M, C], diag f_var [M, C]) without Jacobians (csrc/predictive.hip), or None: 2-layer GCN / GraphSAGE models
        with a ReLU, hidden width <= 256, classification, Kronecker or diagonal posterior over all weights.  ``out_map`` =
        E [Cm, C]: the variances of E f instead ([M, Cm])."""
        eng = getattr(self.backend, "engine", None)
        if (eng is None or not hasattr(eng, "glm_variance") or getattr(eng, "kind", None) not in ("gcn", "sage") or len(eng.dims) != 3
                or getattr(eng, "has_extras", False)
                or eng.dims[1] > 256 or getattr(eng, "_bind_opts", ("relu",))[0] != "relu" or self.likelihood != "classification"):
            return None
        ops = self._matrix_free_operands(out_map)
        if ops is None:
            return None
        eng.set_likelihood("classification")
        return eng.glm_variance(x, **ops) if out_map is None else eng.glm_variance(x, out_map=out_map, **ops)

    def _bridge_moments_matrix_free(self, x):
        """What the Laplace bridge reads of the C x C predictive covariance S (laplace/baselaplace.py:637-661) -- its diagonal,
        its row sums S 1 and its total 1^T S 1 -- without Jacobians, or None.  One matrix-free pass over the 2 C + 1 outputs
        E f, E = [I ; 1^T / C ; I + 1 1^T / C]: with m = mean(f), var(f_c + m) - var(f_c) - var(m) = 2 cov(f_c, m) = 2 (S 1)_c / C
        and 1^T S 1 = C^2 var(m) (polarisation against the MEAN keeps the three terms at the same magnitude)."""
        C = self.n_outputs
        eye = torch.eye(C, device=self._device)
        ones = torch.full((1, C), 1.0 / C, device=self._device)
        fast = self._glm_variance_matrix_free(x, out_map=torch.cat([eye, ones, eye + ones], dim=0))
        if fast is None:
            return None
        f_mu, v = fast
        diag, vm, vsum = v[:, :C], v[:, C:C + 1], v[:, C + 1:]
        return f_mu, diag, 0.5 * C * (vsum - diag - vm), (C * C) * vm.squeeze(1)

    _JACOBIAN_BYTES_MAX = 1 << 30  # above this size of Js [M, C, P] the full-covariance links go matrix free where they can

    def _glm_covariance_matrix_free(self, x, budget_floats: int = 1 << 30):
        """(f_mu [M, C], f_var [M, C, C]) without Jacobians, or None: the C (C + 1) / 2 variances var(f_c + f_c') of one
        matrix-free pass (lgnn_glm_variance_mapped) give every entry by polarisation, cov(f_c, f_c') = (var(f_c + f_c') -
        var(f_c) - var(f_c')) / 2.  Evaluation nodes go in chunks sized so that the rotated-row table of a chunk (needed nodes x
        pairs x hidden floats) stays near ``budget_floats``.  Off-diagonal entries carry the rounding of the three variances
        (absolute ~1e-6 of the diagonal): what ``link_approx="mc"`` needs of a model whose Jacobians do not fit."""
        C = self.n_outputs
        iu = torch.triu_indices(C, C, device=self._device)
        if iu.shape[1] > 4096:
            return None
        E = torch.zeros(iu.shape[1], C, device=self._device)
        ar = torch.arange(iu.shape[1], device=self._device)
        E[ar, iu[0]] += 1.0
        E[ar, iu[1]] += 1.0
        eng = getattr(self.backend, "engine", None)
        if eng is None:
            return None
        per_node = iu.shape[1] * eng.dims[1] * (eng.nnz / max(eng.num_nodes, 1) + 2.0)  # table floats per evaluation node
        chunk = max(1, int(budget_floats / per_node))
        dsel = (iu[0] == iu[1]).nonzero().squeeze(1)  # positions of the pairs (c, c): var(2 f_c) = 4 var(f_c)
        mus, covs = [], []
        for s0 in range(0, x.shape[0], chunk):
            fast = self._glm_variance_matrix_free(x[s0:s0 + chunk], out_map=E)
            if fast is None:
                return None
            f_mu, v = fast
            d = 0.25 * v[:, dsel]
            off = 0.5 * (v - d[:, iu[0]] - d[:, iu[1]])
            cov = torch.zeros(v.shape[0], C, C, device=v.device)
            cov[:, iu[0], iu[1]] = off
            cov[:, iu[1], iu[0]] = off
            cov[:, dsel.new_tensor(range(C)), dsel.new_tensor(range(C))] = d
            mus.append(f_mu)
            covs.append(cov)
        return torch.cat(mus), torch.cat(covs)

    def _glm_predictive_distribution(self, x, diagonal_output: bool = False):
        """(f_mu [M, C], f_var [M, C, C]) (laplace/baselaplace.py:1123-1158); ``diagonal_output``: f_var [M, C], matrix free
        where the model family allows it."""
        if diagonal_output:
            fast = self._glm_variance_matrix_free(x)
            if fast is not None:
                return fast
        # Jacobian route (models with res / norm, deeper models, full posteriors): the reference forms Js [M, C, P] for all
        # evaluation nodes at once (laplace/baselaplace.py:1123-1158), which cannot exist at scale; here the nodes go in chunks
        # whose Jacobians stay under _JACOBIAN_BYTES_MAX -- the same numbers, any M
        per_node = max(1, self.n_outputs * self.n_params * 4)
        chunk = max(1, min(int(x.shape[0]), self._JACOBIAN_BYTES_MAX // per_node))
        mus, fvars = [], []
        for s0 in range(0, max(int(x.shape[0]), 1), chunk):
            Js, f_mu = self.backend.jacobians(x[s0:s0 + chunk], enable_backprop=False)
            if Js.shape[1:] != (self.n_outputs, self.n_params):
                raise ValueError("Invalid Jacobians shape for Laplace posterior approx.")
            f_var = self.functional_variance(Js)
            if diagonal_output:
                f_var = torch.diagonal(f_var, dim1=-2, dim2=-1)
            mus.append(f_mu)
            fvars.append(f_var)
            del Js
        if len(mus) == 1:
            return mus[0], fvars[0]
        return torch.cat(mus), torch.cat(fvars)

    def _glm_forward_call(self, x, link_approx, n_samples, diagonal_output, generator, eps):
        """laplace/baselaplace.py:570-665 (classification branches)."""
        if link_approx == "probit":  # reads the diagonal of f_var only (laplace/baselaplace.py:610-616)
            f_mu, f_var_diag = self._glm_predictive_distribution(x, diagonal_output=True)
            kappa = 1 / torch.sqrt(1.0 + pi / 8 * f_var_diag)
            return torch.softmax(kappa * f_mu, dim=-1)
        if link_approx == "mc":
            if diagonal_output:  # samples of N(f_mu, diag f_var): the diagonal is all that is read (:667-709)
                f_mu, f_var_diag = self._glm_predictive_distribution(x, diagonal_output=True)
                return self._glm_predictive_samples(f_mu, f_var_diag, n_samples, False, generator, eps).mean(dim=0)
            big = x.shape[0] * self.n_outputs * self.n_params * 4 > self._JACOBIAN_BYTES_MAX
            full = self._glm_covariance_matrix_free(x) if big else None
            if full is not None:
                f_mu, f_var = full
                # polarised off-diagonals can leave a node's matrix indefinite in the last bits: those nodes (if any) take the
                # Jacobian route, 32 at a time
                info = torch.linalg.cholesky_ex(f_var).info
                bad = info.nonzero().squeeze(1)
                for b0 in range(0, bad.numel(), 32):
                    sel = bad[b0:b0 + 32]
                    f_var[sel] = self._glm_predictive_distribution(x[sel])[1]
            else:
                f_mu, f_var = self._glm_predictive_distribution(x)
            return self._glm_predictive_samples(f_mu, f_var, n_samples, diagonal_output, generator, eps).mean(dim=0)
        # Laplace bridge with zero-mean correction (:630-660): reads the diagonal, the row sums and the total of f_var only
        moments = self._bridge_moments_matrix_free(x)
        if moments is None:
            f_mu, f_var = self._glm_predictive_distribution(x)
            moments = (f_mu, torch.diagonal(f_var, dim1=1, dim2=2), f_var.sum(-1), f_var.sum(dim=(1, 2)))
        f_mu, f_var_diag, rows, total = moments  # (symmetric f_var: f_var.sum(-1) == f_var.sum(-2))
        f_mu = f_mu - rows * f_mu.sum(-1).reshape(-1, 1) / total.reshape(-1, 1)
        f_var_diag = f_var_diag - rows * rows / total.reshape(-1, 1)
        K = f_mu.size(-1)
        if link_approx == "bridge_norm":
            f_var_diag_mean = f_var_diag.mean(dim=1)
            f_var_diag_mean = f_var_diag_mean / torch.as_tensor([K / 2], device=self._device).sqrt()
            f_mu = f_mu / f_var_diag_mean.sqrt().unsqueeze(-1)
            f_var_diag = f_var_diag / f_var_diag_mean.unsqueeze(-1)
        sum_exp = torch.exp(-f_mu).sum(dim=1).unsqueeze(-1)
        alpha = (1 - 2 / K + f_mu.exp() / K ** 2 * sum_exp) / f_var_diag
        return torch.nan_to_num(alpha / alpha.sum(dim=1).unsqueeze(-1), nan=1.0)

    def _glm_predictive_samples(self, f_mu, f_var, n_samples, diagonal_output=False, generator=None, eps=None):
        """softmax of samples of N(f_mu, f_var): laplace/baselaplace.py:667-709 + utils.normal_samples (:329-369)."""
        if diagonal_output:
            f_var = torch.diagonal(f_var, dim1=1, dim2=2)
        C = f_mu.shape[1]
        if eps is None:
            eps = torch.randn((C, n_samples), device=f_mu.device, dtype=f_mu.dtype, generator=generator)
        eps = eps.to(f_mu.device)
        if f_var.ndim == 2:
            scaled = f_var.sqrt().unsqueeze(-1) * eps.unsqueeze(0)
        else:
            scaled = torch.matmul(torch.linalg.cholesky(f_var), eps.unsqueeze(0))
        return torch.softmax((f_mu.unsqueeze(-1) + scaled).permute((2, 0, 1)), dim=-1)

    @property
    def log_likelihood(self) -> torch.Tensor:
        """laplace/baselaplace.py:895-922: -H_factor * loss for classification."""
        factor = -self._H_factor
        if self.likelihood == "regression":
            c = self.n_data * self.n_outputs * log(self.sigma_noise * (2 * pi) ** 0.5)
            return factor * self.loss - c
        return factor * self.loss

    @property
    def scatter(self) -> torch.Tensor:
        delta = self.mean - torch.as_tensor(self.prior_mean, device=self.mean.device, dtype=self.mean.dtype)
        return (delta * self.prior_precision_diag) @ delta

    @property
    def log_det_prior_precision(self) -> torch.Tensor:
        return self.prior_precision_diag.log().sum()

    @property
    def log_det_ratio(self) -> torch.Tensor:
        return self.log_det_posterior_precision - self.log_det_prior_precision

    def log_marginal_likelihood(self, prior_precision=None, sigma_noise=None) -> torch.Tensor:
        """laplace/baselaplace.py:938-973: ``log_lik - 0.5 (log det P - log det P_0 + scatter)``; passing
        ``prior_precision`` overwrites the current value (useful when iterating on the marginal likelihood)."""
        if prior_precision is not None:
            self.prior_precision = prior_precision
        if sigma_noise is not None:
            if self.likelihood != "regression":
                raise ValueError("Can only change sigma_noise for regression.")
            self.sigma_noise = sigma_noise
        return self.log_likelihood - 0.5 * (self.log_det_ratio + self.scatter)


    # ---- serialisation (laplace/baselaplace.py:1314-1368): plain dict of tensors and scalars, torch.save-able -----
    def state_dict(self) -> dict:
        if getattr(self, "H", None) is None:
            raise AttributeError("Laplace not fit. Run fit() first.")
        return {"mean": self.mean, "H": self.H, "loss": self.loss, "prior_mean": self.prior_mean,
                "prior_precision": self.prior_precision, "sigma_noise": self.sigma_noise, "n_data": self.n_data,
                "n_outputs": self.n_outputs, "likelihood": self.likelihood, "temperature": self.temperature,
                "enable_backprop": self.enable_backprop, "cls_name": self.__class__.__name__}

    def load_state_dict(self, state_dict: dict) -> None:
        if self.__class__.__name__ != state_dict["cls_name"]:
            raise ValueError("Loading a wrong Laplace type. Make sure `subset_of_weights` and `hessian_structure` "
                             "are correct!")
        if self.n_params is not None and len(state_dict["mean"]) != self.n_params:
            raise ValueError("Attempting to load Laplace with different number of parameters than the model.")
        if self.likelihood != state_dict["likelihood"]:
            raise ValueError("Different likelihoods detected!")
        import warnings
        if self.temperature != state_dict["temperature"]:
            warnings.warn("Different `temperature` parameters detected. Some calculation might be off!")
        for k in ("mean", "H", "loss", "prior_mean", "prior_precision", "sigma_noise", "n_data", "n_outputs",
                  "likelihood", "temperature", "enable_backprop"):
            setattr(self, k, state_dict[k])
        setattr(self.model, "output_size", self.n_outputs)

    # ---- post-hoc prior precision tuning (laplace/baselaplace.py:342-560) ------------------------------------------
    def optimize_prior_precision(self, pred_type: str = "glm", method: str = "marglik", n_steps: int = 100,
                                 lr: float = 1e-1, init_prior_prec=1.0, prior_structure: str = "scalar",
                                 val_loader=None, loss=None, log_prior_prec_min: float = -4,
                                 log_prior_prec_max: float = 4, grid_size: int = 100, link_approx: str = "probit",
                                 n_samples: int = 100, verbose: bool = False, progress_bar: bool = False) -> None:
        """``method="marglik"``: Adam on the log prior precision against ``-log_marginal_likelihood`` (the fitted
        factors and their decomposition are constants, only ``logdet(P)``, ``logdet(P_0)`` and the scatter term move:
        :444-463).  ``method="gridsearch"``: the value of a log-spaced grid with the lowest validation loss of the
        chosen predictive (:464-560; default loss = mean negative log likelihood of the predictive, the
        reference's RunningNLLMetric)."""
        if method == "marglik":
            pp = init_prior_prec if torch.is_tensor(init_prior_prec) else torch.tensor(float(init_prior_prec))
            pp = pp.detach().to(self._device, torch.float32).reshape(-1)
            if pp.numel() == 1 and prior_structure != "scalar":
                n = {"layerwise": self.n_layers, "diag": self.n_params}.get(prior_structure)
                if n is None:
                    raise ValueError(f"Invalid prior structure {prior_structure}.")
                pp = torch.full((n,), float(pp), device=self._device)
            log_pp = pp.log().clone().requires_grad_(True)
            opt = torch.optim.Adam([log_pp], lr=lr)
            with torch.enable_grad():
                for _ in range(n_steps):
                    opt.zero_grad()
                    neg = -self.log_marginal_likelihood(prior_precision=log_pp.exp())
                    neg.backward()
                    opt.step()
            self.prior_precision = log_pp.detach().exp()
        elif method == "gridsearch":
            if val_loader is None:
                raise ValueError("gridsearch requires a validation set DataLoader")
            if loss is None:
                def loss(probs, y):  # mean NLL of the predictive (laplace/utils/metrics.py RunningNLLMetric)
                    return float(torch.nn.functional.nll_loss(probs.clamp_min(1e-30).log(), y, reduction="mean"))
            best, best_pp = None, None
            for pp in torch.logspace(log_prior_prec_min, log_prior_prec_max, grid_size):
                self.prior_precision = pp
                try:
                    outs, ys = [], []
                    for X, y in val_loader:
                        outs.append(self(X.to(self._device), pred_type=pred_type, link_approx=link_approx,
                                         n_samples=n_samples))
                        ys.append(y.to(self._device))
                    val = loss(torch.cat(outs), torch.cat(ys))
                except RuntimeError:
                    val = float("inf")
                if best is None or val < best:
                    best, best_pp = val, pp
            self.prior_precision = best_pp
        else:
            raise ValueError("For now only marglik and gridsearch is implemented.")
        if verbose:
            print(f"Optimized prior precision is {self.prior_precision}.")


class KronLaplace(ParametricLaplace):
    _key = ("all", "kron")

    def __init__(self, model, likelihood, *args, damping: bool = False, cache_decompositions: bool = True, **kwargs):
        self.damping = damping
        self.H_facs = None
        # reuse the eigendecomposition of a GCN's first input covariance across fits of the same model (``_decompose_cache``)
        self.cache_decompositions = bool(cache_decompositions)
        super().__init__(model, likelihood, *args, **kwargs)

    def _init_H(self):
        self.H = Kron.init_from_model(self.params, self._device)
        self._flat = None
        self._flat_used = False

    def state_dict(self) -> dict:  # laplace/baselaplace.py:1664-1677: the factors, not their decomposition
        sd = super().state_dict()
        sd["H"] = self.H_facs.kfacs
        return sd

    def load_state_dict(self, state_dict: dict) -> None:
        super().load_state_dict(state_dict)
        # a fresh container WITHOUT the tied-factor hint: the loaded bias blocks need not repeat their weight block's B
        # (regression: sqrt(.5) B vs .5 B), so decompose compares the tensors
        self.H_facs = Kron(state_dict["H"])
        self.H = self.H_facs.decompose(damping=self.damping)

    def _inplace_backend(self) -> bool:
        """The in-place (flat buffer, class-range capable) fast path exists for the GGN factors of the HIP backend; the
        empirical / MC Fisher variants go through ``backend.kron`` like any other backend."""
        be = self.backend
        return hasattr(be, "kron_accumulate_") and getattr(be, "_kron_fisher_type", "type2") == "type2"

    def _curv_closure(self, X, y, N, classes=None):
        be = self.backend
        if self._inplace_backend():  # in-place fast path of the HIP backend
            if self._flat is None:
                self._flat = be.new_kfac_buffers() if hasattr(be, "new_kfac_buffers") else be.engine.new_kfac_buffers()
            first = not self._flat_used
            self._flat_used = True
            _, views, loss_buf = self._flat
            be.kron_accumulate_(views, loss_buf, X, y, N, classes=classes)
            if first and getattr(self, "_early_ok", False):
                self._snapshot_large_input_factors(views)
            return 0.0, None
        return be.kron(X, y, N=N, **self._asdl_fisher_kwargs)

    def _snapshot_large_input_factors(self, views):
        """The input covariances do not depend on the batch: every batch adds the same ``in_l^T in_l / N_train`` (the forward
        pass is cached per weight version), so after the FIRST batch of an ``override=True`` fit a factor already has its
        final eigenvectors and its eigenvalues up to the scale ``T``.  Factors too large for the one-workgroup
        decomposition (more than 256 rows: a GraphSAGE's ``A_1`` is 2 H wide, 512 at the arxiv shape) cost a library
        ``syevd`` of ~12 ms -- a chain of a few thousand tiny launches that used to run after the batch loop.  They are
        copied here, behind the first batch; ``_finish_accumulate`` -- the host has queued every batch by then and the
        device is tens of milliseconds behind -- starts their decomposition on a side stream, where it overlaps the
        remaining batches; ``fit`` hands the result (eigenvalues rescaled by the ratio of the traces, a device scalar: no
        synchronisation) to ``decompose``."""
        from .matrix import _SMALL_EIG
        big = [l for l, (A, _) in enumerate(views) if A.is_cuda and A.shape[0] > _SMALL_EIG]
        if not big:
            return
        snaps = [views[l][0].clone() for l in big]  # A_l after one batch (the buffers keep accumulating)
        ev = torch.cuda.Event()
        ev.record()
        self._early_pending = (big, snaps, [a.diagonal().sum() for a in snaps], ev)

    def _finish_accumulate(self):
        pending, self._early_pending = getattr(self, "_early_pending", None), None
        if pending is None:
            return
        from .matrix import symeig_batched_hip
        big, snaps, traces, ev = pending
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=snaps[0].device)
        self._side_stream.wait_event(ev)
        with torch.cuda.stream(self._side_stream):
            pairs = symeig_batched_hip(snaps)
        for a in snaps:
            a.record_stream(self._side_stream)
        self._early = {l: (lam, Q, tr) for l, (lam, Q), tr in zip(big, pairs, traces)}

    def _shard_plan(self, train_loader, rank: int, world: int):
        """Backends that can restrict a call to a range of class columns get the balanced (batch, class)
        decomposition of ``data.units_of_rank``; others fall back to whole batches round-robin."""
        be = self.backend
        if world == 1 or not (self._inplace_backend() and hasattr(be, "num_classes")):
            return super()._shard_plan(train_loader, rank, world)
        from .data import units_of_rank
        mine = {}
        for t, c0, c1 in units_of_rank(len(train_loader), be.num_classes, rank, world):
            mine[t] = (c0, c1)
        return lambda t, M: mine.get(t)

    def _accumulate(self, H_batch):
        if H_batch is not None:
            self.H += H_batch

    def _reduce_tensors(self):
        be = self.backend
        if self._inplace_backend():
            if self._flat is None:  # a rank without local batches still takes part in the all-reduce
                self._flat = be.new_kfac_buffers() if hasattr(be, "new_kfac_buffers") else be.engine.new_kfac_buffers()
            flat, views, loss_buf = self._flat
            if flat is not None:
                return [flat]  # [A_0|B_0|...|loss] already is one buffer
            return [t for pair in views for t in pair] + [loss_buf]
        # every block is reduced on its own: a ring sums the elements of a bias block's B and of its weight block's B in
        # different rank orders, so the two may differ in the last bit afterwards -- and a rank without local batches
        # would still carry the "tied" hint of its zero-initialised container.  All ranks compare instead.
        self.H._tied = frozenset()
        return [Hi for F in self.H.kfacs for Hi in F]

    def _fold_flat(self):
        if self._flat is not None:
            flat, views, loss_buf = self._flat
            be = self.backend
            self.H = self.H + be.pack_kron(views)
            self._flat_loss = be.factor * loss_buf[0].clone()
            self._flat = None
        else:
            self._flat_loss = None

    @staticmethod
    def _rescale_factors(kron: Kron, factor: float) -> Kron:
        for F in kron.kfacs:
            if len(F) == 2:
                F[1] *= factor
        return kron

    def fit(self, train_loader, override: bool = True, progress_bar: bool = False, process_group=None) -> None:
        rank_world = _dist_info(process_group)
        self._early, self._early_pending = {}, None
        self._side_stream = getattr(self, "_side_stream", None)
        # (see _snapshot_large_input_factors; single process, fresh factors, and no exact-key cache for the same factor)
        self._early_ok = (override and rank_world[1] == 1 and self.cache_decompositions and self._inplace_backend()
                          and os.environ.get("LGNN_NO_EARLY_EIG", "") in ("", "0")
                          and self._decompose_cache(train_loader, override)[0] is None)
        if override:
            self.H_facs = None
        if self.H_facs is not None:
            n_data_old, n_data_new = self.n_data, len(train_loader.dataset)
            self._init_H()
            self.H_facs = self._rescale_factors(self.H_facs, n_data_old / (n_data_old + n_data_new))
        super().fit(train_loader, override=override, progress_bar=progress_bar, process_group=process_group)
        self._fold_flat()
        if self._flat_loss is not None:
            self.loss = self.loss + self._flat_loss
        if self.H_facs is None:
            self.H_facs = self.H
        else:
            self.H = self._rescale_factors(self.H, n_data_new / (n_data_new + n_data_old))
            self.H_facs += self.H
        cache, keys = self._decompose_cache(train_loader, override)
        if self._early and cache is None:
            torch.cuda.current_stream(self._side_stream.device).wait_stream(self._side_stream)
            cache, keys = {}, {}
            main = torch.cuda.current_stream(self._side_stream.device)
            for l, (lam, Q, tr) in self._early.items():  # block 2 l = layer l's weight block, factor 1 = its A
                lam.record_stream(main)  # (allocated on the side stream, read on this one from here on)
                Q.record_stream(main)
                final = self.H_facs.kfacs[2 * l][1]
                cache[("early", l)] = (lam * (final.diagonal().sum() / tr), Q)
                keys[(2 * l, 1)] = ("early", l)
        self._early = {}
        self.H = self.H_facs.decompose(damping=self.damping, process_group=process_group, cache=cache, cache_keys=keys)

    def _decompose_cache(self, train_loader, override: bool):
        """A GCN's first input covariance is ``factor^(1/2) (T / N_train) X^T X``: a function of the feature tensor, the batch
        count and N_train only -- not of the weights, not of the adjacency -- yet at the Cora shape (1 433 x 1 433) its
        decomposition is 30 ms of a 2 ms fit, and the fork's structure-learning loop refits after every adjacency step
        (gnn/marglik_training.py:197-216).  The eigenpairs are kept ON THE MODEL OBJECT under an exact key: the identity and
        version counter of the feature tensor (a strong reference is held, so the storage cannot be recycled under the key),
        T, N_train and the likelihood factor.  Nothing is compared numerically; ``cache_decompositions=False`` opts out."""
        be = self._backend
        eng = getattr(be, "engine", None) if be is not None else None
        if (not self.cache_decompositions or not override or eng is None or getattr(eng, "kind", None) != "gcn"
                or not hasattr(eng, "feature_token") or getattr(be, "_kron_fisher_type", "type2") != "type2"):
            return None, None
        store = self.model.__dict__.setdefault("_lgnn_eig_cache", {})
        key = ("gcn_A0", eng.feature_token(), len(train_loader), len(train_loader.dataset), float(be.factor))
        return store, {(0, 1): key}

    @property
    def posterior_precision(self) -> KronDecomposed:
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float32, device=self._device).reshape(-1)
        if pp.numel() not in (1, self.n_layers):
            raise ValueError("Prior precision for Kron either scalar or per-layer.")
        return self.H * self._H_factor + pp

    # ---- fp64 eigenpairs of the fitted factors: what divides by (eigenvalue products + prior precision) reads these -----
    _FP64_EIG_MAX = 512  # factors up to this size are re-decomposed in fp64 (larger ones keep the fit's own eigenpairs)

    def _eigh64(self, block: int, k: int):
        """(eigenvalues, eigenvectors) in fp64 of factor ``k`` of block ``block`` of the fitted (fp32) factors, cached per fit.
        The fit's own decomposition is fp32 (hand-written tridiagonalisation + divide and conquer): its eigenvalues carry an
        ABSOLUTE error of ~1e-6 of the factor's largest one, i.e. the small ones -- which dominate ``1 / (f lB_i lA_j + delta)``
        -- have no correct digit, and the float atomics of the accumulation make that error differ from run to run.  An fp64
        decomposition of the same fp32 factor has none of it: what remains is the factors' own last-bit noise (~1e-7).
        Factors of up to 128 rows on a host core (numpy: 0.1 ms; the device needs 2 ms of launches for a 64 x 64 matrix, and
        torch's CPU eigh spins up the whole intra-op pool), up to ``_FP64_EIG_MAX`` rows on the device, larger ones (Cora's
        1 433 x 1 433 X^T X: 40 ms in fp64) keep the fit's pairs."""
        store = self.__dict__.setdefault("_eig64", {})
        if store.get("for") is not self.H:  # a new fit / load_state_dict made a new decomposition object
            store.clear()
            store["for"] = self.H
        t = self.H_facs.kfacs[block][k]
        key = (t.data_ptr(), tuple(t.shape))
        if key not in store:
            n = t.shape[0]
            if n <= 128:
                import numpy as np
                lam, Q = np.linalg.eigh(t.double().cpu().numpy())
                pair = (torch.from_numpy(lam).to(t.device), torch.from_numpy(Q).to(t.device))
            elif n <= self._FP64_EIG_MAX:
                pair = torch.linalg.eigh(t.double())
            else:
                pair = (self.H.eigenvalues[block][k].double(), self.H.eigenvectors[block][k].double())
            store[key] = (pair[0].clamp(min=0.0), pair[1])  # (symeig's clamp, laplace/utils/utils.py:193-226)
        return store[key]

    def _refined_decomposition(self) -> KronDecomposed:
        """The fit's ``KronDecomposed`` with every factor's eigenpairs replaced by the fp64 ones (rounded to fp32: each
        eigenvalue then has a RELATIVE error of 6e-8).  What the predictive and the samples invert; the marginal likelihood
        -- the per-epoch quantity -- keeps the fit's own pairs (its logdet is insensitive: 3.7e-7 measured)."""
        H = self.H
        if not isinstance(H, KronDecomposed):
            return H
        store = self.__dict__.setdefault("_eig64", {})
        if store.get("for") is H and "refined" in store:
            return store["refined"]
        vals, vecs = [], []
        for i, ls in enumerate(H.eigenvalues):
            pairs = [self._eigh64(i, k) for k in range(len(ls))]
            vals.append([lam.float() for lam, _ in pairs])
            vecs.append([Q.float() for _, Q in pairs])
        store["refined"] = KronDecomposed(vecs, vals, H.deltas, H.damping)
        return store["refined"]

    @property
    def posterior_precision_refined(self) -> KronDecomposed:
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float32, device=self._device).reshape(-1)
        if pp.numel() not in (1, self.n_layers):
            raise ValueError("Prior precision for Kron either scalar or per-layer.")
        return self._refined_decomposition() * self._H_factor + pp

    # ---- 8(f)-4: what the GNN driver differentiates (gnn/marglik_training.py:197-216) ------------------------------
    def _logdet_factor_gradients(self):
        """``d logdet(P) / d B_l`` and ``/ d A_l`` per layer (laplace/utils/matrix.py:371-394: ``sum log(f lB_i lA_j + delta)``
        per weight block, ``sum log(f lB_i + delta)`` per bias block, f = H_factor): ``Q diag(.) Q^T`` in the factor's own
        eigenbasis, formed in fp64 from the fp64 eigenpairs (``_eigh64``): with the fit's fp32 pairs these matrices are off
        by up to 3e-3, and that was the adjacency gradient's whole device error (3.3e-4 at a GraphSAGE with 256 hidden units,
        4.4e-6 with this; tools/adjgrad_attribution.py, profiles/r03_adjgrad_attribution.log)."""
        f = float(self._H_factor)
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float64, device=self._device).reshape(-1)
        deltas = pp.expand(self.n_layers) if pp.numel() == 1 else pp
        kf = self.H_facs.kfacs
        eigh64 = self._eigh64

        gB, gA = [], []
        for l in range(len(kf) // 2):
            (lB, QB), (lA, QA), (lBb, QBb) = eigh64(2 * l, 0), eigh64(2 * l, 1), eigh64(2 * l + 1, 0)
            den = f * torch.outer(lB, lA) + deltas[2 * l]
            cB = (f * lA.unsqueeze(0) / den).sum(dim=1)
            cA = (f * lB.unsqueeze(1) / den).sum(dim=0)
            cBb = f / (f * lBb + deltas[2 * l + 1])
            gB.append(((QB * cB) @ QB.T + (QBb * cBb) @ QBb.T).float())
            gA.append(((QA * cA) @ QA.T).float())
        return gB, gA

    def neg_marglik_adj_grad(self, train_loader, prior_precision=None, process_group=None, candidates=None):
        """``-log_marginal_likelihood()`` of this fit and its gradient w.r.t. the adjacency the model propagates with --
        what ``neg_marglik.backward()`` leaves in ``model.adj.grad`` in the reference's structure-learning loop
        (gnn/marglik_training.py:197-216), here on the stored sparsity pattern: returns ``(neg_marglik, edge_index [2, nnz],
        grad [nnz])`` over the stored entries of the 0/1 adjacency (``model.engine.export_adj()`` order; a GCN's self loops
        carry gradient 0 like the reference's overwritten diagonal).  2-layer GCN (STEGCN) and GraphSAGE (STEGraphSAGE).  ``train_loader`` must be the loader of the fit
        (same batch boundaries: the B factors depend on them).  Inside a ``torch.distributed`` job whole batches are
        dealt round-robin and the accumulators are all-reduced once.

        ``candidates`` (int64 [2, K], pairs (i, j) that are NOT stored): the reference's dense ``adj.grad`` also has an
        entry for every non-edge -- that is how its structure learning proposes new edges.  With candidates a fourth value
        is returned: ``d(-marglik) / d adj[i, j]`` for each listed pair."""
        if self.H_facs is None:
            raise AttributeError("Laplace not fitted. Run fit() first.")
        if prior_precision is not None:
            self.prior_precision = prior_precision
        if self.likelihood != "classification":
            raise NotImplementedError("adjacency gradient: classification likelihood")
        eng = self.backend.engine
        value = -self._log_marginal_likelihood64()
        gB, gA = self._logdet_factor_gradients()
        gB = [0.5 * g for g in gB]  # neg marglik = H_factor * loss + 1/2 (logdet P - logdet P_0 + scatter)
        gA = [0.5 * g for g in gA]
        grad_P = torch.zeros(eng.nnz, dtype=torch.float32, device=eng.device)
        out_bar = torch.zeros(eng.num_nodes, eng.dims[-1], dtype=torch.float32, device=eng.device)
        rank, world = _dist_info(process_group)
        sym = bool(getattr(self.model, "symmetric", False))
        cand = _adjacency_candidates(eng, candidates, sym)
        for t, (X, y) in enumerate(train_loader):
            if t % world != rank:
                continue
            eng.adjgrad_batch(X.to(eng.device), y.to(eng.device), gB, grad_P, out_bar,
                              fork_exact=getattr(self.backend, "fork_exact_seed", True), loss_scale=self._H_factor, cand=cand)
        if world > 1:
            all_reduce_flat_([grad_P, out_bar] + ([cand[2]] if cand is not None else []), process_group)
        a_scale = len(train_loader) / len(train_loader.dataset)
        rows, cols = eng.export_adj()
        if cand is None:
            grad = eng.adjgrad_finish(out_bar, gA, a_scale, grad_P)
            return value, torch.stack([rows, cols]), grad
        grad, gc = eng.adjgrad_finish(out_bar, gA, a_scale, grad_P, cand=cand)
        if sym:
            K = candidates.shape[1]
            gc = 0.5 * (gc[:K] + gc[K:])
        return value, torch.stack([rows, cols]), grad, gc

    @property
    def log_det_posterior_precision(self) -> torch.Tensor:
        return self.posterior_precision.logdet()

    def _log_marginal_likelihood64(self) -> torch.Tensor:
        """``log_marginal_likelihood()`` with the log determinant summed in fp64 over the fp64 eigenvalues (the structure-
        learning step needs them for its gradient anyway): fp32 value within 1e-6 of the fp64 oracle where the fit's own
        pairs left 5.7e-5 on one configuration of 420 (seed 346 of tools/stress_adjgrad.py, H = 256: a near-singular B_0
        whose ~200 smallest eigenvalues each carried the fp32 decomposition's absolute error into log(f lB lA + delta))."""
        if self.damping:
            return self.log_marginal_likelihood()
        f = float(self._H_factor)
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float64, device=self._device).reshape(-1)
        deltas = pp.expand(self.n_layers) if pp.numel() == 1 else pp
        ld = torch.zeros((), dtype=torch.float64, device=self._device)
        for i, F in enumerate(self.H_facs.kfacs):
            l0 = self._eigh64(i, 0)[0]
            if len(F) == 1:
                ld = ld + torch.log(f * l0 + deltas[i]).sum()
            else:
                ld = ld + torch.log(f * torch.outer(l0, self._eigh64(i, 1)[0]) + deltas[i]).sum()
        ld_prior = self.prior_precision_diag.double().log().sum()
        mean = self.mean.double()
        delta = mean - torch.as_tensor(self.prior_mean, device=mean.device, dtype=mean.dtype)
        scatter = (delta * self.prior_precision_diag.double()) @ delta
        return (self.log_likelihood - 0.5 * (ld - ld_prior + scatter)).float()

    def _matrix_free_operands(self, out_map=None):
        H = self._refined_decomposition()
        if not isinstance(H, KronDecomposed) or H.damping or len(H.eigenvalues) != 4:
            return None
        (lB0, lA0), (QB0, QA0) = H.eigenvalues[0], H.eigenvectors[0]
        (lB0b,), (QB0b,) = H.eigenvalues[1], H.eigenvectors[1]
        (lB1, lA1), (QB1, QA1) = H.eigenvalues[2], H.eigenvectors[2]
        (lB1b,), (QB1b,) = H.eigenvalues[3], H.eigenvectors[3]
        if not (QB0b is QB0 or torch.equal(QB0b, QB0)):  # the bias block must share its weight block's eigenbasis of B_0
            return None
        pp = torch.as_tensor(self.prior_precision, dtype=torch.float32, device=self._device).reshape(-1)
        d = pp.expand(4) if pp.numel() == 1 else pp
        f = self._H_factor
        S0 = torch.cat([1.0 / (f * torch.outer(lB0, lA0) + d[0]), (1.0 / (f * lB0b + d[1])).unsqueeze(1)], dim=1)
        S1 = 1.0 / (f * torch.outer(lB1, lA1) + d[2])
        if out_map is not None:  # the rows of E f: E Q_B takes Q_B's place in the last-layer blocks
            E = out_map.to(QB1)
            QB1, QB1b = E @ QB1, E @ QB1b
        kappa = (QB1b * QB1b) @ (1.0 / (f * lB1b + d[3]))
        return dict(S0=S0, S1=S1, kappa=kappa, QA0=QA0, QB0=QB0, QA1=QA1, QB1sq=QB1 * QB1)

    def _scale_samples(self, eps):  # laplace/baselaplace.py:1646-1655
        return self.posterior_precision_refined.bmm(eps, exponent=-0.5).reshape(eps.shape[0], self.n_params)

    def functional_variance(self, Js):  # laplace/baselaplace.py:1635-1636 (fp64 eigenpairs: see _eigh64)
        return self.posterior_precision_refined.inv_square_form(Js)


class DiagLaplace(ParametricLaplace):
    _key = ("all", "diag")
    _sample_additive = True  # einsum('bcp,bck,bkp->p') is a sum over b: batches may be sliced by samples

    def _init_H(self):
        # H and the raw loss share one flat buffer [H | loss]: one fill per fit, one all-reduce message
        self._hl = torch.zeros(self.n_params + 1, device=self._device)
        self.H = self._hl[:self.n_params]

    def _inplace_backend(self) -> bool:
        """The HIP backend adds a batch's diagonal GGN and raw loss straight into caller-owned buffers (no per-batch
        temporaries, no per-batch torch kernels: a Cora-shaped fit is launch bound); other backends return (loss, H)."""
        be = self.backend
        return (hasattr(be, "diag_accumulate_") and not getattr(be, "stochastic", False) and not self._asdl_fisher_kwargs
                and type(be).diag is getattr(HipGGN, "diag", None) and getattr(self, "_hl", None) is not None
                and self.H.data_ptr() == self._hl.data_ptr())

    def _curv_closure(self, X, y, N):
        if self._inplace_backend():
            self.backend.diag_accumulate_(self.H, self._hl[self.n_params:], X, y)
            return 0.0, None
        return self.backend.diag(X, y, N=N, **self._asdl_fisher_kwargs)

    def _accumulate(self, H_batch):
        if H_batch is not None:
            self.H += H_batch

    def _reduce_tensors(self):
        if getattr(self, "_hl", None) is not None and self.H.data_ptr() == self._hl.data_ptr():
            return [self._hl]
        return [self.H]

    # ---- a whole fit as ONE graph launch ---------------------------------------------------------------------------------
    # A Cora-shaped diagonal fit is ~12 kernels of 3-40 us each: the gaps between their launches (host work of this front,
    # dispatch latency) are a quarter of its wall time.  ``fit_graph = True`` (or LGNN_FIT_GRAPH=1) lets a fit that is repeated
    # unchanged -- the reference's structure-learning and hyper-parameter loops refit after every step
    # (gnn/marglik_training.py:197-224) -- be captured into a hipGraph (torch.cuda.CUDAGraph) on its second identical call and
    # replayed afterwards: the same kernels on the same buffers, forward pass included (it is recomputed on every replay, as the
    # reference recomputes it per batch), one launch.  "Identical" is an exact key: the loader object and its tensors, every
    # parameter's storage, the feature tensor's binding -- values may change (they are read by the kernels), pointers and shapes
    # may not; anything else falls back to the ordinary path.  After a replay ``H`` / ``loss`` / ``mean`` are the tensors of the
    # captured fit (overwritten by the next replay).  Single process, override=True, in-place HIP backend, classification.
    fit_graph = os.environ.get("LGNN_FIT_GRAPH", "") not in ("", "0")
    _FIT_GRAPH_WARM = 1  # ordinary fits with the same key before the capture (the workspaces reach their final size in one)

    def _fit_graph_key(self, train_loader, override, process_group):
        eng = getattr(self.backend, "engine", None)
        if (not self.fit_graph or not override or eng is None or _dist_info(process_group)[1] != 1
                or self.likelihood != "classification" or not isinstance(train_loader, TensorBatchLoader)
                or getattr(eng, "_timing_on", False) or getattr(self, "_on_phase", None) is not None):
            return None
        ts = (train_loader.indices, train_loader.labels)
        if any((not t.is_cuda) for t in ts):
            return None
        return (id(train_loader), tuple((t.data_ptr(), tuple(t.shape)) for t in ts), getattr(train_loader, "batch_size", None),
                tuple((p.data_ptr(), tuple(p.shape)) for p in self.params), eng.feature_token() if hasattr(eng, "feature_token")
                else None, getattr(eng, "_graph_edits", 0), getattr(eng, "_ws_limit", None), float(self.backend.factor),
                tuple(p for p, _ in eng._param_versions()) if hasattr(eng, "_param_versions") else None)

    def _fit_graph_try(self, train_loader, key) -> bool:
        st = self.__dict__.setdefault("_fit_graph_state", {"key": None, "seen": 0, "graph": None, "off": False})
        if st["off"]:
            return False
        if st["key"] != key:
            st.update(key=key, seen=0, graph=None, out=None)
        if st["graph"] is None:
            st["seen"] += 1
            if st["seen"] <= self._FIT_GRAPH_WARM or not self._inplace_backend():
                return False
            eng = self.backend.engine
            try:
                torch.cuda.synchronize()
                eng.invalidate()  # the captured fit always contains its forward pass
                g = torch.cuda.CUDAGraph()
                self._capturing = True
                with torch.cuda.graph(g):
                    self._fit_plain(train_loader)
                st["graph"] = g
                st["out"] = {k: getattr(self, k) for k in ("H", "_hl", "loss", "mean", "n_data", "n_outputs")}
            except Exception as exc:  # capture not possible here (e.g. a workspace still had to grow): ordinary path from now on
                st["off"] = True
                torch.cuda.synchronize()
                import warnings
                warnings.warn(f"fit_graph: capture failed, falling back to the ordinary path ({exc})")
                return False
            finally:
                self._capturing = False
        st["graph"].replay()
        for k, v in st["out"].items():
            setattr(self, k, v)
        hook = getattr(self, "_on_accumulated", None)
        if hook is not None:
            hook()
        # invalid ids / labels: the replay is not waited for (that round trip per fit is what the graph is there to save) -- the
        # sticky flags of everything that has finished are read; what this replay raises is reported by the next fit at the
        # latest, or by backend.check_async_errors()
        if hasattr(self.backend, "peek_async_errors"):
            self.backend.peek_async_errors()
        elif hasattr(self.backend, "check_async_errors"):
            self.backend.check_async_errors()
        return True

    def fit(self, train_loader, override: bool = True, progress_bar: bool = False, process_group=None) -> None:
        key = self._fit_graph_key(train_loader, override, process_group)
        if key is not None and self._fit_graph_try(train_loader, key):
            return
        self._fit_plain(train_loader, override=override, progress_bar=progress_bar, process_group=process_group)

    def _fit_plain(self, train_loader, override: bool = True, progress_bar: bool = False, process_group=None) -> None:
        if not override and getattr(self, "_hl", None) is not None:
            slot = self._hl[self.n_params:]
            if torch.is_tensor(self.loss) and self.loss.data_ptr() == slot.data_ptr():
                self.loss = self.loss.clone()  # the earlier fit handed the slot itself out
            slot.zero_()  # the loss slot restarts; H keeps the earlier fits
        super().fit(train_loader, override=override, progress_bar=progress_bar, process_group=process_group)
        if getattr(self, "_hl", None) is not None and self.H.data_ptr() == self._hl.data_ptr():
            # override=True allocated _hl for this fit: its loss slot can be handed out as it is (no kernel); a buffer that
            # the next override=False fit zeroes again must be copied
            raw = self._hl[self.n_params] if override else self._hl[self.n_params].clone()
            batch_loss = raw if self.backend.factor == 1.0 else self.backend.factor * raw
            self.loss = batch_loss if _is_zero_number(self.loss) else self.loss + batch_loss

    @property
    def posterior_precision(self) -> torch.Tensor:
        return self.H * self._H_factor + self.prior_precision_diag

    @property
    def log_det_posterior_precision(self) -> torch.Tensor:
        return self.posterior_precision.log().sum()

    def _matrix_free_operands(self, out_map=None):
        shapes = [tuple(p.shape) for p in self.params]
        if len(shapes) != 4:
            return None
        (Hd, F), _, (C, D1), _ = shapes  # (GraphSAGE: F and D1 are the widths of the concatenations)
        inv = 1.0 / self.posterior_precision
        o = 0
        w0 = inv[o:o + Hd * F].view(Hd, F); o += Hd * F
        b0 = inv[o:o + Hd].view(Hd, 1); o += Hd
        w1 = inv[o:o + C * D1].view(C, D1); o += C * D1
        b1 = inv[o:o + C]
        if out_map is not None:  # independent parameters: the variance of a combination weighs each class by E^2
            E2 = out_map.to(w1).square()
            w1, b1 = E2 @ w1, E2 @ b1
        return dict(S0=torch.cat([w0, b0], dim=1), S1=w1, kappa=b1)

    def neg_marglik_adj_grad(self, train_loader, prior_precision=None, process_group=None, candidates=None):
        """``-log_marginal_likelihood()`` of this fit and its gradient w.r.t. the adjacency -- what ``neg_marglik.backward()``
        leaves in ``model.adj.grad`` when the structure-learning loop runs with ``hessian_structure="diag"``, the shipped
        STE-GCN configuration (gnn/configs/original/stegcn_config.yaml:7; gnn/marglik_training.py:197-216; the fork's
        Jacobians keep the graph, laplace/curvature/curvature.py:89-130).  Same return values and candidate pairs as
        ``KronLaplace.neg_marglik_adj_grad``.  2-layer GCN (STEGCN, also with res / norm) and plain 2-layer GraphSAGE (STEGraphSAGE),
        classification; the diagonal GGN is a sum over samples, so
        the loader's batch boundaries do not matter and the ranks of a job split every batch by samples."""
        if self.H is None or not self.n_data:
            raise AttributeError("Laplace not fitted. Run fit() first.")
        if prior_precision is not None:
            self.prior_precision = prior_precision
        if self.likelihood != "classification":
            raise NotImplementedError("adjacency gradient: classification likelihood")
        eng = getattr(self.backend, "engine", None)
        if eng is None or not hasattr(eng, "diag_adjgrad_batch") or eng.kind not in ("gcn", "sage"):
            raise NotImplementedError("adjacency gradient under a diagonal posterior: 2-layer GCN / GraphSAGE on the HIP backend")
        value = -self.log_marginal_likelihood()
        f = self._H_factor
        gamma = (0.5 * f / self.posterior_precision).to(torch.float32).contiguous()  # d(1/2 logdet P) / dH_p
        N, Hd, F, C = eng.num_nodes, eng.dims[1], eng.dims[0], eng.dims[-1]
        grad_P = torch.zeros(eng.nnz, dtype=torch.float32, device=eng.device)
        out_bar = torch.zeros(N, C, dtype=torch.float32, device=eng.device)
        h1_bar = torch.zeros(N, Hd, dtype=torch.float32, device=eng.device)
        e_bar = torch.zeros(N, F + 1, dtype=torch.float32, device=eng.device)
        rank, world = _dist_info(process_group)
        sym = bool(getattr(self.model, "symmetric", False))
        cand = _adjacency_candidates(eng, candidates, sym)
        eng.set_likelihood("classification")
        for X, y in train_loader:
            # every term is a sum over samples: inside a job every rank takes its slice of every batch (a Cora-shaped loader has
            # ONE batch); a candidate pair sees a repeated node id through each slice's own multiplicity
            M = X.shape[0]
            lo, hi = M * rank // world, M * (rank + 1) // world
            if hi > lo:
                eng.diag_adjgrad_batch(X[lo:hi].to(eng.device), y[lo:hi].to(eng.device), gamma, grad_P, out_bar, h1_bar, e_bar,
                                       loss_scale=f, cand=cand)
        if world > 1:
            all_reduce_flat_([grad_P, out_bar, h1_bar, e_bar] + ([cand[2]] if cand is not None else []), process_group)
        rows, cols = eng.export_adj()
        if cand is None:
            return value, torch.stack([rows, cols]), eng.diag_adjgrad_finish(out_bar, h1_bar, e_bar, grad_P)
        grad, gc = eng.diag_adjgrad_finish(out_bar, h1_bar, e_bar, grad_P, cand=cand)
        if sym:
            K = candidates.shape[1]
            gc = 0.5 * (gc[:K] + gc[K:])
        return value, torch.stack([rows, cols]), grad, gc

    def _scale_samples(self, eps):  # laplace/baselaplace.py:1912-1919: samples * posterior_scale
        return eps * (1.0 / self.posterior_precision.sqrt()).reshape(1, self.n_params)

    def functional_variance(self, Js):  # laplace/baselaplace.py:1901-1903
        return torch.einsum("ncp,p,nkp->nck", Js, 1.0 / self.posterior_precision, Js)


class FullLLLaplace(ParametricLaplace):
    """Last-layer full GGN (intent of laplace/lllaplace.py:369-378): parameters are the
    final ``nn.Linear``'s weight (row-major) then bias."""
    _key = ("last_layer", "full")
    _sample_additive = True

    def __init__(self, model, likelihood, *args, **kwargs):
        super().__init__(model, likelihood, *args, **kwargs)
        last = [m for m in model.modules() if isinstance(m, nn.Linear)][-1]
        self.params = [last.weight, last.bias]
        self.n_params = sum(p.numel() for p in self.params)
        self.n_layers = 2
        self._backend_kwargs = dict(self._backend_kwargs, last_layer=True)

    def _init_H(self):
        self.H = torch.zeros(self.n_params, self.n_params, device=self._device)
        self._loss_buf = None
        self._pairs = None

    def _pair_path(self) -> bool:
        """HIP backend: the batches of a fit accumulate in the pair-major layout of the weighted Grams and are placed into
        the P x P matrix once -- no placement / mirror pass per batch (2.3 GB each at the products shape), and the
        all-reduce of a data-parallel fit moves the pair buffers, half the bytes of H."""
        return hasattr(self.backend, "lastlayer_pairs_")

    def _curv_closure(self, X, y, N):
        be = self.backend
        if self._pair_path():
            if self._pairs is None:
                self._pairs = be.engine.new_lastlayer_pair_buffers()
            _, S, Sb, loss_buf = self._pairs
            be.lastlayer_pairs_(S, Sb, loss_buf, X, y)
            return 0.0, None
        return be.full(X, y, N=N)

    def _accumulate(self, H_batch):
        if H_batch is not None:
            self.H += H_batch

    def fit(self, train_loader, override: bool = True, progress_bar: bool = False, process_group=None) -> None:
        super().fit(train_loader, override=override, progress_bar=progress_bar, process_group=process_group)
        if self._loss_buf is not None:  # (all-reduced together with the pair buffers, see _reduce_tensors)
            self.loss = self.loss + self._loss_buf
            self._loss_buf = None

    def _reduce_tensors(self):
        if self._pair_path():
            if self._pairs is None:  # a rank without local samples still takes part in the all-reduce
                self._pairs = self.backend.engine.new_lastlayer_pair_buffers()
            return [self._pairs[0]]  # [S | Sb | loss]: one flat buffer, reduced in place
        return [self.H]

    def _post_accumulate(self):
        if self._pairs is not None:
            _, S, Sb, loss_buf = self._pairs
            self.backend.engine.lastlayer_pairs_place(S, Sb, self.H)
            self._loss_buf = self.backend.factor * loss_buf[0].clone()
            self._pairs = None

    @property
    def posterior_precision(self) -> torch.Tensor:
        return self._H_factor * self.H + torch.diag(self.prior_precision_diag)

    @property
    def log_det_posterior_precision(self) -> torch.Tensor:
        return self.posterior_precision.logdet()

    def functional_variance(self, Js):  # laplace/lllaplace.py via FullLaplace.functional_variance (baselaplace.py:1488-1489)
        cov = torch.linalg.inv(self.posterior_precision.double()).float()  # (fp64 inverse: see FullLaplace.posterior_scale)
        return torch.einsum("ncp,pq,nkq->nck", Js, cov, Js)


class FullLaplace(ParametricLaplace):
    """Full GGN over all weights (laplace/baselaplace.py:1380-1470): dense ``P x P`` precision, Cholesky-based scale,
    log determinant, functional variance and samples.  For the small models it fits (P^2 floats)."""
    _key = ("all", "full")
    _sample_additive = True

    def _init_H(self):
        self.H = torch.zeros(self.n_params, self.n_params, device=self._device)
        self._posterior_scale = None

    def _curv_closure(self, X, y, N):
        return self.backend.full(X, y, N=N)

    def _reduce_tensors(self):
        return [self.H]

    @property
    def posterior_precision(self) -> torch.Tensor:
        return self._H_factor * self.H + torch.diag(self.prior_precision_diag)

    @property
    def posterior_scale(self) -> torch.Tensor:
        """``P^-1/2`` as a lower-triangular factor (laplace/utils/utils.py:118-129).  Factorised in fp64 (P x P of a small
        model): the fp32 Cholesky of a GGN with near-zero directions lost three digits of the functional variance."""
        from torch.distributions.multivariate_normal import _precision_to_scale_tril
        return _precision_to_scale_tril(self.posterior_precision.double()).float()

    @property
    def posterior_covariance(self) -> torch.Tensor:
        scale = self.posterior_scale
        return scale @ scale.T

    @property
    def log_det_posterior_precision(self) -> torch.Tensor:
        return self.posterior_precision.logdet()

    def functional_variance(self, Js):  # :1488-1489
        return torch.einsum("ncp,pq,nkq->nck", Js, self.posterior_covariance, Js)

    def _scale_samples(self, eps):  # :1497-1508: samples @ posterior_scale
        return eps @ self.posterior_scale


def _all_subclasses(cls) -> set:
    return set(cls.__subclasses__()).union([s for c in cls.__subclasses__() for s in _all_subclasses(c)])


def Laplace(model: nn.Module, likelihood: str, subset_of_weights: str = "last_layer",
            hessian_structure: str = "kron", *args, **kwargs) -> BaseLaplace:
    """String-keyed factory with the reference's call signature (laplace/laplace.py:13-47)."""
    if subset_of_weights == "subnetwork" and hessian_structure not in ["full", "diag"]:
        raise ValueError("Subnetwork Laplace requires a full or diagonal Hessian approximation!")
    laplace_map = {sub._key: sub for sub in _all_subclasses(BaseLaplace) if hasattr(sub, "_key")}
    key = (subset_of_weights, hessian_structure)
    if key not in laplace_map:
        raise NotImplementedError(f"{key} is outside the accelerated path; available: {sorted(laplace_map)}")
    return laplace_map[key](model, likelihood, *args, **kwargs)
