"""``laplace.curvature``-compatible backend that runs on the HIP engine.

Drop-in for ``laplace.curvature.CurvlinopsGGN`` on GCN / GraphSAGE models: pass the class as
``backend=`` to ``Laplace(...)``.  Same constructor signature, attributes and return
conventions as laplace/curvature/curvature.py:12-83 (CurvatureInterface) and
laplace/curvature/curvlinops.py:20-167 (CurvlinopsInterface / CurvlinopsGGN):

* ``kron(x, y, N) -> (loss, Kron)`` with ``Kron.kfacs = [[B0, A0], [B0], [B1, A1], [B1], ...]``,
  A already rescaled by M/N, everything multiplied by ``factor`` the way ``Kron.__mul__`` does
  (curvlinops.py:55-75, 46-53, 104-108); fresh tensors on every call (callers mutate them);
* ``diag(x, y, **kw) -> (loss, H[P])``  (curvature.py:412-432);
* ``full(x, y, **kw) -> (loss, H[P, P])`` -- implemented for ``last_layer=True`` (curvature.py:374-410
  with last_layer_jacobians :132-167);
* attributes ``lossfunc``, ``factor``, ``params``, ``params_dict``, ``buffers_dict``, ``model``,
  ``_model``, ``likelihood``, ``last_layer``, ``stochastic``.

In-place fast paths (``kron_accumulate_`` ...) let the fit loop add straight into one flat buffer.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn
from torch.nn import CrossEntropyLoss, MSELoss

from .matrix import Kron

_BRIDGED = {}


def _kron_class():
    """The container ``kron()`` returns.  Inside the reference's own front, ``ParametricLaplace.fit`` does
    ``self.H += H_batch`` (laplace/baselaplace.py:852) and ``Kron.__add__`` insists on ITS class
    (laplace/utils/matrix.py:85-86).  When that module is loaded in this process the returned object therefore is an
    instance of both: this package's ``Kron`` (so this package's front keeps working) and the host application's --
    ``backend=HipGGN`` is a drop-in without an adapter.  The host's ``__add__`` returns plain host ``Kron`` objects,
    so everything downstream (``decompose``, ``logdet`` ...) is the host's own code."""
    import sys

    mod = sys.modules.get("laplace.utils.matrix")
    host = getattr(mod, "Kron", None)
    if host is None or host is Kron:
        return Kron
    if host not in _BRIDGED:
        def _add(self, other):
            # Python asks the subclass's reflected method first (``host_kron + bridged`` lands here with the host object
            # as ``other``): sums that involve a host container are the host's (its own class comes back), sums with
            # this package's plain container stay here.  Elementwise addition commutes, operand order is irrelevant.
            if isinstance(other, host):
                return host.__add__(other, self)
            return Kron.__add__(self, other)

        _BRIDGED[host] = type("Kron", (Kron, host), {
            "__doc__": "laplace_gnn_amd.Kron that is also a laplace.utils.matrix.Kron", "__add__": _add, "__radd__": _add})
    return _BRIDGED[host]


class HipCurvatureInterface:
    def __init__(self, model: nn.Module, likelihood: str, last_layer: bool = False,
                 subnetwork_indices: torch.LongTensor | None = None, dict_key_x: str = "input_ids",
                 dict_key_y: str = "labels", stochastic: bool = False, fork_exact_seed: bool = True,
                 num_samples: int = 1, generator: torch.Generator | None = None):
        assert likelihood in ["regression", "classification"]
        if subnetwork_indices is not None:
            raise NotImplementedError("subnetwork Laplace is out of scope for the HIP backend")
        self.likelihood = likelihood
        self.model = model
        self.last_layer = last_layer
        self.subnetwork_indices = None
        self.dict_key_x, self.dict_key_y = dict_key_x, dict_key_y
        # stochastic=True: Monte-Carlo Fisher instead of the GGN (CurvlinopsGGN, laplace/curvature/curvlinops.py:143-167;
        # GGNInterface, laplace/curvature/curvature.py:327-364 with ``num_samples`` draws for diag / full)
        self.stochastic = bool(stochastic)
        self.num_samples = int(num_samples)
        self.generator = generator
        self.fork_exact_seed = fork_exact_seed
        if likelihood == "regression":
            self.lossfunc, self.factor = MSELoss(reduction="sum"), 0.5
        else:
            self.lossfunc, self.factor = CrossEntropyLoss(reduction="sum"), 1.0
        # the fork's parameter filter (laplace/curvature/curvature.py:74-79)
        self.params, self.params_dict = [], {}
        for k, v in self.model.named_parameters():
            if v.requires_grad and "adj" not in k and "norms" not in k:
                self.params.append(v)
                self.params_dict[k] = v
        self.buffers_dict = {k: v for k, v in self.model.named_buffers()}
        if not hasattr(model, "engine"):
            raise TypeError("HipGGN needs a model that exposes a HIP `engine` (laplace_gnn_amd.models.GCN / "
                            "GraphSAGE); there is no generic autograd fallback")
        if likelihood == "regression" and last_layer:
            raise NotImplementedError("last-layer full GGN: classification likelihood only")
        if stochastic and last_layer:
            # the reference applies the MC functional Fisher to last_layer_jacobians here (laplace/curvature/curvature.py:
            # 343-364, 374-432); the last-layer kernels build the exact GGN only -- refuse rather than ignore the flag
            raise NotImplementedError("stochastic=True together with last_layer=True is not implemented")

    @property
    def _model(self) -> nn.Module:
        return self.model.last_layer if self.last_layer else self.model

    @property
    def engine(self):
        return self.model.engine

    # ---- kron ----------------------------------------------------------------------------------
    @property
    def num_classes(self) -> int:
        return self.engine.dims[-1]

    def kron_accumulate_(self, views, loss_buf, x: torch.Tensor, y: torch.Tensor, N: int, fuse: bool = True,
                         classes: tuple[int, int] | None = None):
        """Add this batch's RAW factors (A_l/N_train, B_l) and loss into caller-owned buffers; ``classes``
        restricts the call to a range of class columns (exact additive share, see the C ABI)."""
        self.engine.set_likelihood(self.likelihood)  # regression: sqrt(2) I seeds and the MSE loss on the device
        if classes is not None and getattr(self.engine, "supports_shares", False):
            # the multi-GPU unit (begin, end) of C parts per batch: the engine cuts a batch the way its route splits best
            # (lgnn_kfac_accumulate_share: destination-node ranges on the path routes -- a class range would repeat the
            # path products on every rank -- class ranges otherwise); the parts still add up to the batch exactly
            self.engine.kfac_accumulate(x, y, N, views, loss_buf, fork_exact=self.fork_exact_seed, fuse=fuse,
                                        share=(int(classes[0]), int(classes[1]), self.num_classes))
            return
        self.engine.kfac_accumulate(x, y, N, views, loss_buf, fork_exact=self.fork_exact_seed, fuse=fuse,
                                    classes=classes)

    def pack_kron(self, views) -> Kron:
        """[[B0, A0], [B0], ...] in parameter order, times ``factor`` distributed like Kron.__mul__
        (laplace/utils/matrix.py:95-113): sqrt(factor) per factor of a 2-block, factor on a 1-block."""
        kfacs = []
        f2 = self.factor ** 0.5
        for A, B in views:
            kfacs.append([B * f2 if self.factor != 1.0 else B.clone(), A * f2 if self.factor != 1.0 else A.clone()])
            kfacs.append([B * self.factor if self.factor != 1.0 else B.clone()])
        # factor == 1: a bias block's B is its weight block's B (clones of the same view)
        return _kron_class()(kfacs, tied=range(1, len(kfacs), 2) if self.factor == 1.0 else None)

    # Fisher type of the KFAC factors: "type2" (GGN), "mc" (stochastic=True), "empirical" (HipEF)
    @property
    def _kron_fisher_type(self) -> str:
        return "mc" if self.stochastic else "type2"

    def draw_labels(self, x: torch.Tensor, kfac: bool = True) -> torch.Tensor:
        """One draw from the model's predictive distribution at the batch nodes: categorical for classification; for
        regression N(f, 1/2) in the KFAC route (MSELoss implies variance 1/2, curvlinops/kfac.py:721-729) and N(f, 1) in
        the Jacobian route (laplace/curvature/curvature.py:349-351)."""
        f = self.engine.forward(x)
        if self.likelihood == "regression":
            std = 0.5 ** 0.5 if kfac else 1.0
            return f + std * torch.randn(f.shape, device=f.device, dtype=f.dtype, generator=self.generator)
        return torch.multinomial(torch.softmax(f, dim=-1), 1, generator=self.generator).squeeze(-1)

    def _fisher_kron(self, x, y, N, draws, fuse: bool = True):
        """KFAC with FisherType.EMPIRICAL (``draws = [y]``) / FisherType.MC (``draws`` = sampled labels): one backward
        pass per draw, gradient covariance scaled by 1 / len(draws) (curvlinops/kfac.py:663-674, 806-810)."""
        self.engine.set_likelihood(self.likelihood)
        rs = 2.0 if self.likelihood == "regression" else 1.0  # d MSE_sum / d f = 2 (f - y)
        _, views, loss = self.engine.new_kfac_buffers()
        for s_, ys in enumerate(draws):
            self.engine.kfac_accumulate_fisher(x, ys, y if s_ == 0 else None, N, views, loss, resid_scale=rs,
                                               b_scale=1.0 / len(draws), fuse=fuse)
        return self.factor * loss[0].clone(), self.pack_kron(views)

    def kron(self, x: torch.Tensor, y: torch.Tensor, N: int, mc_samples: int = 1, mc_labels=None, **kwargs: Any):
        """``mc_samples`` / ``mc_labels`` only matter for ``stochastic=True`` (MC Fisher): the labels are drawn on the
        device unless the caller brings them (reproducibility across devices)."""
        if kwargs:
            raise NotImplementedError(f"unsupported kron kwargs {sorted(kwargs)} (kfac_approx)")
        if self._kron_fisher_type == "mc":
            draws = list(mc_labels) if mc_labels is not None else [self.draw_labels(x) for _ in range(int(mc_samples))]
            return self._fisher_kron(x, y, N, draws)
        if self._kron_fisher_type == "empirical":
            return self._fisher_kron(x, y, N, [y])
        _, views, loss = self.engine.new_kfac_buffers()
        self.kron_accumulate_(views, loss, x, y, N)
        return self.factor * loss[0].clone(), self.pack_kron(views)

    def _mc_functional_fisher(self, x, y, full: bool, mc_labels=None):
        """GGNInterface with stochastic=True (laplace/curvature/curvature.py:343-364, 401-432): the functional Fisher
        ``1/S sum_s r_s r_s^T`` (r = softmax(f) - onehot(y_s), regression: f - y_s) replaces Lambda; no factor on H."""
        eng = self.engine
        eng.set_likelihood(self.likelihood)
        P = eng.n_params
        H = torch.zeros((P, P) if full else (P,), dtype=torch.float32, device=eng.device)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        draws = list(mc_labels) if mc_labels is not None else [self.draw_labels(x, kfac=False)
                                                               for _ in range(self.num_samples)]
        for s_, ys in enumerate(draws):
            eng.ef_accumulate(x, ys, y if s_ == 0 else None, 1.0, 1.0 / len(draws), diag=None if full else H,
                              full=H if full else None, loss=loss if s_ == 0 else None)
        return self.factor * loss[0], H

    # ---- diag ----------------------------------------------------------------------------------
    def diag(self, x: torch.Tensor, y: torch.Tensor, mc_labels=None, **kwargs: Any):
        if self.stochastic:
            return self._mc_functional_fisher(x, y, full=False, mc_labels=mc_labels)
        self.engine.set_likelihood(self.likelihood)  # regression: H = sum J^T J on the device (no factor on H)
        eng = self.engine
        H = torch.zeros(eng.n_params, dtype=torch.float32, device=eng.device)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        eng.diag_accumulate(x, y, H, loss)
        return self.factor * loss[0], H

    def diag_accumulate_(self, H: torch.Tensor, loss_buf: torch.Tensor, x: torch.Tensor, y: torch.Tensor):
        """Diagonal GGN of one batch added IN PLACE to the caller's ``H`` and the raw loss sum to ``loss_buf`` (the caller
        applies ``factor``): no temporaries, no torch kernels per batch (laplace/curvature/curvature.py:412-432)."""
        if self.stochastic:
            raise NotImplementedError("in-place accumulation exists for the GGN diagonal")
        self.engine.set_likelihood(self.likelihood)
        self.engine.diag_accumulate(x, y, H, loss_buf)

    # ---- full (last layer) ---------------------------------------------------------------------
    def full(self, x: torch.Tensor, y: torch.Tensor, mc_labels=None, **kwargs: Any):
        if self.stochastic and not self.last_layer:
            return self._mc_functional_fisher(x, y, full=True, mc_labels=mc_labels)
        if not self.last_layer:
            return self._full_all_weights(x, y)
        eng = self.engine
        p_ll = eng.in_dims[-1] * eng.dims[-1] + eng.dims[-1]
        H = torch.zeros(p_ll, p_ll, dtype=torch.float32, device=eng.device)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        eng.lastlayer_full_accumulate(x, y, H, loss)
        return self.factor * loss[0], H

    def lastlayer_pairs_(self, S: torch.Tensor, Sb: torch.Tensor, loss_buf: torch.Tensor, x: torch.Tensor, y: torch.Tensor):
        """Last-layer full GGN of one batch added to the caller's pair-major accumulators (``GraphEngine.
        new_lastlayer_pair_buffers``); ``GraphEngine.lastlayer_pairs_place`` turns them into the P x P matrix once per fit."""
        if not self.last_layer or self.likelihood != "classification" or self.factor != 1.0:
            raise NotImplementedError("pair-major accumulation exists for the last-layer classification GGN")
        self.engine.set_likelihood("classification")
        self.engine.lastlayer_pairs_accumulate(x, y, S, Sb, loss_buf)

    def full_accumulate_(self, H: torch.Tensor, loss_buf: torch.Tensor, x: torch.Tensor, y: torch.Tensor):
        """Last-layer full GGN of one batch added IN PLACE to the caller's ``H`` (and the raw loss to ``loss_buf``): no
        ``P x P`` temporary per batch -- 2.3 GB each at the products shape."""
        if not self.last_layer or self.likelihood != "classification" or self.factor != 1.0:
            raise NotImplementedError("in-place accumulation exists for the last-layer classification GGN")
        self.engine.set_likelihood("classification")
        self.engine.lastlayer_full_accumulate(x, y, H, loss_buf)

    def _full_all_weights(self, x: torch.Tensor, y: torch.Tensor):
        """Full GGN over all weights, ``H = sum_n J_n^T Lambda_n J_n`` (GGNInterface.full, laplace/curvature/curvature.py:
        374-410; the reference's default backend routes this case through un-vendored backpack, the same matrix): device
        Jacobians in chunks, mixed with the Hessian square root and contracted by the MFMA Gram kernel
        (``lgnn_full_accumulate``).  Regression: ``sum J^T J`` (H_lik = None, :406-407), no factor on H."""
        eng = self.engine
        eng.set_likelihood(self.likelihood)
        P = eng.n_params
        H = torch.zeros(P, P, dtype=torch.float32, device=eng.device)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        eng.full_accumulate(x, y, H, loss)
        if self.likelihood == "regression":
            return self.factor * loss[0], H
        return self.factor * loss[0], self.factor * H

    def check_async_errors(self):
        self.engine.check_async_errors()

    def peek_async_errors(self):
        self.engine.peek_async_errors()

    def jacobians(self, x: torch.Tensor, enable_backprop: bool = False):
        """(Js [M, C, P], f [M, C]) as CurvatureInterface.jacobians (laplace/curvature/curvature.py:89-130); the
        M * C backward passes run as planes through the HIP engine (csrc/jacobian.hip).  No autograd graph."""
        if enable_backprop:
            raise NotImplementedError("the HIP Jacobians carry no autograd graph (enable_backprop=False only)")
        if self.last_layer:
            return self.last_layer_jacobians(x)
        return self.engine.jacobians(x)

    functorch_jacobians = jacobians

    def last_layer_jacobians(self, x: torch.Tensor, enable_backprop: bool = False):
        """(Js [M, C, C * D + C], f [M, C]) of the final nn.Linear alone (CurvatureInterface.last_layer_jacobians,
        laplace/curvature/curvature.py:132-167): ``J_n = [I_C (x) phi_n^T | s_n I_C]`` -- weight index c * D + d, then the C
        bias entries; phi_n / s_n from the device (``lgnn_lastlayer_features``)."""
        if enable_backprop:
            raise NotImplementedError("the HIP Jacobians carry no autograd graph (enable_backprop=False only)")
        phi, s, f = self.engine.lastlayer_features(x)
        M, D = phi.shape
        C = f.shape[1]
        eye = torch.eye(C, device=phi.device, dtype=phi.dtype)
        Jw = torch.einsum("kp,ij->kijp", phi, eye).reshape(M, C, C * D)
        Jb = s.reshape(M, 1, 1) * eye.unsqueeze(0)
        return torch.cat([Jw, Jb], dim=2), f

    def gradients(self, x: torch.Tensor, y: torch.Tensor):
        """(Gs [M, P], loss): per-sample gradients of the summed loss, ``G_n = J_n^T d loss_n / d f_n``
        (CurvatureInterface.gradients, laplace/curvature/curvature.py:169-210); loss without the interface factor."""
        eng = self.engine
        eng.set_likelihood(self.likelihood)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        rs = 2.0 if self.likelihood == "regression" else 1.0
        Gs = eng.ef_accumulate(x, y, y, rs, 1.0, grads=True, loss=loss)
        return Gs, loss[0]


class HipGGN(HipCurvatureInterface):
    """GGN backend on MI355X; the counterpart of ``CurvlinopsGGN`` (curvlinops.py:143-167); ``stochastic=True`` switches
    to the Monte-Carlo Fisher like the reference's flag."""


class HipEF(HipCurvatureInterface):
    """Empirical Fisher backend; the counterpart of ``CurvlinopsEF`` / ``EFInterface`` (laplace/curvature/curvlinops.py:
    170-179, laplace/curvature/curvature.py:435-504): KFAC with FisherType.EMPIRICAL, ``diag = factor * sum_n G_n^2``,
    ``full = factor * G^T G`` from the per-sample loss gradients."""

    def __init__(self, model, likelihood, last_layer: bool = False, subnetwork_indices=None, dict_key_x: str = "input_ids",
                 dict_key_y: str = "labels", **kwargs):
        if last_layer:
            raise NotImplementedError("empirical Fisher: all-weights Laplace only")
        super().__init__(model, likelihood, last_layer, subnetwork_indices, dict_key_x, dict_key_y, stochastic=False, **kwargs)

    @property
    def _kron_fisher_type(self) -> str:
        return "empirical"

    def _ef(self, x, y, full: bool):
        eng = self.engine
        eng.set_likelihood(self.likelihood)
        P = eng.n_params
        H = torch.zeros((P, P) if full else (P,), dtype=torch.float32, device=eng.device)
        loss = torch.zeros(1, dtype=torch.float32, device=eng.device)
        rs = 2.0 if self.likelihood == "regression" else 1.0
        eng.ef_accumulate(x, y, y, rs, self.factor, diag=None if full else H, full=H if full else None, loss=loss)
        return self.factor * loss[0], H

    def diag(self, x: torch.Tensor, y: torch.Tensor, **kwargs: Any):
        return self._ef(x, y, full=False)

    def full(self, x: torch.Tensor, y: torch.Tensor, **kwargs: Any):
        return self._ef(x, y, full=True)
