"""TEST-ONLY backend: the CPU oracle behind the ``CurvatureInterface`` signatures, so that the host
logic of ``laplace_gnn_amd.laplace`` (fit loop, Kron accumulation, override semantics, batch
sharding + all-reduce) can be exercised without a GPU.  Never shipped, never the product path."""
import numpy as np
import torch
from torch.nn import CrossEntropyLoss

import gnn_laplace_oracle as O
from laplace_gnn_amd.matrix import Kron


def oracle_model_of(m):
    """The oracle's plain-array mirror of a laplace_gnn_amd model on the CPU, res / norm state included."""
    rp, col = O.edge_index_to_adj_csr(m.edge_index.numpy(), m.num_nodes, m.kind, m.symmetric)
    Ws = [c.lin.weight.detach().numpy() for c in m.convs]
    bs = [c.lin.bias.detach().numpy() for c in m.convs]
    kw = {}
    if len(getattr(m, "res", [])):
        kw.update(res_weights=[r.weight.detach().numpy() for r in m.res], res_biases=[r.bias.detach().numpy() for r in m.res])
    if getattr(m, "norm_kind", None) is not None:
        kw.update(norm=m.norm_kind, norm_eps=float(m.norms[0].eps), norm_weight=[n.weight.detach().numpy() for n in m.norms],
                  norm_bias=[n.bias.detach().numpy() for n in m.norms])
        if m.norm_kind == "batch":
            kw.update(norm_mean=[n.running_mean.numpy() for n in m.norms], norm_var=[n.running_var.numpy() for n in m.norms])
    return O.GnnModel(m.kind, rp, col, m.X.numpy(), Ws, bs, **kw)


class CpuForwardGCN(torch.nn.Module):
    """Wraps a laplace_gnn_amd model so that forward() also runs through the oracle on the CPU."""

    def __init__(self, model):
        super().__init__()
        self.inner = model
        self.convs = model.convs
        self._oracle_engine = None

    @property
    def engine(self):
        """The CPU stand-in for the HIP engine (tests/oracle_engine.py): lets the real ``HipGGN`` class run here."""
        if self._oracle_engine is None:
            from oracle_engine import OracleEngine
            self._oracle_engine = OracleEngine(self.inner)
        return self._oracle_engine

    def oracle_model(self):
        return oracle_model_of(self.inner)

    def forward(self, x_indices):
        out, _, _ = O.forward_all(self.oracle_model())
        return torch.from_numpy(out[x_indices.numpy()])


class OracleBackend:
    def __init__(self, model, likelihood, last_layer=False, subnetwork_indices=None, dict_key_x="input_ids",
                 dict_key_y="labels", stochastic=False):
        self.model, self.likelihood, self.last_layer = model, likelihood, last_layer
        self.lossfunc, self.factor = CrossEntropyLoss(reduction="sum"), 1.0
        if likelihood == "regression":
            from torch.nn import MSELoss
            self.lossfunc, self.factor = MSELoss(reduction="sum"), 0.5
        self.params = [p for k, p in model.named_parameters() if p.requires_grad and "adj" not in k and "norms" not in k]
        self.params_dict = dict(model.named_parameters())
        self.buffers_dict = dict(model.named_buffers())
        self.calls = []

    def kron(self, x, y, N, **kw):
        self.calls.append(("kron", tuple(x.tolist())))
        loss, kfacs = O.kfac_batch(self.model.oracle_model(), x.numpy(), y.numpy(), N, likelihood=self.likelihood)
        return torch.tensor(float(loss)), Kron([[torch.from_numpy(np.ascontiguousarray(h)) for h in F] for F in kfacs])

    def jacobians(self, x, enable_backprop=False):
        self.calls.append(("jacobians", tuple(x.tolist())))
        Js, f = O.jacobians_batch(self.model.oracle_model(), x.numpy())
        return torch.from_numpy(Js), torch.from_numpy(f)

    def diag(self, x, y, **kw):
        self.calls.append(("diag", tuple(x.tolist())))
        loss, H = O.diag_batch(self.model.oracle_model(), x.numpy(), y.numpy(), self.likelihood)
        return torch.tensor(float(loss)), torch.from_numpy(H)

    def full(self, x, y, **kw):
        self.calls.append(("full", tuple(x.tolist())))
        fn = O.lastlayer_full_batch if self.last_layer else O.full_batch
        loss, H = fn(self.model.oracle_model(), x.numpy(), y.numpy())
        return torch.tensor(float(loss)), torch.from_numpy(H)


class OracleClassBackend(OracleBackend):
    """Adds the in-place, class-range capable fast path of HipGGN (kron_accumulate_ / pack_kron /
    new_kfac_buffers / num_classes) on top of the oracle, to exercise the (batch, class) unit scheduling."""

    @property
    def num_classes(self):
        return self.model.convs[-1].lin.weight.shape[0]

    def new_kfac_buffers(self):
        views = []
        for lin in [c.lin for c in self.model.convs] + list(getattr(self.model.inner, "res", [])):
            o, i = lin.weight.shape
            views.append((torch.zeros(i, i), torch.zeros(o, o)))
        return None, views, torch.zeros(1)

    def kron_accumulate_(self, views, loss_buf, x, y, N, classes=None):
        self.calls.append(("kron_", tuple(x.tolist()), classes))
        loss, kfacs = O.kfac_batch(self.model.oracle_model(), x.numpy(), y.numpy(), N, classes=classes)
        for l, (A, B) in enumerate(views):
            B += torch.from_numpy(kfacs[2 * l][0])
            A += torch.from_numpy(kfacs[2 * l][1])
        loss_buf += float(loss)

    def pack_kron(self, views):
        kfacs = []
        for A, B in views:
            kfacs.append([B.clone(), A.clone()])
            kfacs.append([B.clone()])
        return Kron(kfacs)
