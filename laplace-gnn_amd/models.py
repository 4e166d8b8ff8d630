"""Drop-in model modules: ``GCN`` and ``GraphSAGE`` with the reference's constructor signature,
parameter names (``convs.{i}.lin.weight|bias``) and ``forward(x_indices) -> [M, C]`` contract
(gnn/models/base_gnn.py:11-161, gnn/models/models.py:14-62, gnn/models/layers.py:5-46), but a
sparse adjacency held by the HIP engine instead of a dense N x N ``nn.Parameter``.

Inference (eval mode) runs on the GPU through the C ABI; training the weights is outside the
accelerated path.  ``res=True`` (``res.{i}`` Linears, base_gnn.py:97-113) and
``norm="layer"|"batch"`` (``norms.{i}``, base_gnn.py:86-95) are part of the HIP forward / backward (csrc/resnorm.hip).
Unsupported reference options raise immediately: neighbour sampling (``num_sampled_nodes_per_hop``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .engine import GraphEngine


class _Conv(nn.Module):
    """Holds the ``lin`` of GCNConv / GraphSAGEConv (layers.py:5-46) so parameter names match."""

    def __init__(self, in_channels: int, out_channels: int, mult: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(mult * in_channels, out_channels, bias=True)

    def reset_parameters(self):
        self.lin.reset_parameters()


def _to_edge_index(init_adj: torch.Tensor) -> torch.Tensor:
    """Dense 0/1 (or counted) adjacency -> COO.  Values > 1 are clamped like
    gnn/marglik_training.py:405 by the engine's deduplication."""
    if init_adj.dim() == 2 and init_adj.shape[0] == 2 and init_adj.dtype == torch.int64 and init_adj.shape[1] != 2:
        return init_adj
    if init_adj.dim() != 2 or init_adj.shape[0] != init_adj.shape[1]:
        raise ValueError("init_adj must be a dense [N, N] matrix or an int64 edge_index [2, E]")
    if not torch.all((init_adj == 0) | (init_adj == 1)):
        raise AssertionError("adjacency must be binary (gnn/models/base_gnn.py:73)")
    return init_adj.nonzero().t().contiguous()


class BaseGNN(nn.Module):
    kind = "gcn"
    _mult = 1

    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, num_layers: int,
                 X: torch.Tensor, init_adj: torch.Tensor, dropout_p: float = 0.5, act: Optional[str] = "relu",
                 act_kwargs=None, update_adj: bool = False, norm: Optional[str] = None, res: bool = False,
                 symmetric: bool = False, **kwargs):
        super().__init__()
        if norm not in (None, "none", "layer", "batch"):
            raise ValueError(f"Unknown normalization type: {norm}")  # base_gnn.py:94-95
        if update_adj:
            raise NotImplementedError("adjacency learning (STE variants) is out of scope")
        if act not in ("relu", "tanh"):
            raise NotImplementedError(f"activation {act!r} is not supported (relu, tanh)")
        if act_kwargs:
            raise NotImplementedError("act_kwargs are not supported")
        if num_layers < 1:
            raise ValueError("num_layers must be >= 1")
        self.X = X
        self.symmetric = symmetric
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.out_channels, self.num_layers = out_channels, num_layers
        self.act_name = act
        self.dropout = nn.Dropout(p=dropout_p)  # identity in eval mode, kept for interface parity
        self.register_buffer("edge_index", _to_edge_index(init_adj).to(torch.int64), persistent=False)
        self.num_nodes = X.shape[0]
        dims = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        # registration order as in the reference (base_gnn.py:94-98: norms, convs, res) -- it is the order of
        # named_parameters(), hence of the Laplace parameter vector: convs.* first, then res.* (norms.* are filtered out)
        self.norm_kind = None if norm in (None, "none") else norm
        if self.norm_kind == "layer":
            make_norm = lambda: nn.LayerNorm(hidden_channels)  # noqa: E731
        elif self.norm_kind == "batch":
            make_norm = lambda: nn.BatchNorm1d(hidden_channels)  # noqa: E731
        else:
            make_norm = nn.Identity
        self.norms = nn.ModuleList(make_norm() for _ in range(num_layers - 1))
        self.convs = nn.ModuleList(_Conv(dims[i], dims[i + 1], self._mult) for i in range(num_layers))
        # x = res[i](x) + convs[i](adj, x) for every hidden layer (base_gnn.py:100-113, 141-144)
        self.res = nn.ModuleList(nn.Linear(dims[i], hidden_channels) for i in range(num_layers - 1)) if res else nn.ModuleList()
        self._engine: GraphEngine | None = None

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self.X = fn(self.X)
        self._engine = None  # device may have changed; rebuilt lazily
        return self

    @property
    def engine(self) -> GraphEngine:
        """The HIP context for this model (graph ingested once, parameters bound by pointer)."""
        dev = self.convs[0].lin.weight.device
        if self._engine is None:
            if dev.type != "cuda":
                raise RuntimeError("the HIP engine needs the model on a GPU: call model.to('cuda') first "
                                   "(there is no CPU fallback)")
            eng = GraphEngine(self.edge_index.to(dev), self.num_nodes, kind=self.kind, symmetric=self.symmetric)
            eng.bind(self.X.to(dev).contiguous(), [c.lin.weight for c in self.convs],
                     [c.lin.bias for c in self.convs], act=self.act_name, **self._extras())
            self._engine = eng
        return self._engine

    def _extras(self) -> dict:
        """``res`` / ``norm`` state for ``GraphEngine.bind`` (borrowed tensors; BatchNorm runs on its running statistics:
        Laplace.fit calls model.eval(), laplace/baselaplace.py:805)."""
        kw = {}
        if len(self.res):
            kw["res_weights"], kw["res_biases"] = [m.weight for m in self.res], [m.bias for m in self.res]
        if self.norm_kind is not None and len(self.norms):
            kw.update(norm=self.norm_kind, norm_eps=float(self.norms[0].eps), norm_weight=[m.weight for m in self.norms],
                      norm_bias=[m.bias for m in self.norms])
            if self.norm_kind == "batch":
                kw.update(norm_mean=[m.running_mean for m in self.norms], norm_var=[m.running_var for m in self.norms])
        return kw

    def full_adj(self) -> torch.Tensor:
        """Dense 0/1 adjacency as the reference's ``model.adj`` holds it (small graphs only)."""
        rows, cols = self.engine.export_adj()
        adj = torch.zeros(self.num_nodes, self.num_nodes, device=rows.device)
        adj[rows, cols] = 1
        return adj

    @property
    def n_outputs(self) -> int:
        """Width of the logits: Laplace.fit reads it instead of running the one-sample forward pass the reference uses to
        find it (laplace/baselaplace.py:806-816)."""
        return self.out_channels

    def forward_adj(self) -> torch.Tensor:
        """Propagation matrix as a sparse COO tensor (normalize_adj / mean_agg result)."""
        r, c, v = self.engine.export_propagation()
        return torch.sparse_coo_tensor(torch.stack([r, c]), v, (self.num_nodes, self.num_nodes))

    def forward(self, x_indices: torch.Tensor) -> torch.Tensor:
        if self.training and (self.dropout.p > 0 or self.norm_kind == "batch"):
            raise NotImplementedError("only eval-mode forward runs on the HIP path (Laplace.fit calls model.eval())")
        return self.engine.forward(x_indices.to(self.convs[0].lin.weight.device))


class GCN(BaseGNN):
    """gnn/models/models.py:14-34 (self loops added, D^-1/2 A^T D^-1/2 propagation)."""
    kind = "gcn"
    _mult = 1


class GraphSAGE(BaseGNN):
    """gnn/models/models.py:37-62 with ``num_sampled_nodes_per_hop=None`` (no sampling: the reference's
    sampler is unseeded, gnn/models/utils.py:115-131)."""
    kind = "sage"
    _mult = 2

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, X, init_adj,
                 num_sampled_nodes_per_hop=None, **kwargs):
        if num_sampled_nodes_per_hop is not None:
            raise NotImplementedError("neighbour sampling is not supported (unseeded in the reference)")
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, X, init_adj, **kwargs)
        self.num_sampled_nodes_per_hop = None
