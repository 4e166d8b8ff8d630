"""Drop-in model modules: ``GCN`` and ``GraphSAGE`` with the reference's constructor signature,
parameter names (``convs.{i}.lin.weight|bias``) and ``forward(x_indices) -> [M, C]`` contract
(gnn/models/base_gnn.py:11-161, gnn/models/models.py:14-62, gnn/models/layers.py:5-46), but a
sparse adjacency held by the HIP engine instead of a dense N x N ``nn.Parameter``.

Inference (eval mode) runs on the GPU through the C ABI; training the weights is outside the
accelerated path.  ``res=True`` (``res.{i}`` Linears, base_gnn.py:97-113) and
``norm="layer"|"batch"`` (``norms.{i}``, base_gnn.py:86-95) are part of the HIP forward / backward (csrc/resnorm.hip).
Neighbour sampling (``num_sampled_nodes_per_hop``) needs an explicit ``sample_seed`` (the reference's sampler is unseeded).
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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints of the reference carry the graph as the dense ``adj`` parameter (gnn/models/base_gnn.py:75-76; the driver
        insists on the key, gnn/marglik_training.py:78-79; a GCN's holds its self loops, gnn/models/models.py:23).  Here the
        graph is the ``edge_index`` buffer of the HIP engine: an ``adj`` entry -- dense [N, N] or a sparse COO / CSR tensor,
        0/1 or the continuous values of an STE model, which propagate with ``adj > threshold`` (models.py:99-101) -- is turned
        into it (and dropped from the dict so that ``strict`` loading does not trip over it); the engine is rebuilt lazily."""
        key = prefix + "adj"
        if key in state_dict and not isinstance(getattr(self, "adj", None), nn.Parameter):
            adj = state_dict.pop(key)
            thr = float(getattr(self, "threshold", 0.0))
            if adj.layout != torch.strided:
                adj = adj.to_sparse_coo().coalesce()
                ei = adj.indices()[:, adj.values() > thr]
            else:
                if adj.dim() != 2 or adj.shape[0] != adj.shape[1] or adj.shape[0] != self.num_nodes:
                    error_msgs.append(f"adj has shape {tuple(adj.shape)}, expected ({self.num_nodes}, {self.num_nodes})")
                    ei = None
                else:
                    ei = (adj > thr).nonzero().t()
            if ei is not None:
                self.edge_index = ei.to(torch.int64).contiguous().to(self.edge_index.device)
                self._engine = None
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

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
    """gnn/models/models.py:37-62.  ``num_sampled_nodes_per_hop=k``: the reference multiplies the adjacency by a fresh random
    subgraph -- at most k of every row's neighbours, ``torch.randperm`` from the GLOBAL, unseeded generator -- on EVERY call of
    ``forward_adj`` (gnn/models/utils.py:115-131; its ``seed`` argument is commented out), so the three forward passes and the
    backward passes of one KFAC batch each see a different graph and no two runs agree: there is nothing to pin.  Here the draw
    is explicit: ``sample_seed`` (required with k) selects ONE subgraph -- every row keeps ``min(deg, k)`` of its neighbours,
    uniformly, from a generator seeded with it -- that all passes of a fit share; ``resample(seed)`` draws another.  The sampled
    graph is directed (a row's choice is its own), whatever ``symmetric`` says about the full one.  Parity unpinned by
    construction (tests hold the distributional properties and the equality with a model built on the sampled edge list)."""
    kind = "sage"
    _mult = 2

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, X, init_adj,
                 num_sampled_nodes_per_hop=None, sample_seed=None, **kwargs):
        if num_sampled_nodes_per_hop is not None:
            if sample_seed is None:
                raise NotImplementedError("neighbour sampling: the reference's sampler is unseeded (a new graph per forward "
                                          "call); pass sample_seed=<int> for one seeded draw per fit")
            if int(num_sampled_nodes_per_hop) < 1:
                raise ValueError("num_sampled_nodes_per_hop must be >= 1")
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, X, init_adj, **kwargs)
        self.num_sampled_nodes_per_hop = None if num_sampled_nodes_per_hop is None else int(num_sampled_nodes_per_hop)
        self.sample_seed = sample_seed

    def resample(self, seed: int):
        """Draw another subgraph (the engine is rebuilt on next use; fitted posteriors belong to the previous graph)."""
        self.sample_seed = int(seed)
        self._engine = None

    def sampled_edge_index(self, device) -> torch.Tensor:
        """The seeded draw: [2, E'] int64, row-major sorted, at most k entries per row of the full 0/1 adjacency."""
        full = GraphEngine(self.edge_index.to(device), self.num_nodes, kind=self.kind, symmetric=self.symmetric)
        rows, cols = full.export_adj()
        full.close()
        gen = torch.Generator(device="cpu").manual_seed(int(self.sample_seed))
        key = torch.rand(rows.numel(), generator=gen).to(device)
        order = torch.argsort(key)                       # random order ...
        order = order[torch.argsort(rows[order], stable=True)]  # ... within every row
        r_sorted = rows[order]
        start = torch.searchsorted(r_sorted, r_sorted)  # first position of the entry's row
        keep = (torch.arange(rows.numel(), device=device) - start) < self.num_sampled_nodes_per_hop
        sel = torch.sort(order[keep]).values             # back to row-major order
        return torch.stack([rows[sel], cols[sel]]).contiguous()

    @property
    def engine(self) -> GraphEngine:
        if self.num_sampled_nodes_per_hop is None:
            return BaseGNN.engine.fget(self)
        dev = self.convs[0].lin.weight.device
        if self._engine is None:
            if dev.type != "cuda":
                raise RuntimeError("the HIP engine needs the model on a GPU: call model.to('cuda') first "
                                   "(there is no CPU fallback)")
            eng = GraphEngine(self.sampled_edge_index(dev), self.num_nodes, kind=self.kind, symmetric=False)
            eng.bind(self.X.to(dev).contiguous(), [c.lin.weight for c in self.convs],
                     [c.lin.bias for c in self.convs], act=self.act_name, **self._extras())
            self._engine = eng
        return self._engine


class STEGCN(GCN):
    """The reference's structure-learning model (gnn/models/models.py:65-118) on a sparse pattern.

    The reference keeps a dense continuous ``adj`` parameter [N, N] (initialised with the 0/1 adjacency and the GCN's self
    loops), propagates with ``normalize_adj(fill_diagonal_(BinarizeSTE(adj) , 1))`` -- symmetric models binarise
    ``(adj + adj^T) / 2`` -- and lets ``neg_marglik.backward()`` / ``adj_optimizer.step()`` move every entry
    (gnn/marglik_training.py:197-224).  Here ``adj`` holds the continuous values of the TRACKED pairs only: the stored
    off-diagonal entries (1) and the caller's ``candidates`` (0), ``adj_index`` [2, n] names them (row-major sorted); every
    other pair stays an implicit non-edge with value 0.  The diagonal is not tracked: the forward overwrites it and its
    gradient is zero (models.py:114).  The binarised graph lives in the HIP engine; after an optimizer step ``apply_adj()``
    re-thresholds the values and hands the entries that changed side to ``lgnn_update_adjacency`` (no re-ingest).

        la.fit(loader)
        model.adj_backward(la, loader)        # adj.grad <- d(-marglik)/d adj   (neg_marglik.backward())
        adj_optimizer.step(); model.apply_adj()
        la.fit(loader)                        # the loop of gnn/marglik_training.py:211-224

    With every non-edge listed as a candidate this reproduces the reference's loop (tests/golden/steloop_*.npz); with a
    subset, ``clip_grad_norm_`` and the optimizer only see the tracked entries.  ``sign_grad=True`` is refused: the
    reference takes the sign of the gradient of every forward CALL (the curvature graph and the loss forward of each batch)
    and autograd sums the signs, which no accumulated gradient reproduces."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, X, init_adj, dropout_p: float = 0.5,
                 act="relu", act_kwargs=None, threshold: float = 0.5, train_masked_update: bool = False, train_nodes=None,
                 symmetric: bool = False, sign_grad: bool = False, candidates: Optional[torch.Tensor] = None, **kwargs):
        if sign_grad:
            raise NotImplementedError("sign_grad=True: the reference sums the signs of per-forward-call gradients "
                                      "(gnn/models/utils.py:75-76); not reproducible from one accumulated gradient")
        if train_masked_update and train_nodes is None:
            raise ValueError("'train_nodes' must be provided, to use train_masked_update.")  # models.py:91-93
        kwargs.pop("update_adj", None)
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, X, init_adj, dropout_p=dropout_p, act=act,
                         act_kwargs=act_kwargs, symmetric=symmetric, **kwargs)
        self.threshold = float(threshold)
        self.sign_grad = False
        self.train_masked_update = bool(train_masked_update)
        N = self.num_nodes
        ei = self.edge_index.cpu()
        keys = ei[0] * N + ei[1]
        if symmetric:
            keys = torch.cat([keys, ei[1] * N + ei[0]])
        keys = torch.unique(keys[(keys // N) != (keys % N)])  # the stored off-diagonal pattern (duplicates clamp to one entry)
        vals = torch.ones(keys.numel())
        if candidates is not None and candidates.numel():
            c = candidates.cpu().to(torch.int64)
            ck = c[0] * N + c[1]
            if symmetric:
                ck = torch.cat([ck, c[1] * N + c[0]])
            ck = torch.unique(ck[(ck // N) != (ck % N)])
            ck = ck[~torch.isin(ck, keys)]
            keys, order = torch.sort(torch.cat([keys, ck]))
            vals = torch.cat([vals, torch.zeros(ck.numel())])[order]
        self.register_buffer("adj_index", torch.stack([keys // N, keys % N]))
        self.adj = nn.Parameter(vals)  # (the Laplace parameter filter drops names containing 'adj', baselaplace.py:118-122)
        if self.train_masked_update:
            # soft mask of BinarizeSTE (models.py:94-98): 0.1 between two training nodes, 1 elsewhere
            is_train = torch.zeros(N, dtype=torch.bool)
            is_train[torch.as_tensor(train_nodes).cpu().to(torch.int64)] = True
            both = is_train[self.adj_index[0]] & is_train[self.adj_index[1]]
            self.register_buffer("grad_adj_mask", torch.where(both, torch.tensor(0.1), torch.tensor(1.0)))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """A reference checkpoint's dense ``adj`` [N, N] (continuous values) is reduced to the tracked pairs, and the engine's
        graph to its binarisation; this module's own checkpoints hold the tracked values as they are."""
        key = prefix + "adj"
        adj = state_dict.get(key)
        if adj is not None and adj.dim() == 2:
            if tuple(adj.shape) != (self.num_nodes, self.num_nodes):
                error_msgs.append(f"adj has shape {tuple(adj.shape)}, expected ({self.num_nodes}, {self.num_nodes})")
            else:
                dense = adj.to(torch.float32)
                ai = self.adj_index.to(dense.device)
                state_dict[key] = dense[ai[0], ai[1]]
                eff = 0.5 * (dense + dense.T) if self.symmetric else dense
                on = eff > self.threshold
                on.fill_diagonal_(False)
                self.edge_index = on.nonzero().t().to(torch.int64).contiguous().to(self.edge_index.device)
                self._engine = None
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    # -- helpers on the tracked pattern --------------------------------------------------------------------------------
    def _keys(self) -> torch.Tensor:
        return self.adj_index[0] * self.num_nodes + self.adj_index[1]

    def _effective(self) -> torch.Tensor:
        """What is thresholded: the values, or (adj + adj^T) / 2 of a symmetric model (models.py:105-106)."""
        v = self.adj.detach()
        if not self.symmetric:
            return v
        keys = self._keys()
        tpos = torch.searchsorted(keys, self.adj_index[1] * self.num_nodes + self.adj_index[0])
        return 0.5 * (v + v[tpos])  # (the tracked pattern of a symmetric model contains both orientations)

    def full_adj(self) -> torch.Tensor:
        """Dense binarised adjacency (models.py:99-101, with the propagated self loops), small graphs only."""
        return super().full_adj()

    def dense_adj(self) -> torch.Tensor:
        """The continuous parameter as the reference stores it (dense [N, N], untracked pairs 0, diagonal 1): small graphs."""
        out = torch.zeros(self.num_nodes, self.num_nodes, device=self.adj.device)
        out[self.adj_index[0], self.adj_index[1]] = self.adj.detach()
        out.fill_diagonal_(1.0)
        return out

    # -- the structure-learning step -----------------------------------------------------------------------------------
    @torch.no_grad()
    def adj_backward(self, la, train_loader, process_group=None):
        """``neg_marglik.backward()`` of the loop (gnn/marglik_training.py:212-216): ``adj.grad`` <- the gradient of the fitted
        posterior's negative log marginal likelihood w.r.t. the tracked entries, through normalize_adj, the overwritten
        diagonal, the STE (identity times ``grad_adj_mask``, gnn/models/utils.py:67-71) and the symmetric parameterisation.
        Accumulates into an existing ``.grad`` like autograd does.  Returns the negative log marginal likelihood."""
        eng = self.engine
        keys = self._keys()
        sr, sc = eng.export_adj()
        skeys = sr * self.num_nodes + sc
        stored = torch.isin(keys, skeys)  # tracked pairs the engine currently stores
        ci, cj = self.adj_index[0][~stored], self.adj_index[1][~stored]
        if self.symmetric:  # one orientation per unordered pair: the call returns the symmetrised value for both
            half = ci < cj
            ci, cj = ci[half], cj[half]
        cand = torch.stack([ci, cj]) if ci.numel() else None
        res = la.neg_marglik_adj_grad(train_loader, process_group=process_group, candidates=cand)
        value, _, gs = res[0], res[1], res[2]
        g = torch.zeros_like(self.adj)
        pos = torch.searchsorted(keys, skeys).clamp(max=max(keys.numel() - 1, 0))
        hit = keys[pos] == skeys  # (the diagonal and untracked stored pairs -- none by construction -- are skipped)
        g[pos[hit]] = gs[hit]
        if cand is not None:
            gc = res[3]
            ckeys = ci * self.num_nodes + cj
            g[torch.searchsorted(keys, ckeys)] = gc
            if self.symmetric:
                g[torch.searchsorted(keys, cj * self.num_nodes + ci)] = gc
        if self.train_masked_update:
            g = g * self.grad_adj_mask
        self.adj.grad = g if self.adj.grad is None else self.adj.grad + g
        return value

    @torch.no_grad()
    def apply_adj(self) -> int:
        """After ``adj_optimizer.step()``: re-binarise the tracked values (``> threshold``, models.py:103-116) and flip the
        entries of the engine's graph whose side changed.  Returns the number of directed entries flipped."""
        eng = self.engine
        want = self._effective() > self.threshold
        sr, sc = eng.export_adj()
        have = torch.isin(self._keys(), sr * self.num_nodes + sc)
        flip = want != have
        n = int(flip.sum())
        if n:
            eng.update_adjacency(self.adj_index[0][flip], self.adj_index[1][flip], want[flip])
            self.edge_index = eng.adj_to_edge_index()
        return n
