"""Python handle on one ``lgnn_ctx`` (graph + bound model + caches) of the HIP library.

PyTorch-ROCm is used for device memory and streams only: every tensor is handed over as a raw
device pointer, the current torch stream as a ``hipStream_t``.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib

KINDS = {"gcn": _lib.KIND_GCN, "sage": _lib.KIND_SAGE}
ACTS = {"relu": _lib.ACT_RELU, "tanh": _lib.ACT_TANH}
LIKS = {"classification": _lib.LIK_CLASSIFICATION, "regression": _lib.LIK_REGRESSION}
NORMS = {None: _lib.NORM_NONE, "layer": _lib.NORM_LAYER, "batch": _lib.NORM_BATCH}


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_ptr(t: torch.Tensor, dtype, what: str) -> C.c_void_p:
    if not t.is_cuda:
        raise _lib.HipLibraryError(f"{what} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{what} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    return C.c_void_p(t.data_ptr())


def _no_paths_default() -> bool:
    """LGNN_NO_PATHS=1 (read per call): keep the class-plane route of the 2-layer GCN KFAC for A/B timing."""
    import os
    return os.environ.get("LGNN_NO_PATHS", "") not in ("", "0")


def kfac_plan(kind: str, dims: Sequence[int], num_nodes: int, nnz: int, act: str = "relu", fuse: bool = True,
              workspace_limit: int = 32 << 30, paths: bool | None = None) -> dict:
    """Which kernels ``lgnn_kfac_accumulate`` would run for a model of this shape (host-only query, no GPU work):
    per backward step l = L-1 .. 1 whether the fused SpMM^T -> Gram kernel and the compacted backward GEMM are used,
    whether the second plane buffer is needed and how many class planes fit one chunk of the workspace.

    ``paths`` (and with it ``fused`` / ``backgemm`` / ``need_pong`` of the first level) says what the SHAPE allows: whether a call
    really takes the path route also depends on the graph and the call -- the expected number of 2-hop paths per node of the batch
    (hub-heavy graphs keep the class planes), a Fisher-type seed, ``res`` / ``norm``.  ``GraphEngine.last_kfac_used_paths`` /
    ``lgnn_kfac_last_route`` report what the last call did."""
    lib = _lib.load()
    L = len(dims) - 1
    out = (C.c_int64 * (4 + L))()
    no_paths = _no_paths_default() if paths is None else not paths
    rc = lib.lgnn_kfac_plan(KINDS[kind], L, (C.c_int64 * (L + 1))(*dims), int(num_nodes), int(nnz), ACTS[act],
                            (0 if fuse else _lib.FLAG_NO_FUSE) | (_lib.FLAG_NO_PATHS if no_paths else 0),
                            int(workspace_limit), out)
    _lib.check(rc, "lgnn_kfac_plan")
    return {"seeds_on_the_fly": bool(out[0]), "sage_compact": bool(out[1]), "need_pong": bool(out[2]),
            "classes_per_chunk": int(out[3]), "fused": [bool(out[4 + l] & 1) for l in range(L)],
            "backgemm": [bool(out[4 + l] & 2) for l in range(L)], "paths": bool(out[4 + L - 1] & 4)}


class _IdentityKey:
    """Hashable wrapper comparing by object identity (tensors define elementwise ``==``)."""
    __slots__ = ("obj",)

    def __init__(self, obj):
        self.obj = obj

    def __hash__(self):
        return id(self.obj)

    def __eq__(self, other):
        return isinstance(other, _IdentityKey) and other.obj is self.obj


class GraphEngine:
    """Graph ingest + model binding + per-batch curvature accumulation on one GPU."""
    supports_shares = True  # kfac_accumulate(share=...): lgnn_kfac_accumulate_share

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, kind: str = "gcn", symmetric: bool = False):
        self.lib = _lib.load()
        if kind not in KINDS:
            raise ValueError(f"kind must be one of {list(KINDS)}")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must have shape [2, E]")
        ei = edge_index.contiguous()
        _dev_ptr(ei, torch.int64, "edge_index")  # raises for CPU tensors: there is no CPU path
        self.device = ei.device
        self.kind = kind
        self.num_nodes = int(num_nodes)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.lgnn_create(
                C.byref(self._h), self.num_nodes, _dev_ptr(ei, torch.int64, "edge_index"), ei.shape[1],
                KINDS[kind], int(bool(symmetric)), _stream(self.device))
        _lib.check(rc, "lgnn_create")
        self._bound = None  # keeps the bound tensors alive (borrowed pointers)
        self.dims = None

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.lgnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- graph ----------------------------------------------------------------------------------
    @property
    def nnz(self) -> int:
        return int(self.lib.lgnn_nnz(self._h))

    @property
    def is_symmetric(self) -> bool:
        return bool(self.lib.lgnn_is_symmetric(self._h))

    @property
    def num_long_rows(self) -> int:
        """Rows with more than 64 stored entries that the 256-wide fused kernel takes from the side kernel (-1: not built)."""
        return int(self.lib.lgnn_num_long_rows(self._h))

    @property
    def last_kfac_used_paths(self) -> bool:
        """Whether the last KFAC accumulate took the two-hop path route (csrc/paths.hip) rather than class planes."""
        return bool(self.lib.lgnn_kfac_last_route(self._h))

    def export_adj(self):
        nnz = self.nnz
        rows = torch.empty(nnz, dtype=torch.int64, device=self.device)
        cols = torch.empty(nnz, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.lgnn_export_adj(self._h, rows.data_ptr(), cols.data_ptr(), _stream(self.device)),
                   "lgnn_export_adj")
        return rows, cols

    def adj_to_edge_index(self) -> torch.Tensor:
        n = C.c_int64(0)
        _lib.check(self.lib.lgnn_adj_to_edge_index(self._h, None, C.byref(n), _stream(self.device)),
                   "lgnn_adj_to_edge_index")
        out = torch.empty(2, n.value, dtype=torch.int64, device=self.device)
        if n.value:
            _lib.check(self.lib.lgnn_adj_to_edge_index(self._h, out.data_ptr(), C.byref(n), _stream(self.device)),
                       "lgnn_adj_to_edge_index")
        return out

    def update_adjacency(self, rows: torch.Tensor, cols: torch.Tensor, state: torch.Tensor):
        """Edit the stored 0/1 adjacency in place (``lgnn_update_adjacency``): pairs ``(rows[k], cols[k])`` and the state each
        shall have afterwards (True / 1 = stored).  A structure-learning step (gnn/marglik_training.py:211-221) without a
        re-ingest; the cached forward pass and everything else derived from the graph is dropped by the library."""
        rows, cols = rows.to(self.device, torch.int64).contiguous(), cols.to(self.device, torch.int64).contiguous()
        state = state.to(self.device, torch.uint8).contiguous()
        if not (rows.shape == cols.shape == state.shape) or rows.dim() != 1:
            raise ValueError("rows, cols, state must be 1-D and of one length")
        if rows.numel() == 0:
            return
        with torch.cuda.device(self.device):
            _lib.check(self.lib.lgnn_update_adjacency(self._h, _dev_ptr(rows, torch.int64, "rows"),
                                                      _dev_ptr(cols, torch.int64, "cols"),
                                                      _dev_ptr(state, torch.uint8, "state"), rows.numel(),
                                                      _stream(self.device)), "lgnn_update_adjacency")
        self._graph_edits = getattr(self, "_graph_edits", 0) + 1  # (the CSR buffers may have moved: captured graphs are stale)

    def export_propagation(self):
        nnz = self.nnz
        rows = torch.empty(nnz, dtype=torch.int64, device=self.device)
        cols = torch.empty(nnz, dtype=torch.int64, device=self.device)
        vals = torch.empty(nnz, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.lgnn_export_propagation(self._h, rows.data_ptr(), cols.data_ptr(), vals.data_ptr(),
                                                    _stream(self.device)), "lgnn_export_propagation")
        return rows, cols, vals

    # -- model ----------------------------------------------------------------------------------
    def bind(self, X: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
             act: str = "relu", likelihood: str = "classification", res_weights=None, res_biases=None, norm=None,
             norm_weight=None, norm_bias=None, norm_mean=None, norm_var=None, norm_eps: float = 1e-5):
        """``res_*`` / ``norm*``: the optional pieces of ``BaseGNN.forward`` (gnn/models/base_gnn.py:86-113, 141-149), one
        entry per hidden layer: ``res.{l}`` Linear [dims[l+1], dims[l]] and ``norms.{l}`` ("layer": LayerNorm weight / bias,
        "batch": eval-mode BatchNorm1d weight / bias + running mean / variance).  The ``res.{l}`` parameters follow all
        ``convs.*`` in every per-parameter result."""
        L = len(weights)
        if L != len(biases) or L == 0:
            raise ValueError("need one bias per weight")
        if act not in ACTS:
            raise NotImplementedError(f"activation {act!r} is not supported (relu, tanh)")
        mult = 2 if self.kind == "sage" else 1
        dims = [X.shape[1]]
        for w, b in zip(weights, biases):
            if w.shape[1] != mult * dims[-1] or b.shape[0] != w.shape[0]:
                raise ValueError(f"weight shape {tuple(w.shape)} does not chain from width {dims[-1]}")
            dims.append(w.shape[0])
        if X.shape[0] != self.num_nodes:
            raise ValueError("X must have one row per node")
        Xp = _dev_ptr(X, torch.float32, "X")
        wp = [_dev_ptr(w, torch.float32, "weight").value for w in weights]
        bp = [_dev_ptr(b, torch.float32, "bias").value for b in biases]
        dims_c = (C.c_int64 * (L + 1))(*dims)
        rc = self.lib.lgnn_bind_model(self._h, L, dims_c, _lib.ptr_array(wp), _lib.ptr_array(bp), Xp, ACTS[act],
                                      LIKS[likelihood])
        _lib.check(rc, "lgnn_bind_model")
        self._bound = (X, list(weights), list(biases))
        self._bind_opts = (act, likelihood)
        self.dims = dims
        self.in_dims = [mult * d for d in dims[:-1]]
        if norm in ("none", ""):
            norm = None
        if norm not in NORMS:
            raise ValueError(f"Unknown normalization type: {norm}")
        self._extras = dict(res_weights=res_weights, res_biases=res_biases, norm=norm, norm_weight=norm_weight,
                            norm_bias=norm_bias, norm_mean=norm_mean, norm_var=norm_var, norm_eps=norm_eps)
        self.has_res = bool(res_weights) and L > 1
        self.norm = norm if L > 1 else None
        if self.has_res or self.norm is not None:
            def ptrs(ts, what, shape_of):
                if ts is None:
                    return None
                if len(ts) != L - 1:
                    raise ValueError(f"need one {what} per hidden layer")
                for l, t in enumerate(ts):
                    if tuple(t.shape) != shape_of(l):
                        raise ValueError(f"{what}[{l}] has shape {tuple(t.shape)}, expected {shape_of(l)}")
                return _lib.ptr_array([_dev_ptr(t, torch.float32, what).value for t in ts])
            hid = lambda l: (dims[l + 1],)  # noqa: E731
            if self.has_res and (res_biases is None or len(res_biases) != L - 1):
                raise ValueError("need one res bias per res weight")
            if self.norm is not None and (norm_weight is None or norm_bias is None):
                raise ValueError("norm weight / bias missing")
            if self.norm == "batch" and (norm_mean is None or norm_var is None):
                raise ValueError("BatchNorm needs its running statistics")
            rc = self.lib.lgnn_bind_extras(
                self._h, ptrs(res_weights if self.has_res else None, "res weight", lambda l: (dims[l + 1], dims[l])),
                ptrs(res_biases if self.has_res else None, "res bias", hid), NORMS[self.norm],
                ptrs(norm_weight if self.norm else None, "norm weight", hid), ptrs(norm_bias if self.norm else None, "norm bias", hid),
                ptrs(norm_mean if self.norm == "batch" else None, "norm running mean", hid),
                ptrs(norm_var if self.norm == "batch" else None, "norm running var", hid), float(norm_eps))
            _lib.check(rc, "lgnn_bind_extras")
        # KFAC blocks in parameter order: (in, out) of convs.{0..L-1}.lin, then of res.{0..L-2}
        self.block_dims = list(zip(self.in_dims, dims[1:])) + ([(dims[l], dims[l + 1]) for l in range(L - 1)] if self.has_res else [])
        self._versions = self._param_versions()

    @property
    def has_extras(self) -> bool:
        return self.has_res or self.norm is not None

    def set_likelihood(self, likelihood: str):
        """Switch the bound model between the classification and the regression likelihood (re-binds the same tensors)."""
        if self._bound is None:
            raise _lib.HipLibraryError("no model bound")
        if self._bind_opts[1] != likelihood:
            X, ws, bs = self._bound
            self.bind(X, ws, bs, self._bind_opts[0], likelihood, **self._extras)

    @property
    def likelihood(self) -> str:
        return self._bind_opts[1]

    def _labels(self, y: torch.Tensor, M: int):
        """int64 class ids [M] (classification) or fp32 targets [M, C] (regression) as a device pointer."""
        if self.likelihood == "regression":
            if y.numel() != M * self.dims[-1]:
                raise ValueError(f"regression targets must have shape [{M}, {self.dims[-1]}]")
            y = y.to(torch.float32).contiguous()
            self._keep = y
            return _dev_ptr(y, torch.float32, "y")
        y = y.contiguous()
        return _dev_ptr(y, torch.int64, "y")

    def feature_token(self):
        """Exact identity of the bound feature tensor for callers that cache what depends on X only: the tensor object itself
        (holding it keeps its storage from being recycled under the token), its version counter and the engine's own
        ``rebind`` count.  Writes through ``X.data`` (``X.data.mul_()``, ``X.data.copy_()``) bump no version counter: after
        them call ``rebind()`` -- the same contract as for parameters written through ``p.data`` (``invalidate()``)."""
        X = self._bound[0]
        return (_IdentityKey(X), X._version, tuple(X.shape), getattr(self, "_rebinds", 0))

    def _param_versions(self):
        _, ws, bs = self._bound
        extra = [t for k in ("res_weights", "res_biases", "norm_weight", "norm_bias", "norm_mean", "norm_var")
                 for t in (self._extras.get(k) or [])]
        X = self._bound[0]
        return [(X.data_ptr(), X._version)] + [(t.data_ptr(), t._version) for t in (*ws, *bs, *extra)]

    def invalidate(self):
        """Drop the cached forward / input Grams.  Needed by hand only after writes the version counters cannot see
        (``p.data.add_()`` and friends); optimizer steps, ``p.add_()`` under ``no_grad`` and storage swaps are detected."""
        _lib.check(self.lib.lgnn_invalidate(self._h), "lgnn_invalidate")

    def rebind(self):
        """Bind the same tensors again: drops everything derived from X too (padded copy, P X, X^T X and the callers'
        caches keyed by ``feature_token``).  For feature writes through ``X.data``, which no version counter records."""
        if self._bound is None:
            raise _lib.HipLibraryError("no model bound")
        X, ws, bs = self._bound
        self._rebinds = getattr(self, "_rebinds", 0) + 1
        self.bind(X, ws, bs, *self._bind_opts, **self._extras)
        self.invalidate()

    def _sync_versions(self):
        """In-place updates of the bound parameters (optimizer steps) invalidate the cached forward."""
        if self._bound is None:
            raise _lib.HipLibraryError("no model bound")
        v = self._param_versions()
        if v != self._versions:
            if [p for p, _ in v] != [p for p, _ in self._versions] or v[0] != self._versions[0]:
                # storage replaced (``param.data = ...``, e.g. torch.nn.utils.vector_to_parameters): bind the new pointers.
                # The feature tensor changed in place: lgnn_invalidate keeps what depends on X only (padded copy, P X,
                # X^T X), a re-bind drops it
                X, ws, bs = self._bound
                self.bind(X, ws, bs, *self._bind_opts, **self._extras)
            self.invalidate()
            self._versions = v

    @property
    def num_layers(self):
        return len(self.dims) - 1

    @property
    def n_params(self):
        return sum(i * o + o for i, o in self.block_dims)

    def device_bytes(self) -> int:
        return int(self.lib.lgnn_device_bytes(self._h))

    def set_workspace_limit(self, nbytes: int):
        _lib.check(self.lib.lgnn_set_workspace_limit(self._h, int(nbytes)), "lgnn_set_workspace_limit")
        self._ws_limit = int(nbytes)

    def kfac_plan(self, fuse: bool = True, paths: bool | None = None) -> dict:
        """The kernel choices a KFAC accumulate on the bound model makes (see ``kfac_plan``)."""
        if self.has_extras:
            fuse = False  # res / norm models take the unfused route (csrc/resnorm.hip)
        return kfac_plan(self.kind, self.dims, self.num_nodes, self.nnz, self._bind_opts[0], fuse,
                         getattr(self, "_ws_limit", 32 << 30), paths)

    # -- forward --------------------------------------------------------------------------------
    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        self._sync_versions()
        idx = idx.contiguous()
        out = torch.empty(idx.shape[0], self.dims[-1], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.lgnn_forward(self._h, _dev_ptr(idx, torch.int64, "idx"), idx.shape[0], out.data_ptr(),
                                         _stream(self.device)), "lgnn_forward")
        return out

    def forward_all(self) -> torch.Tensor:
        self._sync_versions()
        out = torch.empty(self.num_nodes, self.dims[-1], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.lgnn_forward_all(self._h, out.data_ptr(), _stream(self.device)), "lgnn_forward_all")
        return out

    # -- curvature ------------------------------------------------------------------------------
    def new_kfac_buffers(self):
        """Zeroed caller-owned accumulators, as ONE flat fp32 buffer [A_0|B_0|...|A_{L-1}|B_{L-1}|loss]
        (one all-reduce suffices) plus per-factor views."""
        sizes = []
        for i, o in self.block_dims:
            sizes += [i * i, o * o]
        flat = torch.zeros(sum(sizes) + 1, dtype=torch.float32, device=self.device)
        views, off = [], 0
        for l, (i, o) in enumerate(self.block_dims):
            A = flat[off:off + i * i].view(i, i); off += i * i
            B = flat[off:off + o * o].view(o, o); off += o * o
            views.append((A, B))
        loss = flat[off:off + 1]
        return flat, views, loss

    def kfac_accumulate(self, idx, y, n_train: int, views, loss, fork_exact: bool = True, fuse: bool = True,
                        classes: tuple[int, int] | None = None, paths: bool | None = None,
                        share: tuple[int, int, int] | None = None):
        """Add one batch's factors into the caller-owned buffers.  ``classes=(begin, end)`` restricts the call to
        that range of class columns (an exact additive share of the batch; the share with class 0 also adds
        the loss and the A increment).  ``share=(begin, end, count)``: parts ``[begin, end)`` of ``count`` equal parts of
        the batch's work, cut the way the route in use splits best (``lgnn_kfac_accumulate_share``: destination-node ranges
        on the path routes, class ranges otherwise) -- what a data-parallel caller deals to its ranks."""
        self._sync_versions()
        idx = idx.contiguous()
        yp = self._labels(y, idx.shape[0])
        # 2-layer GCN: two-hop path route (csrc/paths.hip).  paths=None: the library decides (shape and the batch's expected
        # paths per node; LGNN_NO_PATHS=1 keeps the planes); True / False force one route where the shape allows
        no_paths = _no_paths_default() if paths is None else not paths
        flags = (_lib.FLAG_FORK_EXACT_SEED if fork_exact else 0) | (0 if fuse else _lib.FLAG_NO_FUSE) | \
            (_lib.FLAG_NO_PATHS if no_paths else 0) | (_lib.FLAG_FORCE_PATHS if paths else 0)
        A = _lib.ptr_array([a.data_ptr() for a, _ in views])
        B = _lib.ptr_array([b.data_ptr() for _, b in views])
        if share is not None:
            if classes is not None:
                raise ValueError("pass either classes or share")
            rc = self.lib.lgnn_kfac_accumulate_share(
                self._h, _dev_ptr(idx, torch.int64, "idx"), yp, idx.shape[0], int(n_train),
                flags, int(share[0]), int(share[1]), int(share[2]), A, B, loss.data_ptr(), _stream(self.device))
            _lib.check(rc, "lgnn_kfac_accumulate_share")
            return
        cb, ce = (0, self.dims[-1]) if classes is None else (int(classes[0]), int(classes[1]))
        rc = self.lib.lgnn_kfac_accumulate_classes(
            self._h, _dev_ptr(idx, torch.int64, "idx"), yp, idx.shape[0], int(n_train),
            flags, cb, ce, A, B, loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_kfac_accumulate_classes")

    def kfac_accumulate_fisher(self, idx, y_seed, y_loss, n_train: int, views, loss, resid_scale: float = 1.0,
                               b_scale: float = 1.0, fuse: bool = True):
        """Empirical / Monte-Carlo Fisher KFAC of one batch: ONE backward pass seeded with
        ``resid_scale * d loss(f, y_seed) / d f``; ``B += b_scale * g^T g``.  ``y_loss`` (the true labels) makes this
        call also add the loss and the A increment -- pass it with the first draw of a batch, ``None`` afterwards."""
        self._sync_versions()
        idx = idx.contiguous()
        M = idx.shape[0]
        ys = self._labels(y_seed, M)
        keep = self._keep if self.likelihood == "regression" else None
        yl = self._labels(y_loss, M) if y_loss is not None else None
        self._keep = (keep, getattr(self, "_keep", None))
        A = _lib.ptr_array([a.data_ptr() for a, _ in views])
        B = _lib.ptr_array([b.data_ptr() for _, b in views])
        rc = self.lib.lgnn_kfac_accumulate_fisher(
            self._h, _dev_ptr(idx, torch.int64, "idx"), ys, yl, M, int(n_train), 0 if fuse else _lib.FLAG_NO_FUSE,
            float(resid_scale), float(b_scale), A, B, loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_kfac_accumulate_fisher")

    def ef_accumulate(self, idx, y_seed, y_loss=None, resid_scale: float = 1.0, scale: float = 1.0,
                      diag: torch.Tensor | None = None, full: torch.Tensor | None = None, grads: bool = False,
                      loss: torch.Tensor | None = None):
        """Per-sample loss gradients ``G = J^T r`` of a batch (returned as [M, P] when ``grads``), ``diag += scale *
        sum G^2``, ``full += scale * G^T G``; ``loss += loss(model(idx), y_loss)`` when ``y_loss`` is given."""
        self._sync_versions()
        idx = idx.contiguous()
        M = idx.shape[0]
        ys = self._labels(y_seed, M)
        keep = self._keep if self.likelihood == "regression" else None
        yl = self._labels(y_loss, M) if y_loss is not None else None
        self._keep = (keep, getattr(self, "_keep", None))
        G = torch.empty(M, self.n_params, dtype=torch.float32, device=self.device) if grads else None
        rc = self.lib.lgnn_ef_accumulate(
            self._h, _dev_ptr(idx, torch.int64, "idx"), ys, yl, M, float(resid_scale), float(scale),
            None if diag is None else _dev_ptr(diag, torch.float32, "diag"),
            None if full is None else _dev_ptr(full, torch.float32, "full"),
            None if G is None else G.data_ptr(), None if loss is None else loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_ef_accumulate")
        return G

    def diag_accumulate(self, idx, y, diag: torch.Tensor, loss: torch.Tensor):
        """diag += diagonal GGN of the batch, loss += raw loss sum (CE, or the MSE sum for a regression binding)."""
        self._sync_versions()
        idx = idx.contiguous()
        rc = self.lib.lgnn_diag_accumulate(
            self._h, _dev_ptr(idx, torch.int64, "idx"), self._labels(y, idx.shape[0]), idx.shape[0], 0,
            _dev_ptr(diag, torch.float32, "diag"), loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_diag_accumulate")

    def full_accumulate(self, idx, y, H: torch.Tensor, loss: torch.Tensor):
        """H [P, P] += full GGN of the batch over all weights (regression binding: sum J^T J), loss += raw loss sum."""
        self._sync_versions()
        idx = idx.contiguous()
        rc = self.lib.lgnn_full_accumulate(
            self._h, _dev_ptr(idx, torch.int64, "idx"), self._labels(y, idx.shape[0]), idx.shape[0],
            _dev_ptr(H, torch.float32, "H"), loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_full_accumulate")

    def lastlayer_full_accumulate(self, idx, y, H: torch.Tensor, loss: torch.Tensor):
        self._sync_versions()
        idx, y = idx.contiguous(), y.contiguous()
        rc = self.lib.lgnn_lastlayer_full_accumulate(
            self._h, _dev_ptr(idx, torch.int64, "idx"), _dev_ptr(y, torch.int64, "y"), idx.shape[0],
            _dev_ptr(H, torch.float32, "H"), loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_lastlayer_full_accumulate")

    def lastlayer_features(self, idx: torch.Tensor):
        """(phi [M, D], s [M], f [M, C]): the last Linear's input row seen from the batch node, the bias scale, the logits."""
        self._sync_versions()
        idx = idx.contiguous()
        M, D, C = idx.shape[0], self.in_dims[-1], self.dims[-1]
        out = torch.empty(M, D + 1, dtype=torch.float32, device=self.device)
        f = torch.empty(M, C, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.lgnn_lastlayer_features(self._h, _dev_ptr(idx, torch.int64, "idx"), M, out.data_ptr(), f.data_ptr(),
                                                    _stream(self.device)), "lgnn_lastlayer_features")
        return out[:, :D], out[:, D], f

    def new_lastlayer_pair_buffers(self):
        """Zeroed pair-major accumulators of the last-layer full GGN as ONE flat buffer [S | Sb | loss] (one all-reduce)
        plus views: S [Q, D, D], Sb [Q, D + 1], loss [1], Q = C (C + 1) / 2."""
        C, D = self.dims[-1], self.in_dims[-1]
        Q = C * (C + 1) // 2
        flat = torch.zeros(Q * (D * D + D + 1) + 1, dtype=torch.float32, device=self.device)
        S = flat[:Q * D * D].view(Q, D, D)
        Sb = flat[Q * D * D:Q * (D * D + D + 1)].view(Q, D + 1)
        return flat, S, Sb, flat[-1:]

    def lastlayer_pairs_accumulate(self, idx, y, S: torch.Tensor, Sb: torch.Tensor, loss: torch.Tensor):
        self._sync_versions()
        idx, y = idx.contiguous(), y.contiguous()
        rc = self.lib.lgnn_lastlayer_pairs_accumulate(
            self._h, _dev_ptr(idx, torch.int64, "idx"), _dev_ptr(y, torch.int64, "y"), idx.shape[0],
            _dev_ptr(S, torch.float32, "S"), _dev_ptr(Sb, torch.float32, "Sb"), loss.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_lastlayer_pairs_accumulate")

    def lastlayer_pairs_place(self, S: torch.Tensor, Sb: torch.Tensor, H: torch.Tensor):
        """H [P_ll, P_ll] += the accumulated blocks (upper triangle), then the mirror pass."""
        rc = self.lib.lgnn_lastlayer_pairs_place(self._h, _dev_ptr(S, torch.float32, "S"), _dev_ptr(Sb, torch.float32, "Sb"),
                                                 _dev_ptr(H, torch.float32, "H"), _stream(self.device))
        _lib.check(rc, "lgnn_lastlayer_pairs_place")

    def jacobians(self, idx: torch.Tensor):
        """(J [M, C, P], f [M, C]): per-sample Jacobians of the logits w.r.t. all parameters, parameters in
        module order (weight row major, bias) -- laplace/curvature/curvature.py:89-130."""
        self._sync_versions()
        idx = idx.contiguous()
        M, C = idx.shape[0], self.dims[-1]
        J = torch.empty(M, C, self.n_params, dtype=torch.float32, device=self.device)
        f = torch.empty(M, C, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.lgnn_jacobians(self._h, _dev_ptr(idx, torch.int64, "idx"), M, J.data_ptr(), f.data_ptr(),
                                           _stream(self.device)), "lgnn_jacobians")
        return J, f

    # -- marginal-likelihood gradient w.r.t. the adjacency ("next" row 8(f)-4) ---------------------------------
    @staticmethod
    def _cand_ptrs(cand):
        """cand = (a int32 [K], b int32 [K], accumulator fp32 [K]) in the propagation matrix's coordinates, or None."""
        if cand is None:
            return None, None, 0, None
        a, b, acc = cand
        return _dev_ptr(a, torch.int32, "cand_a"), _dev_ptr(b, torch.int32, "cand_b"), int(a.shape[0]), \
            _dev_ptr(acc, torch.float32, "grad_cand")

    def adjgrad_batch(self, idx, y, gammas_B, grad_P: torch.Tensor, out_bar: torch.Tensor, fork_exact: bool = True,
                      loss_scale: float = 1.0, cand=None):
        """Add one batch's terms to ``grad_P`` [nnz] (stored entries of the propagation matrix), ``out_bar`` [N, C] and,
        for candidate pairs ``cand = (a, b, acc)``, to ``acc``."""
        self._sync_versions()
        idx, y = idx.contiguous(), y.contiguous()
        self._keep = [g.contiguous() for g in gammas_B]
        gb = _lib.ptr_array([_dev_ptr(g, torch.float32, "gamma_B").value for g in self._keep])
        ca, cb, K, cacc = self._cand_ptrs(cand)
        rc = self.lib.lgnn_kfac_adjgrad_batch(
            self._h, _dev_ptr(idx, torch.int64, "idx"), _dev_ptr(y, torch.int64, "y"), idx.shape[0],
            _lib.FLAG_FORK_EXACT_SEED if fork_exact else 0, gb, float(loss_scale),
            _dev_ptr(grad_P, torch.float32, "grad_P"), _dev_ptr(out_bar, torch.float32, "out_bar"), ca, cb, K, cacc,
            _stream(self.device))
        _lib.check(rc, "lgnn_kfac_adjgrad_batch")

    def adjgrad_finish(self, out_bar: torch.Tensor, gammas_A, a_scale: float, grad_P: torch.Tensor, cand=None):
        """Forward-pass terms + normalize_adj / STE backward: gradient w.r.t. the stored entries of the 0/1 adjacency
        (``export_adj`` order); with candidates also ``d / d adj[i, j]`` of every listed pair (second return value)."""
        self._sync_versions()
        keep = [g.contiguous() for g in gammas_A]
        ga = _lib.ptr_array([_dev_ptr(g, torch.float32, "gamma_A").value for g in keep])
        out = torch.zeros(self.nnz, dtype=torch.float32, device=self.device)
        ca, cb, K, cacc = self._cand_ptrs(cand)
        cout = torch.zeros(K, dtype=torch.float32, device=self.device) if K else None
        rc = self.lib.lgnn_adjgrad_finish(self._h, _dev_ptr(out_bar, torch.float32, "out_bar"), ga, float(a_scale),
                                          _dev_ptr(grad_P, torch.float32, "grad_P"), out.data_ptr(), ca, cb, K, cacc,
                                          None if cout is None else cout.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_adjgrad_finish")
        return out if cand is None else (out, cout)

    def diag_adjgrad_batch(self, idx, y, gamma: torch.Tensor, grad_P: torch.Tensor, out_bar: torch.Tensor, h1_bar: torch.Tensor,
                           e_bar: torch.Tensor, loss_scale: float = 1.0, cand=None):
        """Diagonal posterior (2-layer GCN): add one batch's terms to ``grad_P`` [nnz], ``out_bar`` [N, C], ``h1_bar`` [N, H],
        ``e_bar`` [N, F + 1] and the candidates' accumulator; ``gamma`` [n_params] = d(-marglik)/dH (lgnn_diag_adjgrad_batch)."""
        self._sync_versions()
        idx, y = idx.contiguous(), y.contiguous()
        self._keep = gamma.contiguous()
        ca, cb, K, cacc = self._cand_ptrs(cand)
        rc = self.lib.lgnn_diag_adjgrad_batch(
            self._h, _dev_ptr(idx, torch.int64, "idx"), _dev_ptr(y, torch.int64, "y"), idx.shape[0],
            _dev_ptr(self._keep, torch.float32, "gamma"), float(loss_scale), _dev_ptr(grad_P, torch.float32, "grad_P"),
            _dev_ptr(out_bar, torch.float32, "out_bar"), _dev_ptr(h1_bar, torch.float32, "h1_bar"),
            _dev_ptr(e_bar, torch.float32, "e_bar"), ca, cb, K, cacc, _stream(self.device))
        _lib.check(rc, "lgnn_diag_adjgrad_batch")

    def diag_adjgrad_finish(self, out_bar: torch.Tensor, h1_bar: torch.Tensor, e_bar: torch.Tensor, grad_P: torch.Tensor,
                            cand=None):
        """The forward-pass terms of ``out_bar`` / ``h1_bar`` / ``e_bar`` + normalize_adj / STE backward (as ``adjgrad_finish``)."""
        self._sync_versions()
        out = torch.zeros(self.nnz, dtype=torch.float32, device=self.device)
        ca, cb, K, cacc = self._cand_ptrs(cand)
        cout = torch.zeros(K, dtype=torch.float32, device=self.device) if K else None
        rc = self.lib.lgnn_diag_adjgrad_finish(self._h, _dev_ptr(out_bar, torch.float32, "out_bar"),
                                               _dev_ptr(h1_bar, torch.float32, "h1_bar"), _dev_ptr(e_bar, torch.float32, "e_bar"),
                                               _dev_ptr(grad_P, torch.float32, "grad_P"), out.data_ptr(), ca, cb, K, cacc,
                                               None if cout is None else cout.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_diag_adjgrad_finish")
        return out if cand is None else (out, cout)

    def glm_variance(self, idx: torch.Tensor, S0, S1, kappa, QA0=None, QB0=None, QA1=None, QB1sq=None, out_map=None):
        """Matrix-free GLM predictive of a 2-layer GCN / GraphSAGE: (f_mu [M, C], diag(J P^-1 J^T) [M, C]) from the closed-form
        Jacobian (see include/laplace_gnn_hip.h lgnn_glm_variance for the operand conventions).  ``out_map`` = E [Cm, C]:
        the variances of the linear map E f instead, [M, Cm]; S1 / QB1sq / kappa are then the caller's mapped operands
        (lgnn_glm_variance_mapped)."""
        self._sync_versions()
        idx = idx.contiguous()
        M, C = idx.shape[0], self.dims[-1]
        Cm = C if out_map is None else int(out_map.shape[0])
        f_mu = torch.empty(M, C, dtype=torch.float32, device=self.device)
        f_var = torch.empty(M, Cm, dtype=torch.float32, device=self.device)
        keep = [t.contiguous().to(torch.float32) if t is not None else None for t in (QA0, QB0, S0, QA1, S1, QB1sq, kappa)]
        ptr = [None if t is None else _dev_ptr(t, torch.float32, "posterior operand") for t in keep]
        if out_map is None:
            rc = self.lib.lgnn_glm_variance(self._h, _dev_ptr(idx, torch.int64, "idx"), M, *ptr, f_mu.data_ptr(),
                                            f_var.data_ptr(), _stream(self.device))
            _lib.check(rc, "lgnn_glm_variance")
            return f_mu, f_var
        if out_map.ndim != 2 or out_map.shape[1] != C:
            raise ValueError(f"out_map must be [rows, {C}]")
        W1m = (out_map.to(device=self.device, dtype=torch.float32) @ self._bound[1][-1].detach().to(torch.float32)).contiguous()
        rc = self.lib.lgnn_glm_variance_mapped(self._h, _dev_ptr(idx, torch.int64, "idx"), M, W1m.data_ptr(), Cm, *ptr,
                                               f_mu.data_ptr(), f_var.data_ptr(), _stream(self.device))
        _lib.check(rc, "lgnn_glm_variance_mapped")
        return f_mu, f_var

    def check_async_errors(self):
        """Synchronise and raise if any batch since the last check contained an invalid node id or label."""
        _lib.check(self.lib.lgnn_check_async_errors(self._h, _stream(self.device)), "lgnn_check_async_errors")

    def peek_async_errors(self):
        """The same report without synchronising: errors of the kernels that have finished (lgnn_peek_async_errors)."""
        _lib.check(self.lib.lgnn_peek_async_errors(self._h), "lgnn_peek_async_errors")

    # -- timing hook ----------------------------------------------------------------------------
    def enable_kernel_timing(self, on: bool = True):
        _lib.check(self.lib.lgnn_enable_kernel_timing(self._h, int(on)), "lgnn_enable_kernel_timing")
        self._timing_on = bool(on)  # (event records around the dominant kernel: such a fit is not captured into a graph)

    def kernel_timing_launches(self):
        """Durations [ms] of the dominant kernel's launches since timing was enabled, in launch order."""
        n = C.c_int64(0)
        _lib.check(self.lib.lgnn_kernel_timing_launches(self._h, None, 0, C.byref(n)), "lgnn_kernel_timing_launches")
        buf = (C.c_double * max(n.value, 1))()
        _lib.check(self.lib.lgnn_kernel_timing_launches(self._h, buf, n.value, C.byref(n)), "lgnn_kernel_timing_launches")
        return [buf[i] for i in range(n.value)]

    def kernel_timing(self):
        n, ms, planes = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        _lib.check(self.lib.lgnn_kernel_timing_read(self._h, C.byref(n), C.byref(ms), C.byref(planes)),
                   "lgnn_kernel_timing_read")
        return n.value, ms.value, planes.value
