"""``Kron`` / ``KronDecomposed``: the output container of the kron path.

Own implementation with the same observable behaviour as laplace/utils/matrix.py for what the
fit path touches: ``init_from_model`` (:33-72), ``__add__`` (:74-93), ``__mul__`` (:95-113,
``pow(scalar, 1/len(F))`` per factor), ``decompose`` (:118-145, ``symeig`` clamp semantics of
laplace/utils/utils.py:193-226), ``diag`` (:236-251), ``to_matrix`` (:253-271), ``logdet``.
"""
from __future__ import annotations

from math import pow
from typing import Iterable

import torch
from torch import nn


# Factors larger than the hand-written path's 256 take the library's syevd: 30 ms for the 1 433 x 1 433 input covariance of a
# Cora-shaped GCN, 100 ms for GraphSAGE's 2 866 x 2 866 -- the whole fit otherwise costs 3 ms.  A GCN's first input covariance is
# (batches / N_train) X^T X: it depends on neither the weights nor the adjacency, so in the fork's structure-learning loop
# (gnn/marglik_training.py:197-216: fit, marglik, gradient step on the adjacency, repeat) every fit after the first asks for the
# decomposition of the same matrix.  A caller that KNOWS a factor's identity passes a cache (a dict it owns) and an exact key per
# factor (``Kron.decompose(cache=..., cache_keys=...)``; KronLaplace keys a GCN's A_0 by the feature tensor's identity and
# version, the batch count and N_train): no approximate matching, no comparison of matrix contents, no global state.
_LARGE_EIG_KEEP = 4
_SMALL_EIG = 256  # lgnn_symeig_batched's one-workgroup path


def _symeig_group_hip(mats: list) -> list:
    """ONE ``lgnn_symeig_batched`` call over factors padded to their common size."""
    from . import _lib
    lib = _lib.load()
    dev = mats[0].device
    m = max(H.shape[0] for H in mats)
    A = torch.zeros(len(mats), m, m, device=dev, dtype=torch.float32)
    for b, H in enumerate(mats):
        n = H.shape[0]
        A[b, :n, :n] = H
        if n < m:
            # mean eigenvalue of H (stays on device): below every eigenvalue of a PSD factor once negated, yet of H's own
            # magnitude -- the solver's accuracy is relative to the norm of the whole padded matrix
            tr = H.diagonal().sum() / n
            A[b].diagonal()[n:] = -torch.where(tr > 0, tr, torch.ones_like(tr))
    W = torch.empty(len(mats), m, device=dev, dtype=torch.float32)
    info = torch.empty(len(mats), device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.lgnn_symeig_batched(A.data_ptr(), m, len(mats), W.data_ptr(), info.data_ptr(), stream),
                   "lgnn_symeig_batched")
    bad = info.ne(0).tolist()  # the one synchronisation (torch.linalg.eigh checks its info the same way)
    out = []
    for b, H in enumerate(mats):
        n = H.shape[0]
        if bad[b]:
            out.append(symeig(H))
            continue
        lam = torch.nan_to_num(W[b, m - n:].clamp(min=0.0))
        Q = torch.nan_to_num(A[b, m - n:, :n].T.contiguous())
        out.append((lam, Q))
    return out


def symeig_batched_hip(mats: list, cache: dict | None = None, keys: list | None = None) -> list:
    """Eigendecompose several symmetric PSD factors on the GPU (``lgnn_symeig_batched``): per-factor ``torch.linalg.eigh`` is
    a serial chain of ~100 single-workgroup kernels (3.3 ms per 256 x 256 factor).  Factors of up to 256 rows go through ONE
    call (hand-written tridiagonalisation, all factors side by side: the four arxiv factors in 2 ms), larger ones one call
    each (padding a 64 x 64 factor to 1 433 would cost a second large decomposition).  ``cache`` / ``keys``: a dict owned by
    the caller and one hashable key (or None) per factor -- a large factor whose key is in the cache is served from it, a
    keyed one that is not gets stored (the ``_LARGE_EIG_KEEP`` most recent entries stay); factors without a key are always
    decomposed.

    A factor H (n x n) smaller than the largest of its call (m x m) is embedded as blockdiag(H, -mean_eig(H) I): the blocks never couple (Householder reflectors and the divide and
    conquer splits keep exact zeros), D's eigenpairs sort first, H's are the last n rows restricted to the
    first n columns.  Returns [(eigenvalues, eigenvectors)] with ``symeig``'s conventions (ascending, clamped
    at 0, NaNs zeroed, eigenvectors in columns); non-convergence falls back to ``symeig`` for that factor."""
    out = [None] * len(mats)
    small = [b for b, H in enumerate(mats) if H.shape[0] <= _SMALL_EIG]
    if small:
        for b, pair in zip(small, _symeig_group_hip([mats[b] for b in small])):
            out[b] = pair
    for b, H in enumerate(mats):
        if out[b] is not None:
            continue
        key = keys[b] if (cache is not None and keys is not None) else None
        if key is not None and key in cache:
            lam, Q = cache.pop(key)
            cache[key] = (lam, Q)  # most recently used last
            out[b] = (lam.clone(), Q.clone())
            continue
        out[b] = _symeig_group_hip([H])[0]
        if key is not None:
            cache[key] = (out[b][0].clone(), out[b][1].clone())
            while len(cache) > _LARGE_EIG_KEEP:
                cache.pop(next(iter(cache)))
    return out


def symeig(M: torch.Tensor):
    """eigh with the reference's safety net: on failure add/remove identity jitter; eigenvalues
    clamped at 0, NaNs zeroed (laplace/utils/utils.py:193-226)."""
    try:
        L, W = torch.linalg.eigh(M, UPLO="U")
    except RuntimeError:
        M = M + torch.eye(M.shape[0], device=M.device, dtype=M.dtype)
        L, W = torch.linalg.eigh(M, UPLO="U")
        L = L - 1.0
    L = torch.nan_to_num(L.clamp(min=0.0))
    W = torch.nan_to_num(W)
    return L, W


class Kron:
    def __init__(self, kfacs: list, tied=None):
        self._kfacs = kfacs
        # blocks i whose first factor is KNOWN to equal block i-1's (a bias block repeats the B factor of its weight block,
        # curvlinops.py:64-66): ``decompose`` then skips comparing the tensors, which would synchronise the stream while
        # the accumulation is still queued.  A hint only: blocks not listed are compared.  The hint describes the factors it
        # was given with: assigning ``kfacs`` (load_state_dict, callers that rebuild the list) drops it.
        self._tied = frozenset(tied or ())

    @property
    def kfacs(self) -> list:
        return self._kfacs

    @kfacs.setter
    def kfacs(self, value: list):
        self._kfacs = value
        self._tied = frozenset()

    @classmethod
    def init_from_model(cls, model: nn.Module | Iterable[nn.Parameter], device) -> "Kron":
        params = model.parameters() if isinstance(model, nn.Module) else model
        kfacs, tied, prev_out = [], [], None
        for p in params:
            if p.ndim == 1:
                if prev_out == p.size(0):
                    tied.append(len(kfacs))  # zeros equal zeros
                kfacs.append([torch.zeros(p.size(0), p.size(0), device=device)])
                prev_out = None
            elif p.ndim == 2:
                o, i = p.size()
                kfacs.append([torch.zeros(o, o, device=device), torch.zeros(i, i, device=device)])
                prev_out = o
            else:
                raise ValueError("Invalid parameter shape in network.")
        return cls(kfacs, tied=tied)

    def __add__(self, other: "Kron") -> "Kron":
        if not isinstance(other, Kron):
            raise ValueError("Can only add Kron to Kron.")
        return Kron([[Hi.add(Hj) for Hi, Hj in zip(Fi, Fj)] for Fi, Fj in zip(self.kfacs, other.kfacs)],
                    tied=getattr(self, "_tied", frozenset()) & getattr(other, "_tied", frozenset()))

    def __mul__(self, scalar) -> "Kron":
        if not isinstance(scalar, (int, float)) and not (torch.is_tensor(scalar) and scalar.numel() == 1):
            raise ValueError("Input not valid python or torch scalar.")
        scalar = float(scalar)
        # (a weight block's B takes sqrt(scalar), its bias block's B takes scalar: equal only for scalar == 1)
        return Kron([[pow(scalar, 1 / len(F)) * Hi for Hi in F] for F in self.kfacs],
                    tied=getattr(self, "_tied", frozenset()) if scalar == 1.0 else None)

    __radd__ = __add__
    __rmul__ = __mul__

    def __len__(self):
        return len(self.kfacs)

    def decompose(self, damping: bool = False, process_group=None, cache: dict | None = None,
                  cache_keys: dict | None = None) -> "KronDecomposed":
        """Eigendecompose every factor (laplace/utils/matrix.py:118-145).  Inside a ``torch.distributed``
        job the distinct factors are dealt to the ranks (largest first), each rank decomposes its share and
        ONE all-reduce of a zero-padded flat buffer hands every rank all eigenpairs: the replicated
        rocSOLVER time stops limiting the scaling of a fit, and all ranks hold bit-identical bases."""
        import torch.distributed as dist

        if process_group is not None or (dist.is_available() and dist.is_initialized()):
            world = dist.get_world_size(process_group)
            # GPU: the batched solver call costs one largest-factor decomposition, every rank just runs it on its
            # (bit-identical, all-reduced) factors; dealing the factors out only pays for per-factor solvers (CPU)
            if world > 1 and not self.kfacs[0][0].is_cuda:
                return self._decompose_distributed(damping, dist.get_rank(process_group), world, process_group)
        # distinct factors (a bias block repeats the B factor of its weight block, curvlinops.py:64-66: decompose once)
        distinct, where = self._distinct_factors()
        if distinct[0].is_cuda:
            # cache_keys {(block, factor): key}: exact identities of factors the caller knows (see symeig_batched_hip)
            keys = None
            if cache is not None and cache_keys:
                keys = [None] * len(distinct)
                for (i, k), key in cache_keys.items():
                    keys[where[i][k]] = key
            pairs = symeig_batched_hip(distinct, cache, keys)  # HIP path: one batched solver call, no CPU fallback
        else:
            pairs = [symeig(Hi) for Hi in distinct]
        eigvecs = [[pairs[j][1] for j in idxs] for idxs in where]
        eigvals = [[pairs[j][0] for j in idxs] for idxs in where]
        return KronDecomposed(eigvecs, eigvals, damping=damping)

    def _distinct_factors(self):
        distinct, where = [], []  # where[i][k] = index into `distinct`
        prev = None
        tied = getattr(self, "_tied", frozenset())
        for i, F in enumerate(self.kfacs):
            idxs = []
            for k, Hi in enumerate(F):
                if k == 0 and prev is not None and prev[0].shape == Hi.shape and (i in tied or torch.equal(prev[0], Hi)):
                    idxs.append(prev[1])
                else:
                    distinct.append(Hi)
                    idxs.append(len(distinct) - 1)
                if k == 0:
                    prev = (Hi, idxs[-1])
            where.append(idxs)
        return distinct, where

    def _decompose_distributed(self, damping, rank, world, group) -> "KronDecomposed":
        import torch.distributed as dist

        distinct, where = self._distinct_factors()
        order = sorted(range(len(distinct)), key=lambda i: -distinct[i].shape[0])
        owner = {i: pos % world for pos, i in enumerate(order)}
        sizes = [d.shape[0] for d in distinct]
        flat = torch.zeros(sum(n * n + n for n in sizes), dtype=distinct[0].dtype, device=distinct[0].device)
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n * n + n
        for i, Hi in enumerate(distinct):
            if owner[i] == rank:
                lam, Q = symeig(Hi)
                n = sizes[i]
                flat[offs[i]:offs[i] + n * n] = Q.reshape(-1)
                flat[offs[i] + n * n:offs[i] + n * n + n] = lam
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        pairs = []
        for i, n in enumerate(sizes):
            Q = flat[offs[i]:offs[i] + n * n].view(n, n).clone()
            lam = flat[offs[i] + n * n:offs[i] + n * n + n].clone()
            pairs.append((lam, Q))
        eigvecs = [[pairs[j][1] for j in idxs] for idxs in where]
        eigvals = [[pairs[j][0] for j in idxs] for idxs in where]
        return KronDecomposed(eigvecs, eigvals, damping=damping)

    def diag(self) -> torch.Tensor:
        out = []
        for F in self.kfacs:
            F0 = F[0].diag()
            out.append(F0 if len(F) == 1 else torch.outer(F0, F[1].diag()).flatten())
        return torch.cat(out)

    def to_matrix(self) -> torch.Tensor:
        blocks = [F[0] if len(F) == 1 else torch.kron(F[0], F[1]) for F in self.kfacs]
        return torch.block_diag(*blocks)

    def logdet(self) -> torch.Tensor:
        ld = 0
        for F in self.kfacs:
            if len(F) == 1:
                ld = ld + F[0].logdet()
            else:
                ld = ld + len(F[1]) * F[0].logdet() + len(F[0]) * F[1].logdet()
        return ld


class KronDecomposed:
    """Eigendecomposed factors plus prior precision ``deltas`` (laplace/utils/matrix.py:277-394)."""

    def __init__(self, eigenvectors, eigenvalues, deltas: torch.Tensor | None = None, damping: bool = False):
        self.eigenvectors, self.eigenvalues = eigenvectors, eigenvalues
        dev = eigenvectors[0][0].device
        self.deltas = torch.zeros(len(self), device=dev) if deltas is None else deltas
        self.damping = damping

    def __len__(self):
        return len(self.eigenvalues)

    def __add__(self, deltas: torch.Tensor) -> "KronDecomposed":
        deltas = torch.as_tensor(deltas, device=self.deltas.device, dtype=self.deltas.dtype).reshape(-1)
        if deltas.numel() == 1:
            deltas = deltas.expand(len(self))
        if deltas.numel() != len(self):
            raise ValueError("Invalid shape of delta added.")
        return KronDecomposed(self.eigenvectors, self.eigenvalues, self.deltas + deltas, self.damping)

    def __mul__(self, scalar) -> "KronDecomposed":
        scalar = float(scalar)
        ev = [[pow(scalar, 1 / len(ls)) * l for l in ls] for ls in self.eigenvalues]
        return KronDecomposed(self.eigenvectors, ev, self.deltas, self.damping)

    def logdet(self) -> torch.Tensor:
        """sum_blocks sum log(l1 (x) l2 + delta)  (matrix.py:371-394)."""
        ld = 0
        for ls, delta in zip(self.eigenvalues, self.deltas):
            if len(ls) == 1:
                ld = ld + torch.log(ls[0] + delta).sum()
            else:
                l1, l2 = ls
                if self.damping:
                    ds = torch.sqrt(delta)
                    ld = ld + torch.log(torch.outer(l1 + ds, l2 + ds)).sum()
                else:
                    ld = ld + torch.log(torch.outer(l1, l2) + delta).sum()
        return ld

    def bmm(self, W: torch.Tensor, exponent: float = -1) -> torch.Tensor:
        """``self ** exponent @ W`` for ``W`` of shape (batch, params) or (batch, classes, params)
        (laplace/utils/matrix.py:396-483): per block ``Q (l + delta)^e Q^T w`` resp.
        ``Q1 ((Q1^T W Q2) * (l1 (x) l2 + delta)^e) Q2^T`` -- invariant to the eigenvector gauge."""
        if W.ndim == 1:
            return self._bmm(W.unsqueeze(0).unsqueeze(0), exponent).squeeze()
        if W.ndim == 2:
            return self._bmm(W.unsqueeze(1), exponent).squeeze(1)
        if W.ndim == 3:
            return self._bmm(W, exponent)
        raise ValueError("Invalid shape for W")

    def inv_square_form(self, W: torch.Tensor) -> torch.Tensor:
        """``W P^-1 W^T`` per batch element for W [batch, K, params] (laplace/utils/matrix.py:448-451)."""
        SW = self._bmm(W, -1)
        return torch.bmm(W, SW.transpose(1, 2))

    def _bmm(self, W: torch.Tensor, exponent: float) -> torch.Tensor:
        B, K, P = W.size()
        W = W.reshape(B * K, P)
        cur, out = 0, []
        for ls, Qs, delta in zip(self.eigenvalues, self.eigenvectors, self.deltas):
            if len(ls) == 1:
                Q, lam = Qs[0], ls[0]
                p = len(lam)
                scale = torch.pow(lam + delta, exponent).reshape(-1, 1)
                Wp = W[:, cur:cur + p].T
                out.append((Q @ (scale * (Q.T @ Wp))).T)
            else:
                (Q1, Q2), (l1, l2) = Qs, ls
                p_in, p_out = len(l1), len(l2)
                p = p_in * p_out
                if self.damping:
                    ds = torch.sqrt(delta)
                    scale = torch.pow(torch.outer(l1 + ds, l2 + ds), exponent).unsqueeze(0)
                else:
                    scale = torch.pow(torch.outer(l1, l2) + delta, exponent).unsqueeze(0)
                Wp = W[:, cur:cur + p].reshape(B * K, p_in, p_out)
                Wp = Q1 @ ((Q1.T @ Wp @ Q2) * scale) @ Q2.T
                out.append(Wp.reshape(B * K, p))
            cur += p
        return torch.cat(out, dim=1).reshape(B, K, P)

    def to_matrix(self, exponent: float = 1) -> torch.Tensor:
        blocks = []
        for Qs, ls, delta in zip(self.eigenvectors, self.eigenvalues, self.deltas):
            if len(ls) == 1:
                Q, l = Qs[0], ls[0]
                blocks.append(Q @ torch.diag(torch.pow(l + delta, exponent)) @ Q.T)
            else:
                Q = torch.kron(Qs[0], Qs[1])
                l = torch.pow(torch.outer(ls[0], ls[1]) + delta, exponent).flatten()
                blocks.append(Q @ torch.diag(l) @ Q.T)
        return torch.block_diag(*blocks)
