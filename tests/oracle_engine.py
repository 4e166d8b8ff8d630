"""TEST-ONLY stand-in for ``laplace_gnn_amd.engine.GraphEngine``: the same Python surface, CPU oracle arithmetic.

It lets the REAL backend class (``laplace_gnn_amd.HipGGN``, unmodified) run without a GPU, so that its contract with a
``Laplace`` front -- constructor kwargs, return types and conventions, fresh tensors -- can be exercised in the build
container, including through the reference's own ``laplace.baselaplace`` classes (tests/test_reference_dropin.py).
Never shipped, never the product path, never timed.
"""
import numpy as np
import torch

import gnn_laplace_oracle as O


class OracleEngine:
    def __init__(self, model):
        """``model``: a laplace_gnn_amd.GCN / GraphSAGE living on the CPU."""
        self.m = model
        self.kind = model.kind
        self.device = torch.device("cpu")
        self.num_nodes = model.num_nodes
        mult = 2 if self.kind == "sage" else 1
        self.dims = [model.convs[0].lin.weight.shape[1] // mult] + [c.lin.weight.shape[0] for c in model.convs]
        self.in_dims = [mult * d for d in self.dims[:-1]]
        L = len(self.dims) - 1
        self.has_res = len(getattr(model, "res", [])) > 0
        self.norm = getattr(model, "norm_kind", None)
        self.block_dims = list(zip(self.in_dims, self.dims[1:])) + (
            [(self.dims[l], self.dims[l + 1]) for l in range(L - 1)] if self.has_res else [])
        self.likelihood = "classification"
        self.calls = []

    @property
    def has_extras(self):
        return self.has_res or self.norm is not None

    @property
    def n_params(self):
        return sum(i * o + o for i, o in self.block_dims)

    @property
    def num_layers(self):
        return len(self.dims) - 1

    def set_likelihood(self, likelihood):
        self.likelihood = likelihood

    def _om(self):
        from oracle_backend import oracle_model_of
        return oracle_model_of(self.m)

    def forward(self, idx):
        out, _, _ = O.forward_all(self._om())
        return torch.from_numpy(out[idx.numpy()])

    def new_kfac_buffers(self):
        sizes = []
        for i, o in self.block_dims:
            sizes += [i * i, o * o]
        flat = torch.zeros(sum(sizes) + 1)
        views, off = [], 0
        for i, o in self.block_dims:
            A = flat[off:off + i * i].view(i, i); off += i * i
            B = flat[off:off + o * o].view(o, o); off += o * o
            views.append((A, B))
        return flat, views, flat[off:off + 1]

    def kfac_accumulate(self, idx, y, n_train, views, loss, fork_exact=True, fuse=True, classes=None):
        """RAW factors (A_l / n_train, B_l) and the raw loss sum, as lgnn_kfac_accumulate adds them."""
        self.calls.append(("kfac", tuple(idx.tolist()), classes))
        reg = self.likelihood == "regression"
        l, kfacs = O.kfac_batch(self._om(), idx.numpy(), y.numpy(), n_train, fork_exact, classes, self.likelihood)
        undo2, undo1 = (1.0 / np.sqrt(0.5), 2.0) if reg else (1.0, 1.0)  # the oracle applied the interface factor 0.5
        for k, (A, B) in enumerate(views):
            B += torch.from_numpy(np.ascontiguousarray(kfacs[2 * k][0])) * undo2
            A += torch.from_numpy(np.ascontiguousarray(kfacs[2 * k][1])) * undo2
        loss += float(l) * undo1

    def diag_accumulate(self, idx, y, diag, loss):
        self.calls.append(("diag", tuple(idx.tolist())))
        l, H = O.diag_batch(self._om(), idx.numpy(), y.numpy(), self.likelihood)
        diag += torch.from_numpy(H)
        loss += float(l) * (2.0 if self.likelihood == "regression" else 1.0)

    def full_accumulate(self, idx, y, H, loss):
        self.calls.append(("full", tuple(idx.tolist())))
        l, Hb = O.full_batch(self._om(), idx.numpy(), y.numpy(), self.likelihood)
        H += torch.from_numpy(Hb)
        loss += float(l) * (2.0 if self.likelihood == "regression" else 1.0)

    def kfac_accumulate_fisher(self, idx, y_seed, y_loss, n_train, views, loss, resid_scale=1.0, b_scale=1.0, fuse=True):
        self.calls.append(("kfac_fisher", tuple(idx.tolist())))
        om = self._om()
        out, _, _ = O.forward_all(om)
        r = O.fisher_residual(out[idx.numpy()], y_seed.numpy(), self.likelihood, resid_scale)
        first = y_loss is not None
        l, kfacs = O.kfac_batch(om, idx.numpy(), (y_loss if first else y_seed).numpy(), n_train,
                                likelihood="classification", seed_override=r[:, :, None] * np.sqrt(b_scale))
        for k, (A, B) in enumerate(views):
            B += torch.from_numpy(np.ascontiguousarray(kfacs[2 * k][0]))
            if first:
                A += torch.from_numpy(np.ascontiguousarray(kfacs[2 * k][1]))
        if first:
            f = out[idx.numpy()]
            if self.likelihood == "regression":
                d = f - y_loss.numpy().reshape(f.shape)
                loss += float((d * d).sum())
            else:
                loss += float(O.ce_sum(f, y_loss.numpy()))

    def ef_accumulate(self, idx, y_seed, y_loss=None, resid_scale=1.0, scale=1.0, diag=None, full=None, grads=False,
                      loss=None):
        self.calls.append(("ef", tuple(idx.tolist())))
        Gs, _ = O.ef_gradients(self._om(), idx.numpy(), y_seed.numpy(), self.likelihood, resid_scale)
        G = torch.from_numpy(Gs)
        if diag is not None:
            diag += scale * (G * G).sum(0)
        if full is not None:
            full += scale * G.T @ G
        if y_loss is not None and loss is not None:
            _, l = O.ef_gradients(self._om(), idx.numpy(), y_loss.numpy(), self.likelihood, resid_scale)
            loss += float(l)
        return G if grads else None

    def lastlayer_full_accumulate(self, idx, y, H, loss):
        self.calls.append(("ll_full", tuple(idx.tolist())))
        l, Hb = O.lastlayer_full_batch(self._om(), idx.numpy(), y.numpy())
        H += torch.from_numpy(Hb)
        loss += float(l)

    def jacobians(self, idx):
        self.calls.append(("jacobians", tuple(idx.tolist())))
        Js, f = O.jacobians_batch(self._om(), idx.numpy())
        return torch.from_numpy(Js), torch.from_numpy(f)

    def check_async_errors(self):
        pass

    def invalidate(self):
        pass
