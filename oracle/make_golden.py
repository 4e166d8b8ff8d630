"""TEST INFRASTRUCTURE ONLY -- generates ``tests/golden/*.npz`` by running the
REFERENCE's own Python (``/root/reference`` via ``oracle/ref_loader.py``).

Run in the build container only (the reference does not travel):

    cd /tmp && python3 -B /root/repo/oracle/make_golden.py

Each fixture holds inputs (``edge_index`` int64 with duplicates / self-loops as drawn,
``X``, weights, ``train_idx``, ``train_y``, ``batch_size``) and the reference's outputs
(``logits`` for all nodes, the dense propagation matrix as COO, ``adj_to_edge_index``
order, ``kfacs`` blocks of ``KronLaplace.fit``, ``diag`` H of ``DiagLaplace.fit``, full GGN
of ``FullLaplace``-style ``GGNInterface.full`` on small cases, losses).  Fixtures are data,
not code.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

OUT = os.environ.get("LGNN_GOLDEN_OUT", os.path.join(os.path.dirname(HERE), "tests", "golden"))  # (a scratch directory to compare a regeneration with the committed fixtures)


def reference_dense_adj(torch, edge_index, n):
    """gnn/utils.py:325-330 + gnn/marglik_training.py:404-405.  ``to_scipy_sparse_matrix``
    of torch_geometric (absent here) is ``scipy.sparse.coo_matrix((ones,(row,col)),(N,N))``;
    ``.toarray()`` sums duplicates; the driver then clamps to 1."""
    import scipy.sparse as sp

    row, col = edge_index[0].numpy(), edge_index[1].numpy()
    dense = sp.coo_matrix((np.ones(len(row)), (row, col)), (n, n)).toarray()
    adj = torch.tensor(dense, dtype=torch.int64).float()
    adj[adj > 1] = 1
    return adj


def reference_adj_to_edge_index(adj):
    """gnn/utils.py:333-336 (the file itself imports torch_geometric/sklearn/GPUtil at
    module level, so the three-line function is executed on the reference's tensor ops)."""
    _adj = adj.clone()
    _adj.fill_diagonal_(0)
    return _adj.nonzero().t().contiguous()


def make_case(ns, torch, name, kind, n, f, h, c, layers, n_edges, n_train, batch_size, seed,
              symmetric=False, isolated=0, with_full=False, norm=None, res=False, full_only=False):
    from torch.utils.data import DataLoader, TensorDataset

    g = torch.Generator().manual_seed(seed)
    hi = n - isolated  # the last `isolated` nodes get no edges at all
    ei = torch.randint(0, hi, (2, n_edges), generator=g)
    # force duplicates and explicit self loops into the raw edge list
    ei = torch.cat([ei, ei[:, : max(1, n_edges // 10)], torch.arange(0, hi, 7).repeat(2, 1)], 1)
    X = torch.randn(n, f, generator=g)
    adj0 = reference_dense_adj(torch, ei, n)

    torch.manual_seed(seed)
    extras = dict(norm=norm, res=res) if (norm is not None or res) else {}  # BaseGNN kwargs (base_gnn.py:22-23)
    if kind == "gcn":
        model = ns.gnn_models.GCN(f, h, c, layers, X, adj0.clone(), dropout_p=0.5,
                                  symmetric=symmetric, **extras)
    else:
        model = ns.gnn_models.GraphSAGE(f, h, c, layers, X, adj0.clone(),
                                        num_sampled_nodes_per_hop=None, dropout_p=0.5,
                                        symmetric=symmetric, **extras)
    if norm is not None:
        # non-trivial norm state (a fresh LayerNorm / BatchNorm1d is gamma = 1, beta = 0, running stats 0 / 1): the harness
        # sets it, the reference computes with it
        gn = torch.Generator().manual_seed(seed + 1000)
        with torch.no_grad():
            for nm in model.norms:
                nm.weight.copy_(0.5 + torch.rand(h, generator=gn))
                nm.bias.copy_(0.3 * torch.randn(h, generator=gn))
                if norm == "batch":
                    nm.running_mean.copy_(0.2 * torch.randn(h, generator=gn))
                    nm.running_var.copy_(0.5 + torch.rand(h, generator=gn))
    model.eval()
    perm = torch.randperm(n, generator=g)
    train_idx = perm[:n_train].clone()
    if n_train >= 8:
        train_idx[3] = train_idx[5]  # a duplicated node id inside one batch
    train_y = torch.randint(0, c, (n_train,), generator=g)
    loader = DataLoader(TensorDataset(train_idx, train_y), batch_size=batch_size, shuffle=False)

    out = {
        "kind": kind, "symmetric": symmetric, "num_nodes": n, "batch_size": batch_size,
        "edge_index": ei.numpy(), "X": X.numpy(), "train_idx": train_idx.numpy(),
        "train_y": train_y.numpy(), "num_layers": layers,
    }
    for l, conv in enumerate(model.convs):
        out[f"W{l}"] = conv.lin.weight.detach().numpy().copy()
        out[f"b{l}"] = conv.lin.bias.detach().numpy().copy()
    if extras:
        out["norm"], out["res"] = str(norm), bool(res)
        out["param_names"] = np.array([k for k, v in model.named_parameters()
                                       if v.requires_grad and "adj" not in k and "norms" not in k])
        for l, lin in enumerate(model.res):
            out[f"Wr{l}"], out[f"br{l}"] = lin.weight.detach().numpy().copy(), lin.bias.detach().numpy().copy()
        if norm is not None:
            out["norm_eps"] = np.float64(model.norms[0].eps)
            for l, nm in enumerate(model.norms):
                out[f"norm_w{l}"], out[f"norm_b{l}"] = nm.weight.detach().numpy().copy(), nm.bias.detach().numpy().copy()
                if norm == "batch":
                    out[f"norm_rm{l}"], out[f"norm_rv{l}"] = nm.running_mean.numpy().copy(), nm.running_var.numpy().copy()

    with torch.no_grad():
        adj = model.adj.detach()
        out["adj_edge_index"] = reference_adj_to_edge_index(adj).numpy()
        nzr, nzc = adj.nonzero(as_tuple=True)  # row-major order of the stored 0/1 matrix
        out["adj_nz_row"], out["adj_nz_col"] = nzr.numpy(), nzc.numpy()
        if kind == "gcn":
            P = model.forward_adj()
        else:  # what GraphSAGEConv.mean_agg multiplies with (layers.py:18-24)
            rs = adj.sum(dim=1, keepdims=True)
            rs[rs == 0] = 1
            P = adj / rs
        pr, pc = P.nonzero(as_tuple=True)
        out["prop_row"], out["prop_col"], out["prop_val"] = pr.numpy(), pc.numpy(), P[pr, pc].numpy()
        out["logits"] = model(torch.arange(n)).numpy()

    bl = ns.baselaplace
    la = bl.KronLaplace(model, "classification")
    la.fit(loader)
    out["kron_loss"] = np.float32(float(la.loss))
    out["kron_n_blocks"] = len(la.H_facs.kfacs)
    for i, Fs in enumerate(la.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            out[f"kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
    # eigen-decomposition that KronLaplace.fit ends with (matrix.py:118-145): eigenvalues only
    for i, ls in enumerate(la.H.eigenvalues):
        for j, lam in enumerate(ls):
            out[f"kron_eig_{i}_{j}"] = lam.detach().numpy().astype(np.float32)

    # "next" row 8(f)-1: log marginal likelihood from the decomposed factors (baselaplace.py:938-973,
    # matrix.py:371-394), scalar and per-parameter-group prior precision
    out["kron_marglik_pp1"] = np.float64(float(la.log_marginal_likelihood()))
    out["kron_marglik_pp07"] = np.float64(float(la.log_marginal_likelihood(prior_precision=torch.tensor(0.7))))
    pp_layer = torch.tensor([0.5, 2.0, 1.5, 0.25] * (len(la.H_facs.kfacs) // 4) + [1.0] * (len(la.H_facs.kfacs) % 4))
    out["kron_prior_layerwise"] = pp_layer.numpy()
    out["kron_marglik_layerwise"] = np.float64(float(la.log_marginal_likelihood(prior_precision=pp_layer)))

    # posterior samples + the sampling predictive the GNN driver evaluates with (gnn/marglik_training.py:338-352):
    # fixed standard-normal draws `eps` replace torch.randn so that the result is reproducible on any device
    la.prior_precision = 1.0
    S = 4
    eps = torch.from_numpy(np.random.default_rng(seed).standard_normal((S, la.n_params)).astype(np.float32))
    eval_idx = train_idx[:10].clone()
    out["pred_eps"], out["pred_idx"] = eps.numpy(), eval_idx.numpy()
    k_samples = la.mean.reshape(1, -1) + la.posterior_precision.bmm(eps, exponent=-0.5).reshape(S, -1)
    out["kron_samples"] = k_samples.detach().numpy().astype(np.float32)
    la.sample = lambda n_samples=S, generator=None: k_samples.detach()
    out["kron_nn_py"] = la._nn_predictive_classification(eval_idx, n_samples=S).detach().numpy().astype(np.float32)
    # GLM predictive (the reference's default la(x)): explicit Jacobians, f_var = J P^-1 J^T, link approximations
    # (laplace/baselaplace.py:570-665, 1123-1158).  glm_eps: fixed standard-normal draws for link_approx="mc".
    import laplace.utils.utils as _lu
    glm_eps = torch.from_numpy(np.random.default_rng(seed + 100).standard_normal((int(out["n_outputs"]) if "n_outputs" in out
                               else model.output_size, 6)).astype(np.float32))
    out["glm_eps"] = glm_eps.numpy()

    def _glm_all(L_, key):
        f_mu, f_var = L_._glm_predictive_distribution(eval_idx)
        out[key + "_glm_fmu"] = f_mu.detach().numpy().astype(np.float32)
        out[key + "_glm_fvar"] = f_var.detach().numpy().astype(np.float32)
        for link in ("probit", "bridge", "bridge_norm"):
            out[f"{key}_glm_{link}"] = L_(eval_idx, pred_type="glm", link_approx=link).detach().numpy().astype(np.float32)
        real_randn = torch.randn
        try:  # the reference's normal_samples draws torch.randn((C, n_samples)): hand it the fixed draws
            _lu.torch.randn = lambda *a, **k: glm_eps.clone()
            out[key + "_glm_mc"] = L_(eval_idx, pred_type="glm", link_approx="mc", n_samples=6).detach().numpy().astype(np.float32)
        finally:
            _lu.torch.randn = real_randn
    _glm_all(la, "kron")
    # post-hoc prior precision tuning on the marginal likelihood (laplace/baselaplace.py:444-463): 25 Adam steps
    import warnings as _w
    # (the fork's factors and loss carry the autograd graph of the fit (kfac.py:655-661 create_graph=True), which a
    #  second backward cannot traverse: the tuning loop only works on detached copies of the fitted state)
    la.loss = la.loss.detach() if torch.is_tensor(la.loss) else la.loss
    la.H.eigenvalues = [[l_.detach() for l_ in ls] for ls in la.H.eigenvalues]
    la.H.eigenvectors = [[q_.detach() for q_ in qs] for qs in la.H.eigenvectors]
    with _w.catch_warnings():
        _w.simplefilter("ignore")
        la.optimize_prior_precision(pred_type="glm", method="marglik", n_steps=25, lr=0.1, prior_structure="scalar")
        out["kron_opt_pp_scalar"] = la.prior_precision.detach().numpy().astype(np.float32)
        la.optimize_prior_precision(pred_type="glm", method="marglik", n_steps=25, lr=0.1, prior_structure="layerwise")
        out["kron_opt_pp_layerwise"] = la.prior_precision.detach().numpy().astype(np.float32)
    la.prior_precision = 1.0

    ld = bl.DiagLaplace(model, "classification")
    ld.fit(loader)
    d_samples = ld.mean.reshape(1, -1) + eps * ld.posterior_scale.reshape(1, -1)
    out["diag_samples"] = d_samples.detach().numpy().astype(np.float32)
    ld.sample = lambda n_samples=S, generator=None: d_samples.detach()
    out["diag_nn_py"] = ld._nn_predictive_classification(eval_idx, n_samples=S).detach().numpy().astype(np.float32)
    _glm_all(ld, "diag")
    out["diag_marglik_pp1"] = np.float64(float(ld.log_marginal_likelihood()))
    out["diag_marglik_pp07"] = np.float64(float(ld.log_marginal_likelihood(prior_precision=torch.tensor(0.7))))
    out["diag_loss"] = np.float32(float(ld.loss))
    out["diag_H"] = ld.H.detach().numpy().astype(np.float32)
    out["n_data"], out["n_outputs"], out["n_params"] = ld.n_data, ld.n_outputs, ld.n_params

    if with_full and not full_only:
        # regression likelihood on the same model (MSELoss, Hessian sqrt = sqrt(2) I, factor 0.5): float targets
        reg_y = torch.randn(n_train, c, generator=g)
        out["reg_y"] = reg_y.numpy()
        rloader = DataLoader(TensorDataset(train_idx, reg_y), batch_size=batch_size, shuffle=False)
        lr_ = bl.KronLaplace(model, "regression", sigma_noise=0.7)
        lr_.fit(rloader)
        out["reg_kron_loss"] = np.float32(float(lr_.loss))
        for i, Fs in enumerate(lr_.H_facs.kfacs):
            for j, Hm in enumerate(Fs):
                out[f"reg_kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
        out["reg_kron_marglik"] = np.float32(float(lr_.log_marginal_likelihood()))
        f_mu_r, f_var_r = lr_(eval_idx, pred_type="glm")
        out["reg_kron_glm_fmu"] = f_mu_r.detach().numpy().astype(np.float32)
        out["reg_kron_glm_fvar"] = f_var_r.detach().numpy().astype(np.float32)
        ldr = bl.DiagLaplace(model, "regression", sigma_noise=0.7)
        ldr.fit(rloader)
        out["reg_diag_H"] = ldr.H.detach().numpy().astype(np.float32)
        out["reg_diag_loss"] = np.float32(float(ldr.loss))
        out["reg_diag_marglik"] = np.float32(float(ldr.log_marginal_likelihood()))

    if with_full:  # backend-free pin for "full" (SURVEY.md 8(c)): GGNInterface.full einsum
        be = ns.curvature.GGNInterface(model, "classification")
        loss, H = 0.0, 0.0
        for xb, yb in loader:
            lb, Hb = be.full(xb, yb)
            loss, H = loss + lb, H + Hb
        out["full_H"] = H.detach().numpy().astype(np.float32)
        out["full_loss"] = np.float32(float(loss))
        # upstream-KFAC seeds (detached sqrt) are not what the fork computes; also pin the
        # per-batch Jacobians of the first batch for the oracle's generic backward
        xb, _ = next(iter(loader))
        Js, fb = be.jacobians(xb, enable_backprop=False)
        out["jac_first_batch"] = Js.numpy().astype(np.float32)
        out["f_first_batch"] = fb.numpy().astype(np.float32)
        # FullLaplace over all weights with that backend: marginal likelihood, posterior samples, GLM predictive
        lf = bl.FullLaplace(model, "classification", backend=ns.curvature.GGNInterface)
        lf.fit(loader)
        out["fullla_H"] = lf.H.detach().numpy().astype(np.float32)
        out["fullla_marglik_pp1"] = np.float32(float(lf.log_marginal_likelihood()))
        out["fullla_marglik_pp07"] = np.float32(float(lf.log_marginal_likelihood(prior_precision=torch.tensor(0.7))))
        lf.prior_precision = 1.0
        out["fullla_samples"] = (lf.mean.reshape(1, -1) + eps @ lf.posterior_scale).detach().numpy().astype(np.float32)
        f_mu, f_var = lf._glm_predictive_distribution(eval_idx)
        out["fullla_glm_fvar"] = f_var.detach().numpy().astype(np.float32)
        out["fullla_glm_probit"] = lf(eval_idx, pred_type="glm", link_approx="probit").detach().numpy().astype(np.float32)

    if with_full and not full_only:
        # empirical Fisher (EFInterface / CurvlinopsEF) and Monte-Carlo Fisher (stochastic=True) of the same model; the
        # labels the reference draws (curvlinops/kfac.py:698-745, curvature.py:343-364) are recorded: the device path
        # takes them as an input, a random stream cannot be reproduced across devices
        cv, ccl = ns.curvature, ns.curvature_curvlinops
        ef = cv.EFInterface(model, "classification")
        Gs, l0 = ef.gradients(train_idx, train_y)
        out["ef_grads"], out["ef_loss"] = Gs.detach().numpy().astype(np.float32), np.float32(float(l0))
        out["ef_diag"] = ef.diag(train_idx, train_y)[1].detach().numpy().astype(np.float32)
        out["ef_full"] = ef.full(train_idx, train_y)[1].detach().numpy().astype(np.float32)
        le = bl.KronLaplace(model, "classification", backend=ccl.CurvlinopsEF)
        le.fit(loader)
        out["ef_kron_loss"] = np.float32(float(le.loss))
        for i, Fs in enumerate(le.H_facs.kfacs):
            for j, Hm in enumerate(Fs):
                out[f"ef_kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
        lde = bl.DiagLaplace(model, "classification", backend=ccl.CurvlinopsEF)
        lde.fit(loader)
        out["ef_diag_fit"] = lde.H.detach().numpy().astype(np.float32)
        # regression EF (MSELoss gradient 2 (f - y), factor 0.5)
        efr = cv.EFInterface(model, "regression")
        out["reg_ef_diag"] = efr.diag(train_idx, reg_y)[1].detach().numpy().astype(np.float32)
        ler = bl.KronLaplace(model, "regression", backend=ccl.CurvlinopsEF)
        ler.fit(rloader)
        for i, Fs in enumerate(ler.H_facs.kfacs):
            for j, Hm in enumerate(Fs):
                out[f"reg_ef_kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
        # MC Fisher KFAC, 3 samples per batch
        drawn = []
        orig_draw = ns.kfac.KFACLinearOperator.draw_label

        def recording_draw(self, output):
            r = orig_draw(self, output)
            drawn.append(r.clone())
            return r
        ns.kfac.KFACLinearOperator.draw_label = recording_draw
        try:
            lm = bl.KronLaplace(model, "classification", backend=ccl.CurvlinopsGGN, backend_kwargs=dict(stochastic=True),
                                asdl_fisher_kwargs=dict(mc_samples=3))
            lm.fit(loader)
        finally:
            ns.kfac.KFACLinearOperator.draw_label = orig_draw
        out["mc_samples"] = 3
        out["mc_kron_loss"] = np.float32(float(lm.loss))
        for i, Fs in enumerate(lm.H_facs.kfacs):
            for j, Hm in enumerate(Fs):
                out[f"mc_kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
        for q, lab in enumerate(drawn):  # batch t = q // 3, sample s = q % 3
            out[f"mc_labels_{q // 3}_{q % 3}"] = lab.numpy()
        # MC functional Fisher in the Jacobian route (GGNInterface, stochastic=True, 2 samples): one-hot draws recorded
        hot = []
        Mn = torch.distributions.Multinomial
        orig_sample = Mn.sample

        def recording_sample(self, *a, **k):
            r = orig_sample(self, *a, **k)
            hot.append(r.clone())
            return r
        Mn.sample = recording_sample
        try:
            gmc = cv.GGNInterface(model, "classification", stochastic=True, num_samples=2)
            out["ggnmc_diag"] = gmc.diag(train_idx, train_y)[1].detach().numpy().astype(np.float32)
            first = [h_.argmax(-1).numpy() for h_ in hot]
            hot.clear()
            out["ggnmc_full"] = gmc.full(train_idx, train_y)[1].detach().numpy().astype(np.float32)
            second = [h_.argmax(-1).numpy() for h_ in hot]
        finally:
            Mn.sample = orig_sample
        out["ggnmc_labels_diag"], out["ggnmc_labels_full"] = np.stack(first), np.stack(second)

    if layers == 2 and (not extras or kind == "gcn"):
        # (res / norm: GCN only -- the shipped STE-GCN configurations of Cornell / Texas / Wisconsin / Circle,
        #  gnn/configs/original/stegcn_config.yaml:54-105, 129-145; the STE model gets the same norm state as `model`)
        def same_extras(ste_):
            if not extras:
                return
            with torch.no_grad():
                for a, b in zip(ste_.norms, model.norms):
                    for pa, pb in zip(list(a.parameters()) + list(a.buffers()), list(b.parameters()) + list(b.buffers())):
                        pa.copy_(pb)
            if res:
                for a, b in zip(ste_.res, model.res):
                    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
        # 8(f)-4: what the GNN driver differentiates (gnn/marglik_training.py:197-216): -log marglik of a KronLaplace fit
        # w.r.t. the dense adjacency parameter of the STE model (gnn/models/models.py:65-118), same weights (same seed,
        # same construction order), prior precision 0.7.  Stored: the gradient on the stored entries of the 0/1
        # adjacency (row-major = adj_nz_row / adj_nz_col order) and on 200 random non-edges.
        # GraphSAGE: STEGraphSAGE (gnn/models/models.py:121-183), mean aggregation over the binarised matrix itself.
        torch.manual_seed(seed)
        if kind == "gcn":
            ste = ns.gnn_models.STEGCN(f, h, c, layers, X, adj0.clone(), dropout_p=0.5, threshold=0.5, symmetric=symmetric,
                                       **extras)
            same_extras(ste)
        else:
            ste = ns.gnn_models.STEGraphSAGE(f, h, c, layers, X, adj0.clone(), num_sampled_nodes_per_hop=None,
                                             dropout_p=0.5, threshold=0.5, symmetric=symmetric)
        ste.eval()
        for a, b in zip(ste.convs, model.convs):
            assert torch.equal(a.lin.weight, b.lin.weight) and torch.equal(a.lin.bias, b.lin.bias)
        ls = bl.KronLaplace(ste, "classification", prior_precision=0.7)
        ls.fit(loader)
        neg = -ls.log_marginal_likelihood()
        neg.backward()
        gr = ste.adj.grad.detach().numpy()
        out["adjgrad_prior"] = np.float64(0.7)
        out["adjgrad_neg_marglik"] = np.float64(float(neg))
        out["adjgrad_vals"] = gr[out["adj_nz_row"], out["adj_nz_col"]].astype(np.float32)
        dense01 = np.zeros((n, n), dtype=bool)
        dense01[out["adj_nz_row"], out["adj_nz_col"]] = True
        ner, nec = np.nonzero(~dense01)
        pick = np.random.default_rng(seed + 7).choice(len(ner), size=min(200, len(ner)), replace=False)
        out["adjgrad_ne_row"], out["adjgrad_ne_col"] = ner[pick], nec[pick]
        out["adjgrad_ne_val"] = gr[ner[pick], nec[pick]].astype(np.float32)
        if kind == "sage" and not extras:
            # hessian_structure="diag" with STEGraphSAGE (the driver offers the pair, gnn/utils.py:55-59, 81; no shipped config)
            torch.manual_seed(seed)
            ste_d = ns.gnn_models.STEGraphSAGE(f, h, c, layers, X, adj0.clone(), num_sampled_nodes_per_hop=None, dropout_p=0.5,
                                               threshold=0.5, symmetric=symmetric)
            ste_d.eval()
            for a, b in zip(ste_d.convs, model.convs):
                assert torch.equal(a.lin.weight, b.lin.weight) and torch.equal(a.lin.bias, b.lin.bias)
            ld_ = bl.DiagLaplace(ste_d, "classification", prior_precision=0.7)
            ld_.fit(loader)
            negd = -ld_.log_marginal_likelihood()
            negd.backward()
            grd = ste_d.adj.grad.detach().numpy()
            out["adjgrad_diag_neg_marglik"] = np.float64(float(negd))
            out["adjgrad_diag_vals"] = grd[out["adj_nz_row"], out["adj_nz_col"]].astype(np.float32)
            out["adjgrad_diag_ne_val"] = grd[ner[pick], nec[pick]].astype(np.float32)
        if kind == "gcn":
            # The same with hessian_structure="diag" -- what the shipped STE-GCN config runs (gnn/configs/original/
            # stegcn_config.yaml:7): DiagLaplace.fit keeps the graph through torch.func.jacrev (the fork's
            # GGNInterface.jacobians defaults to enable_backprop=True, laplace/curvature/curvature.py:89-130, 412-432), so
            # neg_marglik.backward() reaches the STE model's adjacency.  Same stored entries, same 200 non-edges.
            torch.manual_seed(seed)
            ste_d = ns.gnn_models.STEGCN(f, h, c, layers, X, adj0.clone(), dropout_p=0.5, threshold=0.5, symmetric=symmetric,
                                         **extras)
            same_extras(ste_d)
            ste_d.eval()
            for a, b in zip(ste_d.convs, model.convs):
                assert torch.equal(a.lin.weight, b.lin.weight) and torch.equal(a.lin.bias, b.lin.bias)
            ld_ = bl.DiagLaplace(ste_d, "classification", prior_precision=0.7)
            ld_.fit(loader)
            negd = -ld_.log_marginal_likelihood()
            negd.backward()
            grd = ste_d.adj.grad.detach().numpy()
            out["adjgrad_diag_neg_marglik"] = np.float64(float(negd))
            out["adjgrad_diag_vals"] = grd[out["adj_nz_row"], out["adj_nz_col"]].astype(np.float32)
            out["adjgrad_diag_ne_val"] = grd[ner[pick], nec[pick]].astype(np.float32)

    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB  kron_loss={out['kron_loss']:.4f}")


def make_regression_mlp(ns, torch):
    """BASELINE config 1: examples/regression_example.py:17-21 model (1-50-1 tanh), data
    re-stated from examples/helper/dataloaders.py:39-47 (the helper imports torchvision)."""
    from torch.utils.data import DataLoader, TensorDataset

    torch.manual_seed(711)
    n = 150
    X = (torch.rand(n) * 8).unsqueeze(-1)
    y = torch.sin(X) + torch.randn_like(X) * 0.3
    model = torch.nn.Sequential(torch.nn.Linear(1, 50), torch.nn.Tanh(), torch.nn.Linear(50, 1))
    loader = DataLoader(TensorDataset(X, y), batch_size=n)
    bl = ns.baselaplace
    ld = bl.DiagLaplace(model, "regression")
    ld.fit(loader)
    lk = bl.KronLaplace(model, "regression")
    lk.fit(loader)
    out = {
        "X": X.numpy(), "y": y.numpy(),
        "W0": model[0].weight.detach().numpy(), "b0": model[0].bias.detach().numpy(),
        "W1": model[2].weight.detach().numpy(), "b1": model[2].bias.detach().numpy(),
        "diag_H": ld.H.detach().numpy(), "diag_loss": np.float32(float(ld.loss)),
        "kron_loss": np.float32(float(lk.loss)), "kron_n_blocks": len(lk.H_facs.kfacs),
    }
    for i, Fs in enumerate(lk.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            out[f"kron_{i}_{j}"] = Hm.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "c1_regression_mlp.npz"), **out)
    print("c1_regression_mlp written")


def make_structure_loop(ns, torch, name, structure, seed, symmetric, grad_norm, lr_adj, momentum, weight_decay, sign_grad=False,
                        masked=False, steps=3, n=64, f=12, h=8, c=3, n_edges=150, n_train=33, batch_size=12, prior=1.0,
                        norm=None, res=False):
    """The fork's structure-learning loop itself (gnn/marglik_training.py:197-224): fit, neg_marglik.backward(), optional
    clip_grad_norm_ on the adjacency (:213-215), ``adj_optimizer.step()`` (SGD with momentum and weight decay, :97-99), refit --
    ``steps`` times, run by the reference's own STEGCN (gnn/models/models.py:65-118, BinarizeSTE gnn/models/utils.py:42-86) and
    KronLaplace / DiagLaplace.  Stored: the inputs, the continuous adjacency parameter after every step (dense, N = 64) and the
    negative log marginal likelihood before the first step and after every refit."""
    from torch.utils.data import DataLoader, TensorDataset

    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, n_edges), generator=g)
    X = torch.randn(n, f, generator=g)
    adj0 = reference_dense_adj(torch, ei, n)
    perm = torch.randperm(n, generator=g)
    train_idx = perm[:n_train].clone()
    train_idx[3] = train_idx[5]
    train_y = torch.randint(0, c, (n_train,), generator=g)
    loader = DataLoader(TensorDataset(train_idx, train_y), batch_size=batch_size, shuffle=False)
    torch.manual_seed(seed)
    extras = dict(norm=norm, res=res) if (norm is not None or res) else {}  # the WebKB / Circle configurations
    ste = ns.gnn_models.STEGCN(f, h, c, 2, X, adj0.clone(), dropout_p=0.5, threshold=0.5, symmetric=symmetric,
                               sign_grad=sign_grad, train_masked_update=masked, train_nodes=train_idx if masked else None,
                               **extras)
    if norm is not None:
        gn = torch.Generator().manual_seed(seed + 1000)
        with torch.no_grad():
            for nm in ste.norms:
                nm.weight.copy_(0.5 + torch.rand(h, generator=gn))
                nm.bias.copy_(0.3 * torch.randn(h, generator=gn))
    ste.eval()
    out = {"kind": "gcn", "symmetric": symmetric, "num_nodes": n, "batch_size": batch_size, "edge_index": ei.numpy(),
           "X": X.numpy(), "train_idx": train_idx.numpy(), "train_y": train_y.numpy(), "num_layers": 2,
           "structure": structure, "grad_norm": bool(grad_norm), "lr_adj": np.float64(lr_adj), "momentum": np.float64(momentum),
           "weight_decay": np.float64(weight_decay), "sign_grad": bool(sign_grad), "masked": bool(masked),
           "prior": np.float64(prior), "threshold": np.float64(0.5)}
    for l, conv in enumerate(ste.convs):
        out[f"W{l}"] = conv.lin.weight.detach().numpy().copy()
        out[f"b{l}"] = conv.lin.bias.detach().numpy().copy()
    if extras:
        out["norm"], out["res"] = str(norm), bool(res)
        for l, lin in enumerate(ste.res):
            out[f"Wr{l}"], out[f"br{l}"] = lin.weight.detach().numpy().copy(), lin.bias.detach().numpy().copy()
        if norm is not None:
            out["norm_eps"] = np.float64(ste.norms[0].eps)
            for l, nm in enumerate(ste.norms):
                out[f"norm_w{l}"], out[f"norm_b{l}"] = nm.weight.detach().numpy().copy(), nm.bias.detach().numpy().copy()
    out["adj_init"] = ste.adj.detach().numpy().copy()  # (symmetrised 0/1 with the GCN's self loops)
    bl = ns.baselaplace
    cls = bl.KronLaplace if structure == "kron" else bl.DiagLaplace
    lap = cls(ste, "classification", prior_precision=prior)
    lap.fit(loader)
    neg = -lap.log_marginal_likelihood()
    opt = torch.optim.SGD([ste.adj], lr=lr_adj, weight_decay=weight_decay, momentum=momentum)
    negs, adjs, grads = [float(neg)], [], []
    for _ in range(steps):
        opt.zero_grad()
        neg.backward()
        grads.append(ste.adj.grad.detach().numpy().copy())  # (before clipping)
        if grad_norm:
            torch.nn.utils.clip_grad_norm_(ste.adj, max_norm=1.0)
        opt.step()
        lap.fit(loader)
        neg = -lap.log_marginal_likelihood()
        negs.append(float(neg))
        adjs.append(ste.adj.detach().numpy().copy())
    out["neg_marglik"] = np.array(negs, dtype=np.float64)
    out["adj_steps"] = np.stack(adjs).astype(np.float32)
    out["grad_steps"] = np.stack(grads).astype(np.float32)
    flips = [int(((a > 0.5) != (b > 0.5)).sum()) for a, b in zip([out["adj_init"]] + adjs[:-1], adjs)]
    margin = min(float(np.abs(a - 0.5).min()) for a in adjs)
    print(f"  smallest distance of a value from the threshold over the steps: {margin:.2e}")
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB  neg_marglik={negs}  entries flipped per step={flips}")


def main():
    import torch

    torch.set_num_threads(8)
    ns = ref_loader.load()
    only = set(sys.argv[1:])  # optional: fixture names to (re)generate; default all
    real_make_case, real_mlp = make_case, make_regression_mlp

    def make_case_(ns_, torch_, name, *a, **k):
        if not only or name in only:
            real_make_case(ns_, torch_, name, *a, **k)
    globals()["make_case"] = make_case_
    small = dict(n=64, f=12, h=8, c=3, layers=2, n_edges=150)
    mid = dict(n=512, f=64, h=32, c=7, layers=2, n_edges=1800)
    for seed in (0, 1, 2):
        make_case(ns, torch, f"gcn_small_1batch_s{seed}", "gcn", **small, n_train=33,
                  batch_size=10000, seed=seed, with_full=(seed == 0))
        make_case(ns, torch, f"gcn_small_3batch_s{seed}", "gcn", **small, n_train=33,
                  batch_size=12, seed=seed, symmetric=(seed == 1))
    make_case(ns, torch, "gcn_small_isolated_s0", "gcn", **small, n_train=40, batch_size=16,
              seed=10, isolated=5)
    make_case(ns, torch, "gcn_mid_1batch_s0", "gcn", **mid, n_train=200, batch_size=10000, seed=0)
    make_case(ns, torch, "gcn_mid_3batch_sym_s1", "gcn", **mid, n_train=200, batch_size=77,
              seed=1, symmetric=True)
    make_case(ns, torch, "sage_small_1batch_s0", "sage", **small, n_train=33, batch_size=10000,
              seed=0, with_full=True)
    make_case(ns, torch, "sage_small_3batch_s1", "sage", **small, n_train=33, batch_size=12,
              seed=1, isolated=4)
    make_case(ns, torch, "sage_mid_2batch_s2", "sage", **mid, n_train=150, batch_size=100, seed=2)
    make_case(ns, torch, "sage_small_3batch_sym_s3", "sage", **small, n_train=33, batch_size=12, seed=3, symmetric=True,
              isolated=2)
    # three layers: the reference only builds them once the live breakpoint at gnn/models/base_gnn.py:109 is a no-op
    # (ref_loader installs one); kron / diag / full GGN / Jacobians all come from the reference's own code
    small3 = dict(n=64, f=12, h=8, c=3, layers=3, n_edges=150)
    mid3 = dict(n=384, f=40, h=24, c=5, layers=3, n_edges=1300)
    make_case(ns, torch, "gcn3_small_1batch_s0", "gcn", **small3, n_train=33, batch_size=10000, seed=20, with_full=True)
    make_case(ns, torch, "gcn3_small_3batch_sym_s1", "gcn", **small3, n_train=33, batch_size=12, seed=21, symmetric=True)
    make_case(ns, torch, "sage3_small_1batch_s0", "sage", **small3, n_train=33, batch_size=10000, seed=22, with_full=True)
    make_case(ns, torch, "sage3_small_3batch_s1", "sage", **small3, n_train=33, batch_size=12, seed=23, isolated=3)
    make_case(ns, torch, "gcn3_mid_2batch_s0", "gcn", **mid3, n_train=120, batch_size=70, seed=24)
    make_case(ns, torch, "sage3_mid_2batch_s1", "sage", **mid3, n_train=120, batch_size=70, seed=25, symmetric=True)
    # res=True / norm="layer"|"batch" (gnn/models/base_gnn.py:86-113, 141-149; gnn/configs/original/gcn_config.yaml:36-58 ship
    # norm: layer + res: True): kron / diag / logits (and full GGN + Jacobians on the *_1batch cases) from the reference
    ex = dict(with_full=True, full_only=True)
    make_case(ns, torch, "gcn_resln_small_1batch_s0", "gcn", **small, n_train=33, batch_size=10000, seed=30, norm="layer",
              res=True, **ex)
    make_case(ns, torch, "gcn_resln_small_3batch_s1", "gcn", **small, n_train=33, batch_size=12, seed=31, norm="layer",
              res=True, symmetric=True)
    make_case(ns, torch, "sage_resln_small_1batch_s0", "sage", **small, n_train=33, batch_size=10000, seed=32,
              norm="layer", res=True, **ex)
    make_case(ns, torch, "sage_resln_small_3batch_s1", "sage", **small, n_train=33, batch_size=12, seed=33, norm="layer",
              res=True, isolated=3)
    make_case(ns, torch, "gcn_res_small_3batch_s2", "gcn", **small, n_train=33, batch_size=12, seed=34, res=True)
    make_case(ns, torch, "gcn_bn_small_3batch_s3", "gcn", **small, n_train=33, batch_size=12, seed=35, norm="batch")
    make_case(ns, torch, "sage_resbn_small_1batch_s4", "sage", **small, n_train=33, batch_size=10000, seed=36,
              norm="batch", res=True)
    make_case(ns, torch, "gcn_resln_mid_2batch_s0", "gcn", **mid, n_train=150, batch_size=100, seed=37, norm="layer",
              res=True)
    make_case(ns, torch, "sage_ln_mid_2batch_s1", "sage", **mid, n_train=150, batch_size=100, seed=38, norm="layer")
    make_case(ns, torch, "gcn3_resln_small_3batch_s5", "gcn", **small3, n_train=33, batch_size=12, seed=39, norm="layer",
              res=True)
    make_case(ns, torch, "sage3_resln_small_1batch_s6", "sage", **small3, n_train=33, batch_size=10000, seed=40,
              norm="layer", res=True, **ex)
    # norm="layer" without res: the Circle configuration (gnn/configs/original/stegcn_config.yaml:129-145)
    make_case(ns, torch, "gcn_ln_small_3batch_sym_s7", "gcn", **small, n_train=33, batch_size=12, seed=41, norm="layer",
              symmetric=True)
    if not only or "c1_regression_mlp" in only:
        make_regression_mlp(ns, torch)
    # the structure-learning loop (VERDICT r3 missing #2): three hyper-steps of the reference's own STEGCN + SGD
    loops = {
        "steloop_kron_sym": dict(structure="kron", seed=62, symmetric=True, grad_norm=True, lr_adj=8.0, momentum=0.9,
                                 weight_decay=5e-4),
        "steloop_kron_dir": dict(structure="kron", seed=51, symmetric=False, grad_norm=False, lr_adj=0.05, momentum=0.5,
                                 weight_decay=0.0, masked=True),
        "steloop_diag_sym": dict(structure="diag", seed=62, symmetric=True, grad_norm=True, lr_adj=8.0, momentum=0.9,
                                 weight_decay=5e-4),
    }
    # Cornell / Texas / Wisconsin (gnn/configs/original/stegcn_config.yaml:54-105): diag + res + LayerNorm, symmetric,
    # momentum 0.9, weight decay 5e-4 on the adjacency, clipped gradient; Circle (:129-145): LayerNorm only.  (seeds: the
    # first of 50..65 whose continuous values stay >= 4e-4 away from the threshold over the three steps)
    loops["steloop_diag_resln_sym"] = dict(structure="diag", seed=55, symmetric=True, grad_norm=True, lr_adj=10.0, momentum=0.9,
                                           weight_decay=5e-4, norm="layer", res=True)
    loops["steloop_kron_ln_sym"] = dict(structure="kron", seed=55, symmetric=True, grad_norm=True, lr_adj=5.0, momentum=0.9,
                                        weight_decay=2e-3, norm="layer")
    # (sign_grad=True is not covered: BinarizeSTE.backward takes the sign of the gradient of EVERY forward call -- the Kronecker
    #  graph and the loss forward of each batch -- and autograd sums those signs, values in {-3 .. 3} with three batches; the
    #  HIP path accumulates one gradient and refuses the option)
    for nm, kw in loops.items():
        if not only or nm in only:
            make_structure_loop(ns, torch, nm, **kw)


if __name__ == "__main__":
    main()
