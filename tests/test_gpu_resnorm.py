"""GPU parity of models built with ``res=True`` / ``norm="layer"|"batch"`` (gnn/models/base_gnn.py:86-113, 141-149)
beyond the golden sizes: seeded mid-size graphs (hubs, ragged last batch, a duplicated node id, 2 and 3 layers) against
the CPU oracle, which is pinned to the reference on the ``*_res*`` / ``*_bn_*`` / ``*_ln_*`` fixtures
(tests/test_oracle_golden.py).  fp32 <= 1e-4 relative Frobenius error per block (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
import os

from gpu_utils import kfac_fit_engine, oracle_from_arrays, rel
from test_gpu_scale import _make

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _extras(H, L, norm, res, in_dims, seed):
    g = torch.Generator().manual_seed(seed)
    kw = {}
    if res:
        kw["res_weights"] = [torch.randn(H, in_dims[l], generator=g) / in_dims[l] ** 0.5 for l in range(L - 1)]
        kw["res_biases"] = [0.1 * torch.randn(H, generator=g) for _ in range(L - 1)]
    if norm is not None:
        kw.update(norm=norm, norm_eps=1e-5, norm_weight=[0.5 + torch.rand(H, generator=g) for _ in range(L - 1)],
                  norm_bias=[0.3 * torch.randn(H, generator=g) for _ in range(L - 1)])
        if norm == "batch":
            kw.update(norm_mean=[0.2 * torch.randn(H, generator=g) for _ in range(L - 1)],
                      norm_var=[0.5 + torch.rand(H, generator=g) for _ in range(L - 1)])
    return kw


def _both(kind, N, F, H, C, E, L, norm, res, seed, skew=False):
    import laplace_gnn_amd as lg

    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=L, seed=seed, skew=skew)
    in_dims = [F] + [H] * (L - 2)
    kw = _extras(H, L, norm, res, in_dims, seed + 50)
    eng = lg.GraphEngine(ei.cuda(), N, kind=kind, symmetric=True)
    dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else v) for k, v in kw.items()}
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs], **dev)
    host = {k: ([t.numpy() for t in v] if isinstance(v, list) else v) for k, v in kw.items()}
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True, **host)
    return eng, om


@pytest.mark.parametrize("kind,L,norm,res,skew", [
    ("gcn", 2, "layer", True, False), ("gcn", 2, "layer", True, True), ("sage", 2, "layer", True, True),
    ("gcn", 3, "layer", True, False), ("sage", 3, "layer", True, False), ("gcn", 2, "batch", False, False),
    ("sage", 2, "batch", True, False), ("gcn", 2, None, True, False), ("sage", 3, None, True, True),
    ("gcn", 3, "batch", True, True)])
def test_kfac_with_res_and_norm_midsize_vs_oracle(kind, L, norm, res, skew):
    N, F, H, C, E = 3000, 72, 64, 9, 12000
    eng, om = _both(kind, N, F, H, C, E, L, norm, res, seed=11, skew=skew)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(N, generator=g)[:700]
    idx[7] = idx[11]  # a node listed twice in one batch
    y = torch.randint(0, C, (700,), generator=g)
    assert rel(eng.forward_all().cpu().numpy(), O.forward_all(om)[0]) < RTOL
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 300)  # 300 / 300 / 100 (ragged)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 300)
    assert len(views) == len(oH) // 2 == (2 * L - 1 if res else L)
    for k, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * k][0]) < RTOL, f"B of block {k}"
        assert rel(A.cpu().numpy(), oH[2 * k][1]) < RTOL, f"A of block {k}"
        assert torch.equal(B, B.T) and torch.equal(A, A.T)
    assert abs(loss - float(oloss)) < RTOL * abs(float(oloss))
    # class ranges stay exact additive shares (the multi-GPU units, lgnn_kfac_accumulate_classes)
    flat, v2, l2 = eng.new_kfac_buffers()
    for s in range(0, 700, 300):
        for cr in ((0, 4), (4, C)):
            eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, v2, l2, classes=cr)
    torch.cuda.synchronize()
    for (A, B), (A2, B2) in zip(views, v2):
        assert rel(B2.cpu().numpy(), B.cpu().numpy()) < 1e-5 and rel(A2.cpu().numpy(), A.cpu().numpy()) < 1e-5
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("kind,L,norm,res", [("gcn", 2, "layer", True), ("sage", 2, "layer", True), ("gcn", 3, "layer", True),
                                            ("sage", 3, "batch", True), ("gcn", 2, "batch", False)])
def test_jacobians_diag_and_full_with_res_and_norm_vs_oracle(kind, L, norm, res):
    N, F, H, C, E = 600, 20, 24, 4, 2400
    eng, om = _both(kind, N, F, H, C, E, L, norm, res, seed=23)
    g = torch.Generator().manual_seed(9)
    idx = torch.randperm(N, generator=g)[:40]
    y = torch.randint(0, C, (40,), generator=g)
    assert eng.n_params == om.n_params
    Js, f = eng.jacobians(idx.cuda())
    oJ, of = O.jacobians_batch(om, idx.numpy())
    assert rel(Js.cpu().numpy(), oJ) < RTOL and rel(f.cpu().numpy(), of) < RTOL
    Hd = torch.zeros(eng.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.diag_accumulate(idx.cuda(), y.cuda(), Hd, loss)
    ol, od = O.diag_batch(om, idx.numpy(), y.numpy())
    assert rel(Hd.cpu().numpy(), od) < RTOL and abs(float(loss.item()) - float(ol)) < RTOL * abs(float(ol))
    Hf = torch.zeros(eng.n_params, eng.n_params, device="cuda")
    loss.zero_()
    eng.full_accumulate(idx.cuda(), y.cuda(), Hf, loss)
    _, of_ = O.full_batch(om, idx.numpy(), y.numpy())
    assert rel(Hf.cpu().numpy(), of_) < RTOL
    # the empirical Fisher's per-sample gradients go through the same Jacobians
    G = eng.ef_accumulate(idx.cuda(), y.cuda(), y.cuda(), grads=True, loss=loss)
    oG, _ = O.ef_gradients(om, idx.numpy(), y.numpy())
    assert rel(G.cpu().numpy(), oG) < RTOL
    eng.close()


def test_unsupported_routes_refuse_models_with_res_or_norm():
    """The matrix-free GLM variance is a closed form of plain 2-layer models, and the adjacency gradient with res / norm covers
    GCNs (the reference's STEGCN configurations): anything else must say so (the Laplace front then takes the Jacobian route
    for the predictive)."""
    import laplace_gnn_amd as lg
    from laplace_gnn_amd._lib import HipLibraryError

    eng, _ = _both("gcn", 200, 8, 8, 3, 600, 2, "layer", True, seed=2)
    idx = torch.arange(10).cuda()
    one = torch.ones(3, 9, device="cuda")
    with pytest.raises(HipLibraryError, match="res / norm"):
        eng.glm_variance(idx, torch.ones(8, 9, device="cuda"), torch.ones(3, 8, device="cuda"), torch.ones(3, device="cuda"))
    eng.close()
    eng, _ = _both("sage", 200, 8, 8, 3, 600, 2, "layer", True, seed=2)
    with pytest.raises(HipLibraryError, match="res / norm"):
        eng.adjgrad_batch(idx, torch.zeros(10, dtype=torch.int64, device="cuda"),
                          [torch.eye(8, device="cuda"), torch.eye(3, device="cuda"), torch.eye(8, device="cuda")],
                          torch.zeros(eng.nnz, device="cuda"), torch.zeros(200, 3, device="cuda"))
    del one
    eng.close()


def test_rebinding_drops_and_restores_the_extras():
    """lgnn_bind_model resets res / norm; binding the same context again without them gives the plain model's results."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 500, 16, 16, 3, 2000
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, L=2, seed=4)
    kw = _extras(H, 2, "layer", True, [F], 77)
    eng = lg.GraphEngine(ei.cuda(), N, kind="gcn", symmetric=True)
    cw, cb = [w.cuda() for w in Ws], [b.cuda() for b in bs]
    eng.bind(X.cuda(), cw, cb)
    plain = eng.forward_all().clone()
    assert eng.n_params == H * F + H + C * H + C and len(eng.new_kfac_buffers()[1]) == 2
    dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else v) for k, v in kw.items()}
    eng.bind(X.cuda(), cw, cb, **dev)
    assert eng.n_params == H * F + H + C * H + C + H * F + H and len(eng.new_kfac_buffers()[1]) == 3
    assert rel(eng.forward_all().cpu().numpy(), plain.cpu().numpy()) > 1e-2
    # an in-place change of a norm parameter invalidates the cached forward like a weight update does
    before = eng.forward_all().clone()
    dev["norm_weight"][0].mul_(1.5)
    assert rel(eng.forward_all().cpu().numpy(), before.cpu().numpy()) > 1e-3
    eng.bind(X.cuda(), cw, cb)
    assert torch.equal(eng.forward_all(), plain)
    eng.close()


def test_glm_predictive_of_a_res_norm_model_in_chunks_of_evaluation_nodes():
    """Models with res / norm take the Jacobian route of the GLM predictive; its evaluation nodes go in chunks whose Jacobians
    [chunk, C, P] stay under a byte budget (the reference forms all M at once, laplace/baselaplace.py:1123-1158): the chunked
    result equals the single-pass one, diagonal and full covariance."""
    import laplace_gnn_amd as lg

    from test_gpu_frontend import model_from_golden

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gcn_resln_small_3batch_s1.npz"))
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    la = lg.KronLaplace(model, "classification")
    la.fit(loader)
    idx = torch.arange(int(g["num_nodes"])).cuda()
    mu1, var1 = la._glm_predictive_distribution(idx)
    _, d1 = la._glm_predictive_distribution(idx, diagonal_output=True)
    old = type(la)._JACOBIAN_BYTES_MAX
    try:
        type(la)._JACOBIAN_BYTES_MAX = 7 * la.n_outputs * la.n_params * 4  # 7 nodes per chunk: 64 = 9 chunks + 1
        mu2, var2 = la._glm_predictive_distribution(idx)
        _, d2 = la._glm_predictive_distribution(idx, diagonal_output=True)
        probs = la(idx)
    finally:
        type(la)._JACOBIAN_BYTES_MAX = old
    assert torch.equal(mu1, mu2) and mu2.shape == (idx.numel(), la.n_outputs)
    assert rel(var2.cpu().numpy(), var1.cpu().numpy()) < 1e-6 and rel(d2.cpu().numpy(), d1.cpu().numpy()) < 1e-6
    assert probs.shape == (idx.numel(), la.n_outputs) and torch.allclose(probs.sum(dim=1), torch.ones_like(probs[:, 0]), atol=1e-5)
