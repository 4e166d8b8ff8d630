"""GPU parity at the shapes BASELINE.json names beyond the arxiv one (tests/test_gpu_scale.py):

* configs[1]  Cora-shaped 2-layer GCN (N = 2 708, F = 1 433, H = 64, C = 7), diag -- the split-K forward GEMM with a
  ragged last K slice and the first-layer diagonal kernel at width 1 434 are only reached at this shape;
* configs[4]  products-shaped 3-layer GraphSAGE (N = 2 449 029, power-law degrees, C = 47, D = 512), last-layer full
  GGN with P = 24 111 and 1 128 class pairs.

Small sample subsets go against the CPU oracle (fp32 <= 1e-4 relative Frobenius error per block, BASELINE.json
north_star); the full sizes are checked through size-independent properties and an fp64 torch restatement of the
forward pass (sparse matmul) built from the engine's exported propagation matrix.
"""
import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
from gpu_utils import oracle_from_arrays, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _prop_csr(eng, N, dtype=torch.float64):
    rows, cols, vals = eng.export_propagation()
    crow = torch.zeros(N + 1, dtype=torch.int64, device=rows.device)
    crow[1:] = torch.cumsum(torch.bincount(rows, minlength=N), 0)
    return torch.sparse_csr_tensor(crow, cols, vals.to(dtype), size=(N, N))


def _torch_forward(kind, P, X, Ws, bs):
    """fp64 restatement of the eval-mode forward (gnn/models/base_gnn.py:136-161, layers.py:18-46) with a sparse
    propagation matrix; returns (logits, input of the last nn.Linear)."""
    h = X.double()
    L = len(Ws)
    last_in = None
    for l in range(L):
        W, b = Ws[l].double(), bs[l].double()
        if kind == "gcn":
            last_in = h
            z = torch.sparse.mm(P, h @ W.T + b)
        else:
            cat = torch.cat([h, torch.sparse.mm(P, h)], dim=1)
            last_in = cat
            z = cat @ W.T + b
        h = torch.relu(z) if l < L - 1 else z
    return h, last_in


# =====================================================================================================================
# configs[1]: Cora shape
# =====================================================================================================================
@pytest.fixture(scope="module")
def cora():
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("cora", "cuda")
    torch.manual_seed(0)
    model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    return w, ei, X, model, train_idx, train_y


def _blocks(eng):
    off = 0
    for i, o in zip(eng.in_dims, eng.dims[1:]):
        for n in (i * o, o):
            yield off, off + n
            off += n


def test_cora_shape_diag_and_kron_subset_vs_oracle(cora):
    """64 batch nodes of the Cora-shaped model (the oracle's Jacobians are 165 MB): diagonal GGN per parameter block,
    KFAC factors and both losses against the CPU oracle."""
    w, ei, X, model, train_idx, train_y = cora
    eng = model.engine
    idx, y = train_idx[:64], train_y[:64]
    Ws = [c.lin.weight.detach().cpu().numpy() for c in model.convs]
    bs = [c.lin.bias.detach().cpu().numpy() for c in model.convs]
    om = oracle_from_arrays("gcn", w["N"], ei.numpy(), X.numpy(), Ws, bs, True)

    Hd = torch.zeros(eng.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.diag_accumulate(idx.cuda(), y.cuda(), Hd, loss)
    ol, od = O.diag_batch(om, idx.numpy(), y.numpy())
    Hd = Hd.cpu().numpy()
    assert Hd.shape[0] == 1433 * 64 + 64 + 64 * 7 + 7
    for a, b in _blocks(eng):
        assert rel(Hd[a:b], od[a:b]) < RTOL, (a, b)
    assert abs(float(loss) - float(ol)) <= RTOL * abs(float(ol))

    for fuse in (True, False):
        _, views, kl = eng.new_kfac_buffers()
        eng.kfac_accumulate(idx.cuda(), y.cuda(), w["n_train"], views, kl, fuse=fuse)
        okl, oH = O.kfac_batch(om, idx.numpy(), y.numpy(), w["n_train"])
        for l, (A, B) in enumerate(views):
            assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, ("B", l, fuse)
            assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, ("A", l, fuse)
        assert abs(float(kl) - float(okl)) <= RTOL * abs(float(okl))
    eng.check_async_errors()


def test_cora_shape_full_size_properties(cora):
    """The whole Cora-shaped fit (one batch of 1 299) through properties that need no CPU reference:
    * logits of the split-K forward GEMM (K = 1 433: 12 slices, ragged tail, float atomics) vs an fp64 torch forward;
    * loss == torch cross entropy on the engine's logits;
    * the diagonal GGN is a sum over samples: one batch == four ragged sub-batches;
    * closed-form diagonal kernels == the contraction of explicit device Jacobians (the generic-depth route) with
      Lambda in fp64, on a slice of the batch;
    * KFAC: fused == unfused, A_0 == X^T X / N_train (fp64), factors symmetric PSD."""
    w, ei, X, model, train_idx, train_y = cora
    eng = model.engine
    N, C = w["N"], w["C"]
    idx, y = train_idx.cuda(), train_y.cuda()
    Ws = [c.lin.weight.detach() for c in model.convs]
    bs = [c.lin.bias.detach() for c in model.convs]
    ref_logits, _ = _torch_forward("gcn", _prop_csr(eng, N), X.cuda(), Ws, bs)
    logits = eng.forward_all()
    assert rel(logits.cpu().numpy(), ref_logits.cpu().numpy()) < 1e-5
    assert rel(model(idx).cpu().numpy(), ref_logits[idx].cpu().numpy()) < 1e-5

    H1 = torch.zeros(eng.n_params, device="cuda")
    l1 = torch.zeros(1, device="cuda")
    eng.diag_accumulate(idx, y, H1, l1)
    ce = torch.nn.functional.cross_entropy(ref_logits[idx], y, reduction="sum")
    assert abs(float(l1) - float(ce)) < 1e-5 * float(ce)
    H4 = torch.zeros_like(H1)
    l4 = torch.zeros(1, device="cuda")
    for s, e in ((0, 500), (500, 501), (501, 1000), (1000, len(idx))):
        eng.diag_accumulate(idx[s:e], y[s:e], H4, l4)
    assert rel(H4.cpu().numpy(), H1.cpu().numpy()) < 1e-5
    assert abs(float(l4) - float(l1)) < 1e-5 * float(l1)
    assert float(H1.min()) >= 0.0

    sl = slice(100, 124)
    Js, f = eng.jacobians(idx[sl])
    assert Js.shape == (24, C, eng.n_params)
    p = torch.softmax(f.double(), -1)
    Jd = Js.double()
    t1 = torch.einsum("mc,mcp->mp", p, Jd)
    ref_diag = torch.einsum("mc,mcp->p", p, Jd * Jd) - (t1 * t1).sum(0)
    Hs = torch.zeros(eng.n_params, device="cuda")
    ls = torch.zeros(1, device="cuda")
    eng.diag_accumulate(idx[sl], y[sl], Hs, ls)
    for a, b in _blocks(eng):
        assert rel(Hs[a:b].cpu().numpy(), ref_diag[a:b].cpu().numpy()) < RTOL, (a, b)

    _, v_f, loss_f = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], v_f, loss_f, fuse=True)
    _, v_u, loss_u = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], v_u, loss_u, fuse=False)
    torch.cuda.synchronize()
    for (Af, Bf), (Au, Bu) in zip(v_f, v_u):
        assert rel(Bf.cpu().numpy(), Bu.cpu().numpy()) < 1e-5
        assert torch.equal(Af, Au) and torch.equal(Bf, Bf.T) and torch.equal(Af, Af.T)
        ev = torch.linalg.eigvalsh(Bf.double())
        assert float(ev.min()) > -1e-5 * float(ev.max())
    Xd = X.cuda().double()
    assert rel(v_f[0][0].cpu().numpy(), (Xd.T @ Xd / w["n_train"]).cpu().numpy()) < 1e-5
    assert abs(float(loss_f) - float(ce)) < 1e-5 * float(ce)
    eng.check_async_errors()


def test_cora_shape_diag_laplace_front(cora):
    """``Laplace(model, 'classification', 'all', 'diag').fit`` at the Cora shape == the engine-level accumulate, and the
    marginal likelihood is finite (BASELINE configs[1] end to end through the backend class)."""
    import laplace_gnn_amd as lg

    w, ei, X, model, train_idx, train_y = cora
    loader = lg.TensorBatchLoader(train_idx.cuda(), train_y.cuda(), batch_size=w["batch"])
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure="diag")
    la.fit(loader)
    H = torch.zeros(model.engine.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    model.engine.diag_accumulate(train_idx.cuda(), train_y.cuda(), H, loss)
    assert rel(la.H.cpu().numpy(), H.cpu().numpy()) < 1e-6 and la.n_data == w["n_train"]
    assert torch.isfinite(la.log_marginal_likelihood())


# =====================================================================================================================
# configs[4]: products shape
# =====================================================================================================================
@pytest.fixture(scope="module")
def products():
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("products", "cuda")
    torch.manual_seed(0)
    model = lg.GraphSAGE(w["F"], w["H"], w["C"], w["layers"], X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    Ws = [c.lin.weight.detach() for c in model.convs]
    bs = [c.lin.bias.detach() for c in model.convs]
    ref_logits, last_in = _torch_forward("sage", _prop_csr(eng, w["N"]), model.X, Ws, bs)
    yield w, model, train_idx.cuda(), train_y.cuda(), ref_logits, last_in
    eng.close()


def test_products_shape_forward_vs_fp64_torch(products):
    w, model, idx, y, ref_logits, last_in = products
    logits = model.engine.forward_all()
    num = torch.linalg.norm(logits.double() - ref_logits)
    assert float(num / torch.linalg.norm(ref_logits)) < 1e-5


def test_products_shape_lastlayer_full_properties(products):
    """One full-size batch (10 000 samples, P = 24 111, 1 128 class pairs) of ``lgnn_lastlayer_full_accumulate``:
    exactly symmetric, positive semi-definite, bias block == sum_n Lambda_n, diagonal == the per-class weighted
    column norms sum_n p_nc (1 - p_nc) phi_n[d]^2 (both from the fp64 torch forward), loss == cross entropy,
    linear in the number of calls; then a 200-sample call against the oracle's Jacobian einsum on a random subset of
    2 000 parameters (every class pair is hit)."""
    w, model, idx, y, ref_logits, last_in = products
    eng = model.engine
    C, D = w["C"], 2 * w["H"]
    P = C * D + C
    M = w["batch"]
    b = slice(0, M)
    H = torch.zeros(P, P, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.lastlayer_full_accumulate(idx[b], y[b], H, loss)
    torch.cuda.synchronize()
    ce = torch.nn.functional.cross_entropy(ref_logits[idx[b]], y[b], reduction="sum")
    assert abs(float(loss) - float(ce)) < 1e-5 * float(ce)
    assert bool((H == H.T).all())
    p = torch.softmax(ref_logits[idx[b]], -1)                       # [M, C] fp64
    phi = last_in[idx[b]]                                            # [M, D] fp64
    Lam_sum = torch.diag(p.sum(0)) - p.T @ p
    assert rel(H[C * D:, C * D:].cpu().numpy(), Lam_sum.cpu().numpy()) < RTOL
    dw = (p * (1 - p)).T @ (phi * phi)                               # [C, D]
    hd = torch.diagonal(H)
    assert rel(hd[:C * D].cpu().numpy(), dw.reshape(-1).cpu().numpy()) < RTOL
    assert rel(hd[C * D:].cpu().numpy(), (p * (1 - p)).sum(0).cpu().numpy()) < RTOL
    # one weight block against its definition: H[(c, :), (c', :)] = Phi^T diag(Lambda[:, c, c']) Phi
    for c, c2 in ((0, 0), (3, 41), (46, 46)):
        lam = (p[:, c] if c == c2 else 0) - p[:, c] * p[:, c2]
        blk = phi.T @ (lam[:, None] * phi)
        assert rel(H[c * D:(c + 1) * D, c2 * D:(c2 + 1) * D].cpu().numpy(), blk.cpu().numpy()) < RTOL, (c, c2)
    # positive semi-definite up to fp32 rounding: lambda_min >= -1e-5 lambda_max, i.e. the Cholesky factorisation of
    # H + 1e-5 lambda_max I succeeds in fp64 (H has an exact null space of dimension D + 1: Lambda_n 1 = 0, so the
    # shift has to be relative to the spectrum's top, found by power iteration, not to the diagonal)
    Hd = H.double()
    v = torch.ones(P, 1, device="cuda", dtype=torch.float64)
    for _ in range(30):
        v = Hd @ v
        lam_max = float(torch.linalg.norm(v))
        v /= lam_max
    Hd.diagonal().add_(1e-5 * lam_max)
    _, info = torch.linalg.cholesky_ex(Hd)
    assert int(info) == 0, (int(info), lam_max)
    del Hd
    # a second call adds the same again (accumulate semantics; the mirror pass must not double the lower triangle)
    Hs = H[:2000, :2000].clone()
    eng.lastlayer_full_accumulate(idx[b], y[b], H, loss)
    assert rel(H[:2000, :2000].cpu().numpy(), 2 * Hs.cpu().numpy()) < 1e-6
    assert bool((H[:3000, :3000] == H[:3000, :3000].T).all())

    # 200 samples vs the oracle's einsum on 2 000 random parameters
    H.zero_()
    loss.zero_()
    s = slice(M, M + 200)
    eng.lastlayer_full_accumulate(idx[s], y[s], H, loss)
    rng = np.random.default_rng(0)
    sub = np.sort(rng.choice(P, 2000, replace=False))
    sub[-C:] = np.arange(C * D, P)  # all bias entries
    sub = np.unique(sub)
    phis = last_in[idx[s]].float().cpu().numpy()
    fs = ref_logits[idx[s]].float().cpu().numpy()
    oh = O.lastlayer_full_from_features(phis, np.ones(200, np.float32), fs, params=sub)
    st = torch.from_numpy(sub).cuda()
    assert rel(H[st][:, st].cpu().numpy(), oh) < RTOL
    eng.check_async_errors()


def test_products_shape_lastlayer_front_two_batches(products):
    """FullLLLaplace.fit over two full batches accumulates in place into one P x P buffer: equals the sum of the two
    engine-level calls on a sub-block, n_data / loss bookkeeping as the reference's fit loop."""
    import laplace_gnn_amd as lg

    w, model, idx, y, ref_logits, last_in = products
    M = w["batch"]
    loader = lg.TensorBatchLoader(idx[:2 * M], y[:2 * M], batch_size=M)
    la = lg.Laplace(model, "classification", subset_of_weights="last_layer", hessian_structure="full")
    la.fit(loader)
    ce = torch.nn.functional.cross_entropy(ref_logits[idx[:2 * M]], y[:2 * M], reduction="sum")
    assert abs(float(la.loss) - float(ce)) < 1e-5 * float(ce) and la.n_data == 2 * M
    C, D = w["C"], 2 * w["H"]
    p = torch.softmax(ref_logits[idx[:2 * M]], -1)
    Lam_sum = torch.diag(p.sum(0)) - p.T @ p
    assert rel(la.H[C * D:, C * D:].cpu().numpy(), Lam_sum.cpu().numpy()) < RTOL
    del la


def test_lastlayer_full_products_head_on_a_small_graph_vs_oracle():
    """The products head (C = 47, D = 512, P = 24 111) on a graph the oracle can run end to end, with a workspace cap
    that forces the class pairs through many chunks (qc_max < Q = 1 128)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E, M = 3000, 24, 256, 47, 14000, 150
    g = torch.Generator().manual_seed(77)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    dims = [F, H, H, C]
    Ws = [torch.randn(dims[l + 1], 2 * dims[l], generator=g) / (2 * dims[l]) ** 0.5 for l in range(3)]
    bs = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(3)]
    idx = torch.randperm(N, generator=g)[:M]
    idx[M // 2] = idx[0]  # a repeated node id
    y = torch.randint(0, C, (M,), generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind="sage", symmetric=True)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    eng.set_workspace_limit(1 << 20)  # the minimum: the class pairs go through two chunks
    P = C * 2 * H + C
    assert P == 24111
    Hl = torch.zeros(P, P, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.lastlayer_full_accumulate(idx.cuda(), y.cuda(), Hl, loss)
    om = oracle_from_arrays("sage", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    phi, s, f = O.lastlayer_features(om, idx.numpy())
    rng = np.random.default_rng(1)
    sub = np.unique(np.concatenate([rng.choice(P, 2500, replace=False), np.arange(C * 2 * H, P)]))
    oh = O.lastlayer_full_from_features(phi, s, f, params=sub)
    st = torch.from_numpy(sub).cuda()
    assert rel(Hl[st][:, st].cpu().numpy(), oh) < RTOL
    assert abs(float(loss) - float(O.ce_sum(f, y.numpy()))) <= RTOL * abs(float(O.ce_sum(f, y.numpy())))
    assert bool((Hl == Hl.T).all())
    eng.check_async_errors()
    eng.close()


# =====================================================================================================================
# every workspace / kernel-choice outcome of the KFAC path, each under the smallest workspace cap
# =====================================================================================================================
@pytest.mark.parametrize("kind,H,C,L,fuse,expect", [
    # (need_pong, fused per backward step l = L-1 .. 1)
    ("gcn", 64, 6, 2, True, (False, [True])),     # two layers fused: the pong planes are never allocated
    ("gcn", 30, 6, 2, True, (True, [False])),     # width % 4 != 0: SpMM output goes through pong
    ("gcn", 64, 6, 2, False, (True, [False])),    # LGNN_FLAG_NO_FUSE (the launch that aborted in round 1)
    ("gcn", 64, 6, 3, True, (True, [True, True])),  # three layers: g_1 is stored to pong for the layer below
    ("sage", 33, 5, 2, True, (True, [False])),    # GraphSAGE, activation row stride 66 (% 4 != 0): unfused
    ("sage", 64, 5, 2, True, (False, [True])),    # GraphSAGE compact top level
    ("sage", 64, 5, 3, True, (True, [True, True])),
    ("gcn", 256, 6, 2, True, (False, [True])),    # 256-wide kernel + compacted backward GEMM
])
def test_every_need_pong_outcome_under_the_smallest_workspace(kind, H, C, L, fuse, expect):
    import laplace_gnn_amd as lg

    N, F, E = 900, 20, 3400
    g = torch.Generator().manual_seed(31)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    mult = 2 if kind == "sage" else 1
    dims = [F] + [H] * (L - 1) + [C]
    Ws = [torch.randn(dims[l + 1], mult * dims[l], generator=g) / (mult * dims[l]) ** 0.5 for l in range(L)]
    bs = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(L)]
    idx = torch.randperm(N, generator=g)[:150]
    y = torch.randint(0, C, (150,), generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind=kind, symmetric=True)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    eng.set_workspace_limit(1 << 20)  # the minimum: one or two class planes per chunk
    plan = eng.kfac_plan(fuse=fuse, paths=False)  # the class-plane routes (the path route has its own tests)
    assert plan["need_pong"] == expect[0], plan
    assert plan["fused"][1:][::-1] == expect[1], plan
    assert plan["classes_per_chunk"] < C, plan
    _, views, loss = eng.new_kfac_buffers()
    for s in (0, 80):
        eng.kfac_accumulate(idx[s:s + 80].cuda(), y[s:s + 80].cuda(), 150, views, loss, fuse=fuse, paths=False)
    torch.cuda.synchronize()
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 80)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, ("B", l)
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, ("A", l)
    assert abs(float(loss) - float(oloss)) <= RTOL * abs(float(oloss))
    eng.check_async_errors()
    eng.close()


def test_rebinding_a_graphsage_context_to_other_widths():
    """The compact GraphSAGE top level keeps its plane buffer zero outside the batch rows for one layout; a new binding
    (narrower hidden width, same context) must start from a clean claim (stale spare rows of the old layout)."""
    import laplace_gnn_amd as lg

    N, F, C, E = 900, 16, 6, 4000
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    idx = torch.randperm(N, generator=g)[:200]
    y = torch.randint(0, C, (200,), generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind="sage", symmetric=True)
    for H in (256, 192, 64, 256):
        dims = [F, H, C]
        Ws = [torch.randn(dims[l + 1], 2 * dims[l], generator=g) / (2 * dims[l]) ** 0.5 for l in range(2)]
        bs = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(2)]
        eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
        assert eng.kfac_plan(paths=False)["sage_compact"]
        om = oracle_from_arrays("sage", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
        _, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 120)
        for paths in (False, None):  # the compact class-plane route this test is about, then the default (H > 128: one-hop paths)
            _, views, loss = eng.new_kfac_buffers()
            for s in (0, 120):
                eng.kfac_accumulate(idx[s:s + 120].cuda(), y[s:s + 120].cuda(), 200, views, loss, paths=paths)
            torch.cuda.synchronize()
            for l, (A, B) in enumerate(views):
                assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, (H, "B", l, paths)
                assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, (H, "A", l, paths)
    eng.close()
