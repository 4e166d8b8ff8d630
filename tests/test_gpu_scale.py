"""GPU parity beyond the golden sizes: (a) seeded mid-size graphs against the CPU oracle, incl. edge
cases the domain has (skewed degrees, empty edge list, single-sample batch, ragged last batch,
3-layer models); (b) BASELINE-size (arxiv-shaped) runs checked through size-independent properties."""
import os

import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
from gpu_utils import kfac_fit_engine, oracle_from_arrays, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _make(kind, N, F, H, C, E, L=2, seed=0, skew=False, symmetric=True):
    g = torch.Generator().manual_seed(seed)
    if skew:  # a few hubs with thousands of neighbours + a long tail (power-law-like)
        w = torch.rand(N, generator=g) ** 6
        src = torch.multinomial(w + 1e-6, E, replacement=True, generator=g)
        dst = torch.randint(0, N, (E,), generator=g)
        ei = torch.stack([src, dst])
    else:
        ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    mult = 2 if kind == "sage" else 1
    dims = [F] + [H] * (L - 1) + [C]
    Ws = [torch.randn(dims[l + 1], mult * dims[l], generator=g) / (mult * dims[l]) ** 0.5 for l in range(L)]
    bs = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(L)]
    return ei, X, Ws, bs


def _engine(kind, N, ei, X, Ws, bs, symmetric=True):
    import laplace_gnn_amd as lg

    eng = lg.GraphEngine(ei.cuda(), N, kind=kind, symmetric=symmetric)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    return eng


@pytest.mark.parametrize("kind,skew,L", [("gcn", False, 2), ("gcn", True, 2), ("sage", True, 2), ("gcn", False, 3),
                                         ("sage", False, 3), ("gcn", False, 1)])
def test_kfac_midsize_vs_oracle(kind, skew, L):
    N, F, H, C, E = 4000, 96, 64, 10, 16000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=L, seed=3, skew=skew)
    g = torch.Generator().manual_seed(7)
    idx = torch.randperm(N, generator=g)[:900]
    y = torch.randint(0, C, (900,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs)
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 400)  # 400 / 400 / 100 (ragged)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 400)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l}"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l}"
    assert abs(loss - float(oloss)) < RTOL * abs(float(oloss))
    eng.close()


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_diag_and_lastlayer_midsize_vs_oracle(kind):
    N, F, H, C, E = 1500, 50, 32, 6, 5000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, seed=5, skew=True)
    g = torch.Generator().manual_seed(9)
    idx = torch.randperm(N, generator=g)[:300]
    y = torch.randint(0, C, (300,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs)
    Hd = torch.zeros(eng.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    for s in (0, 130):
        eng.diag_accumulate(idx[s:s + 130 if s == 0 else None].cuda(), y[s:s + 130 if s == 0 else None].cuda(), Hd, loss)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    _, oH = O.fit_diag(om, idx.numpy(), y.numpy(), 300)
    assert rel(Hd.cpu().numpy(), oH) < RTOL
    p_ll = Ws[-1].numel() + bs[-1].numel()
    Hl = torch.zeros(p_ll, p_ll, device="cuda")
    l2 = torch.zeros(1, device="cuda")
    eng.lastlayer_full_accumulate(idx.cuda(), y.cuda(), Hl, l2)
    _, oHl = O.lastlayer_full_batch(om, idx.numpy(), y.numpy())
    assert rel(Hl.cpu().numpy(), oHl) < RTOL
    # diag(last-layer full) == last-layer part of diag (cross-structure consistency)
    assert rel(torch.diagonal(Hl).cpu().numpy(), oH[-p_ll:]) < RTOL
    eng.close()


def test_edge_cases_empty_graph_single_sample():
    """No edges at all (GCN: A = I after self loops; GraphSAGE: all rows isolated) and a 1-sample batch."""
    N, F, H, C = 50, 8, 16, 4
    for kind in ("gcn", "sage"):
        _, X, Ws, bs = _make(kind, N, F, H, C, 10, seed=1)
        ei = torch.zeros(2, 0, dtype=torch.int64)
        eng = _engine(kind, N, ei, X, Ws, bs, symmetric=False)
        assert eng.nnz == (N if kind == "gcn" else 0)
        idx, y = torch.tensor([7]), torch.tensor([2])
        views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 10)
        om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], False)
        _, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 10)
        for l, (A, B) in enumerate(views):
            assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL and rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL
        eng.close()


def test_bad_inputs_raise():
    import laplace_gnn_amd as lg

    N = 20
    ei = torch.tensor([[0, 1], [1, 25]])  # 25 >= N
    with pytest.raises(lg._lib.HipLibraryError, match="out of"):
        lg.GraphEngine(ei.cuda(), N)
    with pytest.raises(lg._lib.HipLibraryError, match="GPU"):
        lg.GraphEngine(torch.tensor([[0], [1]]), N)  # CPU tensor
    eng = lg.GraphEngine(torch.tensor([[0], [1]]).cuda(), N)
    with pytest.raises(ValueError):
        eng.bind(torch.randn(N, 4).cuda(), [torch.randn(3, 5).cuda()], [torch.randn(3).cuda()])
    with pytest.raises(NotImplementedError):
        eng.bind(torch.randn(N, 4).cuda(), [torch.randn(3, 4).cuda()], [torch.randn(3).cuda()], act="gelu")
    eng.close()


# ---- BASELINE size: arxiv-shaped -----------------------------------------------------------------------
@pytest.fixture(scope="module")
def arxiv():
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("arxiv", "cuda")
    torch.manual_seed(0)
    model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    return w, model, train_idx.cuda(), train_y.cuda(), X


def test_arxiv_shape_properties(arxiv):
    """Full-size run checked through properties that need no CPU reference:
    * fused and unfused kernel paths agree (two independent implementations of B_0);
    * A factors equal batches * X^T X / N_train (torch fp32 reference of the Gram kernel);
    * factors symmetric, positive semi-definite; loss equals torch's CE on the engine's logits;
    * one batch processed twice doubles B (linearity of the accumulate)."""
    w, model, idx, y, X = arxiv
    eng = model.engine
    M = w["batch"]
    b0, b1 = slice(0, M), slice(M, 2 * M)
    _, v_f, loss_f = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[b0], y[b0], w["n_train"], v_f, loss_f, fuse=True)
    _, v_u, loss_u = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[b0], y[b0], w["n_train"], v_u, loss_u, fuse=False)
    torch.cuda.synchronize()
    for (Af, Bf), (Au, Bu) in zip(v_f, v_u):
        assert rel(Bf.cpu().numpy(), Bu.cpu().numpy()) < 1e-5
        assert torch.equal(Af, Au)
        assert torch.equal(Bf, Bf.T)
        ev = torch.linalg.eigvalsh(Bf.double())
        assert float(ev.min()) > -1e-5 * float(ev.max())
    Xd = X.cuda().double()
    assert rel(v_f[0][0].cpu().numpy(), (Xd.T @ Xd / w["n_train"]).cpu().numpy()) < 1e-5
    logits = model(idx[b0])
    ce = torch.nn.functional.cross_entropy(logits.double(), y[b0], reduction="sum")
    assert abs(float(loss_f) - float(ce)) < 1e-5 * float(ce)
    # accumulate twice == 2x ; a different batch gives a different B but the same A increment
    eng.kfac_accumulate(idx[b0], y[b0], w["n_train"], v_f, loss_f, fuse=True)
    torch.cuda.synchronize()
    assert rel(v_f[0][1].cpu().numpy(), 2 * v_u[0][1].cpu().numpy()) < 1e-5
    _, v_2, l_2 = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[b1], y[b1], w["n_train"], v_2, l_2, fuse=True)
    torch.cuda.synchronize()
    assert torch.equal(v_2[0][0], v_u[0][0]) and rel(v_2[0][1].cpu().numpy(), v_u[0][1].cpu().numpy()) > 1e-3


def test_arxiv_shape_graphsage_properties():
    """The same sizes with GraphSAGE (the fused kernel's row-list variant: only the batch nodes and their neighbours are
    visited): fused and unfused paths agree on every factor, class ranges add up to the batch, factors symmetric PSD."""
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("arxiv_sage", "cuda")
    torch.manual_seed(0)
    model = lg.GraphSAGE(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    idx, y = train_idx.cuda()[: w["batch"]], train_y.cuda()[: w["batch"]]
    idx[-50:] = idx[:50]  # repeated node ids
    _, v_f, l_f = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], v_f, l_f, fuse=True)
    _, v_u, l_u = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], v_u, l_u, fuse=False)
    _, v_p, l_p = eng.new_kfac_buffers()
    for r in ((0, 7), (7, 40)):
        eng.kfac_accumulate(idx, y, w["n_train"], v_p, l_p, classes=r)
    torch.cuda.synchronize()
    for (Af, Bf), (Au, Bu), (Ap, Bp) in zip(v_f, v_u, v_p):
        assert rel(Bf.cpu().numpy(), Bu.cpu().numpy()) < 1e-5 and rel(Bp.cpu().numpy(), Bf.cpu().numpy()) < 1e-5
        assert torch.equal(Af, Au) and torch.equal(Bf, Bf.T)
        ev = torch.linalg.eigvalsh(Bf.double())
        assert float(ev.min()) > -1e-5 * float(ev.max())
    assert abs(float(l_f) - float(l_u)) < 1e-5 * abs(float(l_u))  # (one float atomic per workgroup: the order varies run to run)
    eng.check_async_errors()


def test_power_law_arxiv_shape_eight_emulated_ranks_sum_to_the_single_fit():
    """arxiv sizes with power-law degrees (hubs of thousands of neighbours: the long-row side kernel, the top layer's sliced
    hub rows) dealt to 8 emulated ranks as (batch, class-range) units: the ranks' flat buffers add up to the single-process
    fit -- what the RCCL all-reduce produces on a node -- on three batches."""
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("arxiv_powerlaw", "cuda")
    torch.manual_seed(0)
    model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    n = 3 * w["batch"] - 4000  # two full batches and a ragged one
    idx, y = train_idx.cuda()[:n], train_y.cuda()[:n]
    loader = lg.TensorBatchLoader(idx, y, w["batch"])
    flat1, views1, loss1 = eng.new_kfac_buffers()
    for xb, yb in loader:
        eng.kfac_accumulate(xb, yb, n, views1, loss1)
    assert eng.num_long_rows > 100
    total = None
    batches = list(loader)
    for rank in range(8):
        flat, views, loss = eng.new_kfac_buffers()
        for t, c0, c1 in lg.units_of_rank(len(batches), w["C"], rank, 8):
            eng.kfac_accumulate(batches[t][0], batches[t][1], n, views, loss, classes=(c0, c1))
        total = flat.clone() if total is None else total + flat
    torch.cuda.synchronize()
    assert rel(total.cpu().numpy(), flat1.cpu().numpy()) < 1e-5
    eng.check_async_errors()


def test_arxiv_shape_sampled_oracle_check(arxiv):
    """B_1 (40 x 40) of one full-size batch against the oracle's sparse backward restricted to the top
    layer (cheap on the CPU: no 256-wide planes)."""
    w, model, idx, y, X = arxiv
    eng = model.engine
    M = 2000
    _, v, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[:M], y[:M], w["n_train"], v, loss)
    torch.cuda.synchronize()
    logits = eng.forward_all().cpu().numpy()
    rows, cols, vals = [t.cpu().numpy() for t in eng.export_propagation()]
    import scipy.sparse as sp
    P = sp.csr_matrix((vals, (rows, cols)), shape=(w["N"], w["N"]))
    V = O.kfac_seeds(logits[idx[:M].cpu().numpy()])
    B1 = np.zeros((w["C"], w["C"]), np.float64)
    PT = P.T.tocsr()
    sel = sp.csr_matrix((np.ones(M, np.float32), (idx[:M].cpu().numpy(), np.arange(M))), shape=(w["N"], M))
    PTs = (PT @ sel).tocsr()
    for c in range(w["C"]):
        g1 = PTs @ V[:, :, c]
        B1 += g1.T.astype(np.float64) @ g1
    assert rel(v[1][1].cpu().numpy(), B1) < RTOL


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_class_ranges_are_exact_shares_and_units_cover_a_fit(kind):
    """lgnn_kfac_accumulate_classes: (a) disjoint class ranges of one batch add up to the whole batch;
    (b) emulating 8 ranks with data.units_of_rank on one GPU and summing their buffers reproduces the
    single-process fit (what the RCCL all-reduce does on a node)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 3000, 48, 64, 10, 12000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, seed=11)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(N, generator=g)[:700].cuda()
    y = torch.randint(0, C, (700,), generator=g).cuda()
    eng = _engine(kind, N, ei, X, Ws, bs)
    _, v_full, l_full = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[:300], y[:300], 700, v_full, l_full)
    _, v_part, l_part = eng.new_kfac_buffers()
    for r in ((0, 3), (3, 4), (4, 10)):
        eng.kfac_accumulate(idx[:300], y[:300], 700, v_part, l_part, classes=r)
    torch.cuda.synchronize()
    for (Af, Bf), (Ap, Bp) in zip(v_full, v_part):
        assert rel(Bp.cpu().numpy(), Bf.cpu().numpy()) < 1e-5 and rel(Ap.cpu().numpy(), Af.cpu().numpy()) < 1e-6
    assert abs(float(l_part) - float(l_full)) < 1e-5 * abs(float(l_full))
    with pytest.raises(lg._lib.HipLibraryError, match="class range"):
        eng.kfac_accumulate(idx[:10], y[:10], 700, v_part, l_part, classes=(4, 4))

    loader = lg.TensorBatchLoader(idx, y, 300)  # 300 / 300 / 100
    views1, loss1 = kfac_fit_engine(eng, idx, y, 300)
    flat_sum = None
    for rank in range(8):
        flat, views, loss = eng.new_kfac_buffers()
        batches = list(loader)
        for t, c0, c1 in lg.units_of_rank(len(loader), C, rank, 8):
            eng.kfac_accumulate(batches[t][0], batches[t][1], 700, views, loss, classes=(c0, c1))
        flat_sum = flat.clone() if flat_sum is None else flat_sum + flat
    torch.cuda.synchronize()
    ref = torch.cat([t.reshape(-1) for pair in views1 for t in pair]).cpu().numpy()
    assert rel(flat_sum[:-1].cpu().numpy(), ref) < 1e-5
    assert abs(float(flat_sum[-1]) - loss1) < 1e-5 * abs(loss1)
    eng.close()


@pytest.mark.parametrize("kind,H,C,L", [
    ("gcn", 256, 47, 2),   # fused256 + backgemm with odd K (scalar staging), 8 column tiles
    ("gcn", 256, 12, 3),   # fused256 with the store path (g needed by the layer below) + hidden->hidden GEMM (K = 256)
    ("gcn", 128, 9, 2),    # generic fused kernel <128>, backgemm with 4 column tiles
    ("gcn", 96, 5, 2),     # width not a multiple of 32
    ("gcn", 30, 4, 2),     # width % 4 != 0: unfused fallback (SpMM + Gram through HBM)
    ("gcn", 200, 70, 2),   # K > 64: generic GEMM instead of backgemm
    ("sage", 256, 10, 2),  # GraphSAGE through fused256 MODE 1: self rows by flag, mask bits at the destination
    ("sage", 256, 40, 2),  # the same at the arxiv widths (8 mask words, 40 planes)
    ("sage", 192, 9, 3),   # three layers: compact top level, then the general path with stored planes
    ("sage", 64, 33, 3),
])
def test_kfac_kernel_variants_vs_oracle(kind, H, C, L):
    """Every kernel-selection branch of the KFAC path (tile widths, K ranges, store / self / fallback paths)."""
    N, F, E = 1200, 40, 5000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=L, seed=21)
    g = torch.Generator().manual_seed(3)
    idx = torch.randperm(N, generator=g)[:260]
    y = torch.randint(0, C, (260,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs)
    eng.set_workspace_limit(64 << 20)  # forces several class chunks
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 130)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 130)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l}"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l}"
    assert abs(loss - float(oloss)) < RTOL * abs(float(oloss))
    eng.close()


@pytest.mark.parametrize("kind,H,L", [("gcn", 256, 2), ("sage", 256, 2), ("gcn", 192, 3), ("sage", 132, 3)])
def test_long_rows_take_the_side_kernel_vs_oracle(kind, H, L):
    """Hub rows (more than 64 stored entries, up to thousands) leave the 256-wide fused kernel's per-wave gather: a side
    kernel computes them with whole workgroups (rows longer than 1 024 entries split over several, float atomics) and the
    fused kernel fetches the finished rows -- MODE 0 and MODE 1, stored planes for three layers, class chunks."""
    N, F, C, E = 5000, 32, 6, 20000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=L, seed=41)
    g = torch.Generator().manual_seed(2)
    hubs = []  # hubs of 2 600 / 1 100 / 300 / 90 / 65 neighbours on top of the uniform background
    for hub, deg in ((7, 2600), (1234, 1100), (4999, 300), (42, 90), (3000, 65)):
        nb = torch.randperm(N, generator=g)[:deg]
        hubs.append(torch.stack([torch.full((deg,), hub), nb]))
    ei = torch.cat([ei] + hubs, dim=1)
    idx = torch.randperm(N, generator=g)[:600]
    y = torch.randint(0, C, (600,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs)
    eng.set_workspace_limit(96 << 20)
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 250)  # 250 / 250 / 100
    assert eng.num_long_rows >= 5, eng.num_long_rows
    rows, _ = eng.export_adj()
    assert int(torch.bincount(rows, minlength=N).max()) > 1024  # at least one row is split over several tasks
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 250)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l}"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l}"
    assert abs(loss - float(oloss)) < RTOL * abs(float(oloss))
    # a class range of one batch with repeated node ids (a data-parallel rank's share): the top layer's hub slices add their
    # partial tiles for the range's rows only
    sub = idx[:250].clone()
    sub[200:] = sub[:50]
    _, v2, l2 = eng.new_kfac_buffers()
    eng.kfac_accumulate(sub.cuda(), y[:250].cuda(), 600, v2, l2, classes=(1, 4))
    _, ok = O.kfac_batch(om, sub.numpy(), y[:250].numpy(), 600, True, (1, 4))
    for l, (A, B) in enumerate(v2):
        if kind == "sage" and l == L - 1:
            continue  # GraphSAGE: the share with class 0 carries the whole (cheap) top-layer Gram
        assert rel(B.cpu().numpy(), ok[2 * l][0]) < RTOL, f"class range, B_{l}"
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("kind", ["sage", "gcn"])
def test_large_plane_path_at_small_size_vs_oracle(kind, monkeypatch):
    """Planes beyond the fused kernel's 4 GiB addressing (products-shaped GraphSAGE KFAC: 5 GB) take SpMM + Gram through HBM
    with the per-batch masked values: entries whose value is zero are not gathered, and a deeper GraphSAGE model's second
    backward level is masked with the rows the level above can write.  LGNN_PLANE_LIMIT lowers the bound so that a 3-layer,
    256-wide model on 3 000 nodes takes exactly that path; results against the oracle."""
    monkeypatch.setenv("LGNN_PLANE_LIMIT", "1000000")
    N, F, H, C, E, L = 3000, 20, 256, 5, 9000, 3
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=L, seed=77)
    g = torch.Generator().manual_seed(9)
    idx = torch.randperm(N, generator=g)[:300]
    idx[280:] = idx[:20]
    y = torch.randint(0, C, (300,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs)
    plan = eng.kfac_plan()
    assert not any(plan["fused"][1:]), plan  # the backward steps left the fused path
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 128)  # 128 / 128 / 44
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 128)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l}"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l}"
    assert abs(loss - float(oloss)) < RTOL * abs(float(oloss))
    eng.check_async_errors()
    eng.close()


def _random_config(seed):
    r = np.random.default_rng(seed)
    kind = "gcn" if r.random() < 0.6 else "sage"
    L = int(r.choice([1, 2, 2, 2, 3]))
    N = int(r.integers(40, 900))
    F = int(r.integers(1, 70))
    H = int(r.choice([4, 8, 20, 32, 33, 64, 100, 128, 160, 256]))
    C = int(r.choice([1, 2, 3, 7, 10, 16, 17, 40, 47, 64, 65, 70]))
    E = int(r.integers(0, 6 * N))
    M = int(r.integers(1, min(N, 300) + 1))
    bs = int(r.choice([M, max(1, M // 2), max(1, M // 3 + 1)]))
    dup = r.random() < 0.3          # node ids repeated inside the batch
    sym = r.random() < 0.7
    ranges = r.random() < 0.4 and C > 1
    return kind, L, N, F, H, C, E, M, bs, dup, sym, ranges


@pytest.mark.parametrize("seed", range(24))
def test_kfac_random_configurations_vs_oracle(seed):
    """Randomised shapes through every dispatch of the KFAC path (C = 1 ... 70 incl. the > 64 fallback, widths that are
    not multiples of 4 / 32, empty edge lists, single-sample batches, repeated node ids, directed graphs, class-range
    shares): the HIP factors against the oracle's, fp32 <= 1e-4 relative per block."""
    kind, L, N, F, H, C, E, M, bs, dup, sym, ranges = _random_config(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    ei = torch.randint(0, N, (2, E), generator=g) if E > 0 else torch.zeros(2, 0, dtype=torch.int64)
    X = torch.randn(N, F, generator=g)
    mult = 2 if kind == "sage" else 1
    dims = [F] + [H] * (L - 1) + [C]
    Ws = [torch.randn(dims[l + 1], mult * dims[l], generator=g) / (mult * dims[l]) ** 0.5 for l in range(L)]
    bs_ = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(L)]
    idx = torch.randperm(N, generator=g)[:M]
    if dup and M > 2:
        idx[M // 2:] = idx[: M - M // 2].clone()
    y = torch.randint(0, C, (M,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs_, symmetric=sym)
    if ranges:  # the fit as a sum of class-range shares
        _, views, loss = eng.new_kfac_buffers()
        cut = max(1, C // 3)
        for s0 in range(0, M, bs):
            for r in ((0, cut), (cut, C)):
                eng.kfac_accumulate(idx[s0:s0 + bs].cuda(), y[s0:s0 + bs].cuda(), M, views, loss, classes=r)
        torch.cuda.synchronize()
        loss = float(loss)
    else:
        views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), bs)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs_], sym)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), bs)
    cfg = (kind, L, N, F, H, C, E, M, bs, dup, sym, ranges)

    def close(a, b):  # relative in the Frobenius norm; C = 1 has exactly zero B factors (softmax of one class)
        return np.linalg.norm(a - b) <= RTOL * np.linalg.norm(b) + 1e-9

    for l, (A, B) in enumerate(views):
        assert close(B.cpu().numpy(), oH[2 * l][0]), (f"B_{l}", cfg)
        assert close(A.cpu().numpy(), oH[2 * l][1]), (f"A_{l}", cfg)
    assert abs(loss - float(oloss)) <= RTOL * max(abs(float(oloss)), 1e-6), cfg
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("seed", range(100, 112))
def test_diag_lastlayer_jacobians_random_configurations_vs_oracle(seed):
    """The other accumulators and the Jacobians on the same randomised shapes: diagonal GGN (closed form, <= 2
    layers), last-layer full GGN, per-sample Jacobians."""
    kind, L, N, F, H, C, E, M, bs, dup, sym, _ = _random_config(seed)
    M = min(M, 60)
    g = torch.Generator().manual_seed(2000 + seed)
    ei = torch.randint(0, N, (2, E), generator=g) if E > 0 else torch.zeros(2, 0, dtype=torch.int64)
    X = torch.randn(N, F, generator=g)
    mult = 2 if kind == "sage" else 1
    dims = [F] + [H] * (L - 1) + [C]
    Ws = [torch.randn(dims[l + 1], mult * dims[l], generator=g) / (mult * dims[l]) ** 0.5 for l in range(L)]
    bs_ = [torch.randn(dims[l + 1], generator=g) * 0.1 for l in range(L)]
    idx = torch.randperm(N, generator=g)[:M]
    if dup and M > 2:
        idx[M // 2:] = idx[: M - M // 2].clone()
    y = torch.randint(0, C, (M,), generator=g)
    eng = _engine(kind, N, ei, X, Ws, bs_, symmetric=sym)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs_], sym)
    cfg = (kind, L, N, F, H, C, E, M, dup, sym)

    def close(a, b):
        return np.linalg.norm(a - b) <= RTOL * np.linalg.norm(b) + 1e-9

    Js, f = eng.jacobians(idx.cuda())
    oJ, of = O.jacobians_batch(om, idx.numpy())
    assert close(Js.cpu().numpy(), oJ) and close(f.cpu().numpy(), of), ("jacobians", cfg)
    if L <= 2:  # the closed form served that call; the generic plane route (what deeper models use) must agree
        os.environ["LGNN_JAC_PLANES"] = "1"
        try:
            Jp, _ = eng.jacobians(idx.cuda())
        finally:
            del os.environ["LGNN_JAC_PLANES"]
        assert close(Jp.cpu().numpy(), oJ), ("jacobians, plane route", cfg)
    if L <= 2 and kind == "gcn":
        Hd = torch.zeros(eng.n_params, device="cuda")
        loss = torch.zeros(1, device="cuda")
        eng.diag_accumulate(idx.cuda(), y.cuda(), Hd, loss)
        ol, od = O.diag_batch(om, idx.numpy(), y.numpy())
        assert close(Hd.cpu().numpy(), od) and abs(float(loss) - float(ol)) <= RTOL * max(abs(float(ol)), 1e-6), ("diag", cfg)
    p_ll = eng.in_dims[-1] * C + C
    if p_ll <= 6000:
        Hl = torch.zeros(p_ll, p_ll, device="cuda")
        loss = torch.zeros(1, device="cuda")
        if seed % 3 == 0:
            eng.set_workspace_limit(1 << 20)  # the minimum: several chunks of class pairs for the larger heads
        eng.lastlayer_full_accumulate(idx.cuda(), y.cuda(), Hl, loss)
        ol, oh = O.lastlayer_full_batch(om, idx.numpy(), y.numpy())
        assert close(Hl.cpu().numpy(), oh), ("last layer", cfg)
    eng.check_async_errors()
    eng.close()


def test_glm_predictive_on_every_node_of_the_arxiv_shape(arxiv):
    """la(x) -- the reference's default GLM predictive with the probit link -- for ALL 169 343 nodes of the arxiv-shaped
    model: the Jacobian route would need M * C * P = 1.2 PB; the matrix-free route (csrc/predictive.hip) is checked
    against the Jacobian route on 24 of the nodes and must give a valid distribution everywhere."""
    import time

    import laplace_gnn_amd as lg

    w, model, idx, y, X = arxiv
    loader = lg.TensorBatchLoader(idx, y, batch_size=w["batch"])
    la = lg.KronLaplace(model, "classification", prior_precision=2.0)
    la.fit(loader)
    every = torch.arange(w["N"], device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probs = la(every, link_approx="probit")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert probs.shape == (w["N"], w["C"]) and bool(torch.isfinite(probs).all())
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-4 and float(probs.min()) >= 0
    sub = torch.randperm(w["N"], generator=torch.Generator().manual_seed(0))[:24].cuda()
    Js, f = la.backend.jacobians(sub)
    ref = torch.diagonal(la.functional_variance(Js), dim1=1, dim2=2)
    f_mu, f_vd = la._glm_variance_matrix_free(sub)
    assert rel(f_vd.cpu().numpy(), ref.cpu().numpy()) < 1e-4 and rel(f_mu.cpu().numpy(), f.cpu().numpy()) < 1e-6
    print(f"GLM predictive of {w['N']} nodes: {dt * 1e3:.1f} ms")
    assert dt < 30.0
    # the Laplace bridge on every node: 2 C + 1 mapped outputs per node instead of the C x C covariance (lgnn_glm_variance_mapped)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bridge = la(every, link_approx="bridge_norm")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bridge.shape == (w["N"], w["C"]) and bool(torch.isfinite(bridge).all())
    assert float((bridge.sum(-1) - 1).abs().max()) < 1e-4
    fv = la.functional_variance(Js)
    _, diag, rows, total = la._bridge_moments_matrix_free(sub)
    assert rel(diag.cpu().numpy(), ref.cpu().numpy()) < 1e-4
    assert rel(rows.cpu().numpy(), fv.sum(-1).cpu().numpy()) < 1e-4 and rel(total.cpu().numpy(), fv.sum((1, 2)).cpu().numpy()) < 1e-4
    assert rel(bridge[sub].cpu().numpy(), la(sub, link_approx="bridge_norm").cpu().numpy()) < 1e-5
    print(f"Laplace bridge of {w['N']} nodes: {dt * 1e3:.1f} ms")
    assert dt < 60.0
    # link_approx="mc" over the full covariance of 3 000 nodes (their Jacobians would be 20 GB): 820 polarised variances per node
    many = torch.randperm(w["N"], generator=torch.Generator().manual_seed(3))[:3000].cuda()
    many[:24] = sub
    eps = torch.randn(w["C"], 20, generator=torch.Generator().manual_seed(4)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mc = la(many, link_approx="mc", n_samples=20, eps=eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert mc.shape == (3000, w["C"]) and bool(torch.isfinite(mc).all()) and float((mc.sum(-1) - 1).abs().max()) < 1e-4
    ref_mc = la._glm_predictive_samples(f, fv, 20, False, None, eps).mean(dim=0)  # the 24 nodes through their Jacobians
    assert float((mc[:24] - ref_mc).abs().max()) < 1e-4
    print(f"mc link (full covariance) of 3000 nodes: {dt * 1e3:.1f} ms")
    assert dt < 120.0


# ---- independent reference of the dominant kernel's output at the headline shape -------------------
def _fp64_kfac_classes(kind, eng, idx, y, classes, X, Ws, bs):
    """fp64 torch restatement, on the GPU, of one batch's KFAC backward passes for the class columns ``classes`` at FULL size,
    built from the engine's exported propagation matrix and the oracle's fork-exact seeds (curvlinops/kfac.py:637-661,
    777-817; gnn/models/layers.py:26-29, 45-46): B_0, B_1 and the loss.  Nothing of the HIP curvature path is involved --
    only ``export_propagation`` (bit exact vs the reference, tests/test_gpu_parity.py) and ``forward_all`` for the logits,
    which is itself checked against this restatement's own fp64 forward."""
    from test_gpu_baseline_shapes import _prop_csr, _torch_forward

    N = X.shape[0]
    P = _prop_csr(eng, N)
    PT = P.t().to_sparse_csr()
    Xd = X.cuda().double()
    W0, W1 = Ws[0].double(), Ws[1].double()
    logits64, _ = _torch_forward(kind, P, Xd, Ws, bs)
    logits = eng.forward_all()
    assert rel(logits.cpu().numpy(), logits64.cpu().numpy()) < 1e-5
    # pre-activation sign of the hidden layer from the fp64 forward
    if kind == "gcn":
        pre = torch.sparse.mm(P, Xd @ W0.T + bs[0].double())
    else:
        pre = torch.cat([Xd, torch.sparse.mm(P, Xd)], dim=1) @ W0.T + bs[0].double()
    mask = (pre > 0).double()
    f = logits64[idx]
    V = torch.from_numpy(O.kfac_seeds(f.float().cpu().numpy())).cuda().double()  # [M, k, c] from the oracle's closed form
    C, H = W1.shape[0], W0.shape[0]
    B0 = torch.zeros(H, H, dtype=torch.float64, device="cuda")
    B1 = torch.zeros(C, C, dtype=torch.float64, device="cuda")
    for c in classes:
        G = torch.zeros(N, C, dtype=torch.float64, device="cuda")
        G.index_add_(0, idx, V[:, :, c])
        if kind == "gcn":
            g1 = torch.sparse.mm(PT, G)
            g0 = torch.sparse.mm(PT, (g1 @ W1) * mask)
        else:
            g1 = G
            dcat = g1 @ W1
            g0 = mask * (dcat[:, :H] + torch.sparse.mm(PT, dcat[:, H:]))
        B1 += g1.T @ g1
        B0 += g0.T @ g0
    ce = torch.nn.functional.cross_entropy(f, y, reduction="sum")
    return B0, B1, float(ce)


@pytest.mark.parametrize("workload", ["arxiv", "arxiv_powerlaw", "arxiv_sage"])
def test_headline_shape_B0_against_an_independent_fp64_restatement(workload):
    """The 256-wide fused SpMM^T -> Gram kernel at N = 169 343 with one full batch of 10 000: ``lgnn_kfac_accumulate_classes``
    for three class columns against fp64 torch sparse algebra (B_0 256 x 256, B_1 40 x 40, loss), uniform and power-law
    graph, GCN and GraphSAGE.  fp32 <= 1e-4 relative Frobenius error (BASELINE.json north_star)."""
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload(workload, "cuda")
    kind = w.get("kind", "gcn")
    torch.manual_seed(0)
    cls = lg.GraphSAGE if kind == "sage" else lg.GCN
    model = cls(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    M = w["batch"]
    idx, y = train_idx.cuda()[:M].clone(), train_y.cuda()[:M]
    idx[-20:] = idx[:20]  # repeated node ids accumulate like x[x_indices]'s backward
    Ws = [c.lin.weight.detach() for c in model.convs]
    bs = [c.lin.bias.detach() for c in model.convs]
    classes = (0, 17, 39)
    B0, B1, ce = _fp64_kfac_classes(kind, eng, idx, y, classes, X, Ws, bs)
    _, views, loss = eng.new_kfac_buffers()
    for c in classes:
        eng.kfac_accumulate(idx, y, w["n_train"], views, loss, classes=(c, c + 1))
    torch.cuda.synchronize()
    if kind == "sage":
        # GraphSAGE: the whole top-layer Gram (M * C rows of the seeds) travels with the share that holds class 0
        Vall = torch.from_numpy(O.kfac_seeds(eng.forward(idx).cpu().numpy())).cuda().double()
        G = torch.zeros(w["N"], w["C"], w["C"], dtype=torch.float64, device="cuda")
        G.index_add_(0, idx, Vall)  # [node, k, c]: duplicated ids accumulate
        B1 = torch.einsum("nkc,nlc->kl", G, G)
    assert rel(views[0][1].cpu().numpy(), B0.cpu().numpy()) < RTOL, "B_0"
    assert rel(views[1][1].cpu().numpy(), B1.cpu().numpy()) < RTOL, "B_1"
    assert abs(float(loss) - ce) < RTOL * ce
    eng.check_async_errors()
    eng.close()


def test_headline_launch_of_all_40_classes_against_the_fp64_restatement():
    """the headline's ONE launch of ``paths_fused_kernel`` over all 40 class columns of a full batch of
    10 000 at N = 169 343 (what ``bench.py`` times) against the fp64 restatement of every class column -- not three single-class
    calls, and not the plane route (HIP vs HIP).  B_0 256 x 256, B_1 40 x 40, the loss."""
    import bench
    import laplace_gnn_amd as lg

    w, ei, X, train_idx, train_y = bench.make_workload("arxiv", "cuda")
    torch.manual_seed(0)
    model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    M = w["batch"]
    idx, y = train_idx.cuda()[:M].clone(), train_y.cuda()[:M]
    idx[-20:] = idx[:20]
    Ws = [c.lin.weight.detach() for c in model.convs]
    bs = [c.lin.bias.detach() for c in model.convs]
    B0, B1, ce = _fp64_kfac_classes("gcn", eng, idx, y, range(w["C"]), X, Ws, bs)
    _, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], views, loss)  # all classes, one call: one launch of the fused kernel
    torch.cuda.synchronize()
    assert eng.last_kfac_used_paths
    assert rel(views[0][1].cpu().numpy(), B0.cpu().numpy()) < RTOL, "B_0"
    assert rel(views[1][1].cpu().numpy(), B1.cpu().numpy()) < RTOL, "B_1"
    assert abs(float(loss) - ce) < RTOL * ce
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("kind,structure", [("gcn", "kron"), ("sage", "kron"), ("gcn", "diag")])
def test_no_device_allocation_after_the_first_fit(kind, structure):
    """the workspace is sized by the first fit; a second fit of the same shape allocates nothing
    (``lgnn_device_bytes`` before / after), whichever route the batches take (two-hop / one-hop paths, closed-form diagonal)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 6000, 64, 256 if structure == "kron" else 64, 12, 30000
    g = torch.Generator().manual_seed(11)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    torch.manual_seed(3)
    model = (lg.GCN if kind == "gcn" else lg.GraphSAGE)(F, H, C, 2, X, ei, symmetric=True).to("cuda").eval()
    idx, y = torch.randperm(N, generator=g)[:2500].cuda(), torch.randint(0, C, (2500,), generator=g).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=1000)  # 1000 / 1000 / 500
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure=structure)
    la.fit(loader)
    torch.cuda.synchronize()
    before = model.engine.device_bytes()
    for _ in range(2):
        model.engine.invalidate()
        la.fit(loader)
    torch.cuda.synchronize()
    assert model.engine.device_bytes() == before
    if structure == "kron":
        assert model.engine.last_kfac_used_paths
