"""The two-hop path route of the 2-layer GCN KFAC (csrc/paths.hip): B_0 without class planes, from the
diagonal + rank-2 structure of the seed blocks, against the CPU oracle (pinned to the reference, incl. the identity itself:
tests/test_oracle_golden.py::test_first_layer_B_from_two_hop_paths_...) and against the class-plane route of the same
library.  fp32 <= 1e-4 relative Frobenius error per block (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
from gpu_utils import kfac_fit_engine, oracle_from_arrays, rel
from test_gpu_scale import _engine, _make

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.mark.parametrize("H,C,skew,fork_exact", [
    (256, 40, False, True), (256, 40, True, True), (192, 7, False, True), (132, 33, True, True), (256, 64, False, True),
    (256, 1, False, True), (160, 10, True, False), (256, 32, False, False)])
def test_path_route_vs_oracle_and_plane_route(H, C, skew, fork_exact):
    N, F, E = 3000, 48, 12000
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, L=2, seed=H + C, skew=skew)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(N, generator=g)[:700]
    idx[7] = idx[11]
    idx[650:] = idx[100:150]  # node ids listed twice accumulate (x[x_indices] backward)
    y = torch.randint(0, C, (700,), generator=g)
    eng = _engine("gcn", N, ei, X, Ws, bs)
    assert eng.kfac_plan(paths=True)["paths"] and not eng.kfac_plan(paths=False)["paths"]
    flat, views, loss = eng.new_kfac_buffers()
    flat2, views2, loss2 = eng.new_kfac_buffers()
    for s in range(0, 700, 300):  # 300 / 300 / 100
        eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, views, loss, fork_exact=fork_exact, paths=True)
        eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, views2, loss2, fork_exact=fork_exact, paths=False)
    torch.cuda.synchronize()
    om = oracle_from_arrays("gcn", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 300, fork_exact)
    for l, (A, B) in enumerate(views):
        if C == 1:  # one class: the seed is exactly zero (p = 1); the path route leaves fp32 cancellation residue, not zeros
            assert float(B.abs().max()) < 1e-10 and float(np.abs(oH[2 * l][0]).max()) < 1e-10
        else:
            assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l} vs oracle"
            assert rel(B.cpu().numpy(), views2[l][1].cpu().numpy()) < 2e-5, f"B_{l} vs the plane route"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l} vs oracle"
        assert torch.equal(B, B.T)
    assert abs(float(loss) - float(oloss)) <= RTOL * abs(float(oloss))
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("H,C", [(256, 40), (192, 64)])
def test_path_route_class_ranges_and_small_workspace(H, C):
    """Class ranges (the multi-GPU units) are exact shares on the path route too -- ranges of one class, of less and of more
    than one 32-class MFMA tile, ranges that start in the second tile -- and a workspace cap below one batch's Y cuts the
    classes into chunks."""
    N, F, E = 2000, 32, 8000
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, L=2, seed=3)
    g = torch.Generator().manual_seed(6)
    idx = torch.randperm(N, generator=g)[:400].cuda()
    y = torch.randint(0, C, (400,), generator=g).cuda()
    eng = _engine("gcn", N, ei, X, Ws, bs)
    flat, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, 400, views, loss, paths=True)
    flat2, v2, l2 = eng.new_kfac_buffers()
    cuts = [0, 1, 20, 33, 36, C]
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.kfac_accumulate(idx, y, 400, v2, l2, classes=(a, b), paths=True)
    torch.cuda.synchronize()
    assert rel(flat2.cpu().numpy(), flat.cpu().numpy()) < 1e-5
    eng.set_workspace_limit(N * H * 4 * 3 + 1024)  # three classes of Y per chunk
    assert eng.kfac_plan(paths=True)["classes_per_chunk"] == 3
    flat3, v3, l3 = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, 400, v3, l3, paths=True)
    torch.cuda.synchronize()
    assert rel(flat3.cpu().numpy(), flat.cpu().numpy()) < 1e-5
    eng.check_async_errors()
    eng.close()


def test_path_route_regression_and_empty_neighbourhoods():
    """Regression likelihood (V = sqrt(2) I: only the diagonal term survives) and a graph whose nodes are mostly far from the
    batch (their Y rows are zero but still written: the streaming Gram reads every row)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 2500, 16, 256, 3, 1500  # sparse: most nodes have no 2-hop path to the 40 batch nodes
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, L=2, seed=9)
    g = torch.Generator().manual_seed(2)
    idx = torch.randperm(N, generator=g)[:40]
    yr = torch.randn(40, C, generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind="gcn", symmetric=True)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs], likelihood="regression")
    om = oracle_from_arrays("gcn", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    for paths in (True, False):
        _, views, loss = eng.new_kfac_buffers()
        eng.kfac_accumulate(idx.cuda(), yr.cuda(), 40, views, loss, paths=paths)
        torch.cuda.synchronize()
        ol, okf = O.kfac_batch(om, idx.numpy(), yr.numpy(), 40, likelihood="regression")
        # the oracle applied the interface's factor 0.5 (sqrt(.5) per factor of a weight block); the engine returns raw factors
        for l, (A, B) in enumerate(views):
            assert rel(B.cpu().numpy() * np.sqrt(0.5), okf[2 * l][0]) < RTOL, (paths, l)
    yc = torch.randint(0, C, (40,), generator=g)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    _, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx.cuda(), yc.cuda(), 40, views, loss, paths=True)
    torch.cuda.synchronize()
    ol, okf = O.kfac_batch(om, idx.numpy(), yc.numpy(), 40)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), okf[2 * l][0]) < RTOL, l
    eng.check_async_errors()
    eng.close()


def test_streaming_gram_kernel_through_the_path_route_on_a_hub_graph():
    """A star-like graph: one node adjacent to a third of all nodes (thousands of paths: several 128-path windows and more
    than one 512-entry row chunk in ybuild_kernel), batch = everything."""
    N, F, H, C = 1800, 12, 256, 5
    g = torch.Generator().manual_seed(4)
    hub = torch.stack([torch.zeros(600, dtype=torch.int64), torch.randperm(N, generator=g)[:600]])
    ei = torch.cat([hub, torch.randint(0, N, (2, 2500), generator=g)], 1)
    _, X, Ws, bs = _make("gcn", N, F, H, C, 10, L=2, seed=8)
    idx = torch.arange(N)
    y = torch.randint(0, C, (N,), generator=g)
    eng = _engine("gcn", N, ei, X, Ws, bs)
    views, loss = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 1000)
    om = oracle_from_arrays("gcn", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 1000)
    assert eng.kfac_plan()["paths"]
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l}"
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("H,C", [(256, 40), (132, 64)])
def test_path_list_overflow_takes_the_enumerating_route(H, C, monkeypatch):
    """A batch whose path list does not fit its buffer (very large batches on hub-heavy graphs) is handled on the device, no
    host round trip: the fused kernel returns at once, the enumerating Y builder and the streaming Gram take over.  Forced
    here with LGNN_PATH_LIST_CAP=1000 (the batch has ~10^5 paths); same results."""
    N, F, E = 2500, 24, 10000
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, L=2, seed=12, skew=True)
    g = torch.Generator().manual_seed(8)
    idx = torch.randperm(N, generator=g)[:600].cuda()
    y = torch.randint(0, C, (600,), generator=g).cuda()
    eng = _engine("gcn", N, ei, X, Ws, bs)
    flat, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, 600, views, loss, paths=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("LGNN_PATH_LIST_CAP", "1000")
    flat2, v2, l2 = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, 600, v2, l2, paths=True)
    eng.kfac_accumulate(idx, y, 600, v2, l2, paths=True, classes=(0, 3))  # class ranges on that route too
    torch.cuda.synchronize()
    flat3, v3, l3 = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, 600, v3, l3, paths=False, classes=(0, 3))
    torch.cuda.synchronize()
    monkeypatch.delenv("LGNN_PATH_LIST_CAP")
    assert rel((flat2 - flat3).cpu().numpy(), flat.cpu().numpy()) < 2e-5
    eng.check_async_errors()
    eng.close()



@pytest.mark.parametrize("H,C,skew,fork_exact,sym", [
    (256, 40, False, True, True), (256, 40, True, True, False), (192, 7, False, True, True), (132, 33, True, False, True),
    (256, 64, False, True, False), (160, 10, True, False, False)])
def test_graphsage_one_hop_path_route_vs_oracle_and_plane_route(H, C, skew, fork_exact, sym):
    """GraphSAGE: the same fused kernel over one-hop paths -- mask at the destination, the neighbour half
    of W_1 as the kernel's operand, the node's own terms as pseudo paths (one-hot coefficient rows for its alpha term)."""
    import laplace_gnn_amd as lg

    N, F, E = 3000, 48, 12000
    ei, X, Ws, bs = _make("sage", N, F, H, C, E, L=2, seed=H + C, skew=skew)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(N, generator=g)[:700]
    idx[7] = idx[11]
    idx[650:] = idx[100:150]  # node ids listed twice accumulate (x[x_indices] backward)
    y = torch.randint(0, C, (700,), generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind="sage", symmetric=sym)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    assert eng.kfac_plan()["paths"] and not eng.kfac_plan(paths=False)["paths"]
    flat, views, loss = eng.new_kfac_buffers()
    flat2, views2, loss2 = eng.new_kfac_buffers()
    for s in range(0, 700, 300):  # 300 / 300 / 100
        eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, views, loss, fork_exact=fork_exact)
        assert eng.last_kfac_used_paths
        eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, views2, loss2, fork_exact=fork_exact, paths=False)
        assert not eng.last_kfac_used_paths
    torch.cuda.synchronize()
    om = oracle_from_arrays("sage", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], sym)
    oloss, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 300, fork_exact)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), oH[2 * l][0]) < RTOL, f"B_{l} vs oracle"
        assert rel(B.cpu().numpy(), views2[l][1].cpu().numpy()) < 2e-5, f"B_{l} vs the plane route"
        assert rel(A.cpu().numpy(), oH[2 * l][1]) < RTOL, f"A_{l} vs oracle"
        assert torch.equal(B, B.T)
    assert abs(float(loss) - float(oloss)) <= RTOL * abs(float(oloss))
    # class ranges (the multi-GPU units) are exact shares on this route too
    flat3, v3, l3 = eng.new_kfac_buffers()
    cuts = [0, 1, min(20, C - 1), C] if C > 2 else [0, C]
    for s in range(0, 700, 300):
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 700, v3, l3, classes=(a, b), fork_exact=fork_exact)
    torch.cuda.synchronize()
    assert rel(flat3.cpu().numpy(), flat.cpu().numpy()) < 1e-5
    eng.check_async_errors()
    eng.close()


def test_graphsage_path_route_regression_and_isolated_nodes():
    """Regression likelihood (V = sqrt(2) I: the one-hot alpha paths carry everything) and nodes without neighbours (their
    own terms only; rows without neighbours keep mean_agg's divisor 1)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 1500, 16, 256, 3, 900  # sparse: many isolated nodes, most nodes far from the 60 batch nodes
    ei, X, Ws, bs = _make("sage", N, F, H, C, E, L=2, seed=9)
    g = torch.Generator().manual_seed(2)
    idx = torch.randperm(N, generator=g)[:60]
    yr = torch.randn(60, C, generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind="sage", symmetric=True)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs], likelihood="regression")
    om = oracle_from_arrays("sage", N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    for paths in (None, False):
        _, views, loss = eng.new_kfac_buffers()
        eng.kfac_accumulate(idx.cuda(), yr.cuda(), 60, views, loss, paths=paths)
        assert eng.last_kfac_used_paths == (paths is None)
        torch.cuda.synchronize()
        ol, okf = O.kfac_batch(om, idx.numpy(), yr.numpy(), 60, likelihood="regression")
        for l, (A, B) in enumerate(views):
            assert rel(B.cpu().numpy() * np.sqrt(0.5), okf[2 * l][0]) < RTOL, (paths, l)
    eng.check_async_errors()
    eng.close()


@pytest.mark.parametrize("kind,paths", [("gcn", None), ("sage", None), ("gcn", False), ("sage", False)])
def test_shares_of_a_batch_add_up_whichever_way_the_route_cuts_it(kind, paths):
    """lgnn_kfac_accumulate_share: the multi-GPU unit.  On the path routes a part is a range of destination nodes (B_0 is a sum
    over nodes: no rank repeats the path products), otherwise a class range; either way the parts of a batch add up to the
    whole batch, part 0 carries the loss and the A increment, and uneven part counts (more parts than classes) are fine."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 3000, 40, 256, 10, 12000
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, L=2, seed=17)
    g = torch.Generator().manual_seed(5)
    idx = torch.randperm(N, generator=g)[:600]
    idx[590:] = idx[:10]
    y = torch.randint(0, C, (600,), generator=g)
    eng = lg.GraphEngine(ei.cuda(), N, kind=kind, symmetric=True)
    eng.bind(X.cuda(), [w.cuda() for w in Ws], [b.cuda() for b in bs])
    flat, views, loss = eng.new_kfac_buffers()
    for s in (0, 300):
        eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 600, views, loss, paths=paths)
    used_paths = eng.last_kfac_used_paths
    assert used_paths == (paths is None)
    for count, cuts in ((C, [0, 1, 4, 9, C]), (7, [0, 3, 7]), (25, [0, 2, 11, 12, 25])):  # 25 parts of 10 classes: empty class ranges
        flat2, v2, l2 = eng.new_kfac_buffers()
        for s in (0, 300):
            for a, b in zip(cuts[:-1], cuts[1:]):
                eng.kfac_accumulate(idx[s:s + 300].cuda(), y[s:s + 300].cuda(), 600, v2, l2, share=(a, b, count), paths=paths)
                assert eng.last_kfac_used_paths == used_paths
        torch.cuda.synchronize()
        assert rel(flat2.cpu().numpy(), flat.cpu().numpy()) < 1e-5, (count, cuts)
    # a single part that is not part 0 adds neither loss nor A
    flat3, v3, l3 = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx[:300].cuda(), y[:300].cuda(), 600, v3, l3, share=(3, 5, C), paths=paths)
    torch.cuda.synchronize()
    assert float(l3) == 0.0 and all(float(A.abs().max()) == 0.0 for A, _ in v3) and float(v3[0][1].abs().max()) > 0
    eng.check_async_errors()
    eng.close()
