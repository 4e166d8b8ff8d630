"""GPU parity of the launch-bound small-graph route (BASELINE.json configs[1] is its shape: Cora, diag GGN): the 2-layer GCN
forward through the cached P X (csrc/gcn2_forward.hip), the one-launch batch prologue and the LDS-DMA first-layer kernel
(csrc/diag.hip), against the CPU oracle -- which is pinned to the reference's ``GGNInterface.diag``
(laplace/curvature/curvature.py:412-432) on the golden fixtures (tests/test_oracle_golden.py).  Cases the goldens are too
small for: slabs longer than one resolve pass (> 512 entries), hub rows, a ragged last slab, duplicated and (flagged)
out-of-range node ids, widths that do not fill a tile.  fp32 <= 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
from gpu_utils import oracle_from_arrays, rel
from test_gpu_scale import _engine, _make

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _diag(eng, idx, y):
    Hd = torch.zeros(eng.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.diag_accumulate(idx.cuda(), y.cuda(), Hd, loss)
    torch.cuda.synchronize()
    return Hd.cpu().numpy(), float(loss.item())


@pytest.mark.parametrize("kind,F,H,C,E,skew,slab", [
    ("gcn", 100, 48, 5, 15000, False, "64"),   # ~21 entries per sample: 64-sample slabs need three resolve passes
    ("gcn", 100, 48, 5, 15000, False, None),   # the launcher's own slab length
    ("gcn", 67, 64, 7, 6000, True, "64"),      # hubs: one sample alone exceeds a pass; F + 1 = 68 spills into a second tile
    ("sage", 40, 24, 3, 12000, False, "64"),   # self path (three q rows per closing entry), 3-slot ring
    ("sage", 33, 64, 6, 5000, True, "17"),     # ragged slabs
    ("gcn", 130, 20, 4, 4000, False, "1"),     # one sample per workgroup
])
def test_small_graph_diag_route_vs_oracle(kind, F, H, C, E, skew, slab, monkeypatch):
    if slab is not None:
        monkeypatch.setenv("LGNN_DIAG_SLAB", slab)
    N, M = 1500, 333
    ei, X, Ws, bs = _make(kind, N, F, H, C, E, seed=13, skew=skew)
    eng = _engine(kind, N, ei, X, Ws, bs)
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), [w.numpy() for w in Ws], [b.numpy() for b in bs], True)
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, N, (M,), generator=g)  # with replacement: duplicated ids are separate samples
    if skew:  # make sure the heaviest rows are in the batch
        deg = torch.bincount(ei.flatten(), minlength=N)
        idx[:4] = torch.topk(deg, 4).indices
    y = torch.randint(0, C, (M,), generator=g)
    assert rel(eng.forward_all().cpu().numpy(), O.forward_all(om)[0]) < RTOL
    Hd, loss = _diag(eng, idx, y)
    ol, od = O.diag_batch(om, idx.numpy(), y.numpy())
    assert rel(Hd, od) < RTOL
    assert abs(loss - float(ol)) < RTOL * abs(float(ol))
    # every first-layer kernel computes the same numbers: the default (matrix cores for a GCN, the register-staged kernel for
    # GraphSAGE), the LDS-DMA / v_readlane kernel and the register-staged kernel of rounds 1-2
    for switch in ("LGNN_DIAG_VALU", "LGNN_DIAG_STAGED"):
        monkeypatch.setenv(switch, "1")
        Hs, _ = _diag(eng, idx, y)
        assert rel(Hs, od) < RTOL and rel(Hs, Hd) < 1e-5, switch
        monkeypatch.delenv(switch)
    eng.check_async_errors()
    eng.close()


def test_small_forward_matches_the_standard_forward_and_feeds_kfac(monkeypatch):
    """The pass through P X and the standard pass (P (X W^T + b)) agree to fp32 rounding, and a KFAC accumulate that finds the
    small route's cache (activations, ReLU bit masks) gives the standard route's factors."""
    from gpu_utils import kfac_fit_engine

    N, F, H, C, E = 900, 143, 64, 7, 4000  # F % 4 != 0: the padded copy of X feeds P X
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, seed=21)
    g = torch.Generator().manual_seed(1)
    idx, y = torch.randperm(N, generator=g)[:300], torch.randint(0, C, (300,), generator=g)
    eng = _engine("gcn", N, ei, X, Ws, bs)
    out_small = eng.forward_all().clone()
    v_small, l_small = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 128)
    v_small = [(A.clone(), B.clone()) for A, B in v_small]
    eng.close()
    monkeypatch.setenv("LGNN_NO_SMALL_FORWARD", "1")
    eng = _engine("gcn", N, ei, X, Ws, bs)
    out_std = eng.forward_all().clone()
    v_std, l_std = kfac_fit_engine(eng, idx.cuda(), y.cuda(), 128)
    assert rel(out_small.cpu().numpy(), out_std.cpu().numpy()) < 1e-5
    for (A, B), (A2, B2) in zip(v_small, v_std):
        assert rel(A.cpu().numpy(), A2.cpu().numpy()) < 1e-5 and rel(B.cpu().numpy(), B2.cpu().numpy()) < 1e-5
    assert abs(l_small - l_std) < 1e-5 * abs(l_std)
    eng.close()


def test_out_of_range_ids_and_labels_are_flagged_by_the_one_launch_prologue():
    from laplace_gnn_amd._lib import HipLibraryError

    N, F, H, C, E = 400, 16, 16, 3, 1500
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, seed=5)
    eng = _engine("gcn", N, ei, X, Ws, bs)
    idx = torch.arange(50)
    y = torch.zeros(50, dtype=torch.int64)
    bad_idx = idx.clone(); bad_idx[7] = N + 3
    _diag(eng, bad_idx, y)
    with pytest.raises(HipLibraryError, match="node index"):
        eng.check_async_errors()
    bad_y = y.clone(); bad_y[9] = C
    _diag(eng, idx, bad_y)
    with pytest.raises(HipLibraryError, match="label"):
        eng.check_async_errors()
    Hd, _ = _diag(eng, idx, y)  # the flags are cleared by the check: a clean batch passes
    eng.check_async_errors()
    assert np.isfinite(Hd).all()
    eng.close()


def test_diag_laplace_fit_is_launch_lean_and_matches_the_returning_backend():
    """DiagLaplace.fit on the HIP backend accumulates in place ([H | loss] in one buffer); the (loss, H)-returning
    ``backend.diag`` the reference interface defines gives the same posterior."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 600, 50, 32, 4, 2500
    ei, X, Ws, bs = _make("gcn", N, F, H, C, E, seed=8)
    torch.manual_seed(0)
    model = lg.GCN(F, H, C, 2, X.cuda(), ei.cuda(), symmetric=True).cuda()
    g = torch.Generator().manual_seed(2)
    idx, y = torch.randperm(N, generator=g)[:250].cuda(), torch.randint(0, C, (250,), generator=g).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=100)
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure="diag")
    la.fit(loader)
    assert la._inplace_backend()
    H1, l1 = la.H.clone(), float(la.loss)
    Hsum, lsum = torch.zeros_like(H1), 0.0
    for Xb, yb in loader:
        lb, Hb = la.backend.diag(Xb, yb, N=250)
        Hsum += Hb
        lsum += float(lb)
    assert rel(H1.cpu().numpy(), Hsum.cpu().numpy()) < 1e-5 and abs(l1 - lsum) < 1e-5 * abs(lsum)
    # override=False adds a second pass: H and loss double, and the first fit's loss is not overwritten on the way
    la.fit(loader, override=False)
    assert rel(la.H.cpu().numpy(), 2 * H1.cpu().numpy()) < 1e-5 and abs(float(la.loss) - 2 * l1) < 1e-5 * abs(l1)
    assert la.n_data == 500
