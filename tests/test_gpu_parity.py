"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against (a) the golden
vectors generated from the reference and (b) the CPU oracle on seeded inputs.
Tolerances: integer / index paths bit exact; fp32 factors <= 1e-4 relative Frobenius error per block
(BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

import gnn_laplace_oracle as O
from conftest import GOLDEN
from golden_utils import conv_offset
from gpu_utils import engine_from_golden, kfac_fit_engine, oracle_from_arrays, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "c1_" not in p and "steloop_" not in p)
IDS = [os.path.basename(p)[:-4] for p in CASES]


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_graph_ingest_bit_exact(path):
    g = np.load(path)
    eng = engine_from_golden(g)
    rows, cols = eng.export_adj()
    assert np.array_equal(rows.cpu().numpy(), g["adj_nz_row"])
    assert np.array_equal(cols.cpu().numpy(), g["adj_nz_col"])
    assert np.array_equal(eng.adj_to_edge_index().cpu().numpy(), g["adj_edge_index"])
    pr, pc, pv = eng.export_propagation()
    assert np.array_equal(pr.cpu().numpy(), g["prop_row"])
    assert np.array_equal(pc.cpu().numpy(), g["prop_col"])
    assert rel(pv.cpu().numpy(), g["prop_val"]) < 1e-6
    eng.close()


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_forward_matches_reference(path):
    g = np.load(path)
    eng = engine_from_golden(g)
    out = eng.forward_all().cpu().numpy()
    assert rel(out, g["logits"]) < RTOL
    idx = torch.from_numpy(g["train_idx"]).cuda()
    assert rel(eng.forward(idx).cpu().numpy(), g["logits"][g["train_idx"]]) < RTOL
    eng.close()


@pytest.mark.parametrize("fuse", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_kfac_fit_matches_reference(path, fuse):
    g = np.load(path)
    eng = engine_from_golden(g)
    idx = torch.from_numpy(g["train_idx"]).cuda()
    y = torch.from_numpy(g["train_y"]).cuda()
    views, loss = kfac_fit_engine(eng, idx, y, int(g["batch_size"]), fuse=fuse)
    for l, (A, B) in enumerate(views):
        assert rel(B.cpu().numpy(), g[f"kron_{2 * l}_0"]) < RTOL, f"B_{l}"
        assert rel(A.cpu().numpy(), g[f"kron_{2 * l}_1"]) < RTOL, f"A_{l}"
        assert rel(B.cpu().numpy(), g[f"kron_{2 * l + 1}_0"]) < RTOL, f"bias block {l}"
        assert torch.equal(B, B.T) and torch.equal(A, A.T)  # exactly symmetric by construction
    assert abs(loss - float(g["kron_loss"])) <= RTOL * abs(float(g["kron_loss"]))
    eng.close()


def _diag_fit(eng, idx, y, batch_size):
    H = torch.zeros(eng.n_params, device="cuda")
    loss = torch.zeros(1, device="cuda")
    for s in range(0, len(idx), batch_size):
        eng.diag_accumulate(idx[s:s + batch_size], y[s:s + batch_size], H, loss)
    torch.cuda.synchronize()
    return H.cpu().numpy(), float(loss.item())


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_diag_fit_matches_reference(path):
    g = np.load(path)
    eng = engine_from_golden(g)
    idx = torch.from_numpy(g["train_idx"]).cuda()
    y = torch.from_numpy(g["train_y"]).cuda()
    H, loss = _diag_fit(eng, idx, y, int(g["batch_size"]))
    assert H.shape[0] == int(g["n_params"])
    assert rel(H, g["diag_H"]) < RTOL
    # per parameter block as well (a small block must not hide behind a large one)
    L = int(g["num_layers"])
    off = 0
    for l in range(L):
        for n in (g[f"W{l}"].size, g[f"b{l}"].size):
            assert rel(H[off:off + n], g["diag_H"][off:off + n]) < RTOL, (l, n)
            off += n
    assert abs(loss - float(g["diag_loss"])) <= RTOL * abs(float(g["diag_loss"]))
    eng.close()


@pytest.mark.parametrize("path", [p for p in CASES if "full_H" in np.load(p)],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_lastlayer_full_matches_reference_block(path):
    """The last layer's block of the reference's full GGN (GGNInterface.full golden)."""
    g = np.load(path)
    eng = engine_from_golden(g)
    idx = torch.from_numpy(g["train_idx"]).cuda()
    y = torch.from_numpy(g["train_y"]).cuda()
    L = int(g["num_layers"])
    p_ll = g[f"W{L - 1}"].size + g[f"b{L - 1}"].size
    H = torch.zeros(p_ll, p_ll, device="cuda")
    loss = torch.zeros(1, device="cuda")
    eng.lastlayer_full_accumulate(idx, y, H, loss)
    torch.cuda.synchronize()
    o = conv_offset(g, L - 1)  # the last conv's block (res.* parameters, where present, come after it)
    assert rel(H.cpu().numpy(), g["full_H"][o:o + p_ll, o:o + p_ll]) < RTOL
    assert abs(float(loss.item()) - float(g["full_loss"])) <= RTOL * abs(float(g["full_loss"]))
    eng.close()
