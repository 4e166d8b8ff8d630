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
from gpu_utils import engine_from_golden, kfac_fit_engine, oracle_from_arrays, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "c1_" not in p)
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
