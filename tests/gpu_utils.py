import numpy as np
import torch

import gnn_laplace_oracle as O
from golden_utils import model_extras


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def engine_from_golden(g, device="cuda"):
    import laplace_gnn_amd as lg

    kind, n = str(g["kind"]), int(g["num_nodes"])
    ei = torch.from_numpy(g["edge_index"]).to(device)
    eng = lg.GraphEngine(ei, n, kind=kind, symmetric=bool(g["symmetric"]))
    L = int(g["num_layers"])
    X = torch.from_numpy(g["X"]).to(device)
    Ws = [torch.from_numpy(g[f"W{l}"]).to(device) for l in range(L)]
    bs = [torch.from_numpy(g[f"b{l}"]).to(device) for l in range(L)]
    # res / norm fixtures (gnn/models/base_gnn.py:86-113): the same arrays as device tensors
    extras = {k: ([torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in v] if isinstance(v, list) else v)
              for k, v in model_extras(g).items()}
    eng.bind(X, Ws, bs, act="relu", **extras)
    return eng


def oracle_from_arrays(kind, n, edge_index, X, Ws, bs, symmetric=False, **extras):
    rp, col = O.edge_index_to_adj_csr(edge_index, n, kind, symmetric)
    return O.GnnModel(kind, rp, col, X, Ws, bs, **extras)


def oracle_model_from_golden(g):
    L = int(g["num_layers"])
    return oracle_from_arrays(str(g["kind"]), int(g["num_nodes"]), g["edge_index"], g["X"],
                              [g[f"W{l}"] for l in range(L)], [g[f"b{l}"] for l in range(L)], bool(g["symmetric"]),
                              **model_extras(g))


def kfac_fit_engine(eng, train_idx, train_y, batch_size, fork_exact=True, fuse=True):
    flat, views, loss = eng.new_kfac_buffers()
    n_train = len(train_idx)
    for s in range(0, n_train, batch_size):
        eng.kfac_accumulate(train_idx[s:s + batch_size], train_y[s:s + batch_size], n_train, views, loss,
                            fork_exact=fork_exact, fuse=fuse)
    torch.cuda.synchronize()
    return views, float(loss.item())
