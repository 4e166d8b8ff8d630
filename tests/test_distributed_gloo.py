"""N > 1 path on the CPU: two gloo ranks deal whole batches round-robin, accumulate locally (injected
oracle backend) and sum factors + loss with ONE all-reduce of a flat buffer; the result must equal
the single-process fit and the reference golden.  (On GPUs the same code runs over RCCL.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

RTOL = 1e-4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, structure, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import laplace_gnn_amd as lg
    from oracle_backend import OracleBackend
    from test_host_logic import _cpu_model

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", structure, backend=OracleBackend)
    la.fit(loader)
    mine = [c[1] for c in la.backend.calls]
    expect = [tuple(loader.indices[s:s + loader.batch_size].tolist())
              for t, s in enumerate(range(0, len(loader.dataset), loader.batch_size)) if t % world == rank]
    assert mine == expect, "rank must process exactly the batches t % world == rank, unsplit"
    if structure == "kron":
        blocks = [h.numpy() for F in la.H_facs.kfacs for h in F]
    else:
        blocks = [la.H.numpy()]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=float(la.loss), n_data=la.n_data,
             **{f"b{i}": b for i, b in enumerate(blocks)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("structure,name", [("kron", "gcn_small_3batch_s0"), ("diag", "gcn_small_3batch_s1"),
                                            ("kron", "sage_small_3batch_s1")])
def test_two_rank_fit_equals_single_process(tmp_path, world, structure, name):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, structure, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if structure == "kron":
        ref = [g[f"kron_{i}_{j}"] for i in range(int(g["kron_n_blocks"])) for j in range(2)
               if f"kron_{i}_{j}" in g]
        ref_loss = float(g["kron_loss"])
    else:
        ref, ref_loss = [g["diag_H"]], float(g["diag_loss"])
    for r in range(world):  # every rank holds the reduced result (4 ranks > 3 batches: one rank is idle)
        out = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert int(out["n_data"]) == int(g["n_data"])
        assert abs(float(out["loss"]) - ref_loss) < RTOL * abs(ref_loss)
        for i, b in enumerate(ref):
            got = out[f"b{i}"]
            err = np.linalg.norm(got.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)
            assert err < RTOL, (r, i, err)


def test_all_reduce_flat_single_process_is_noop():
    import laplace_gnn_amd as lg

    t = [torch.ones(3), torch.arange(4.0)]
    lg.all_reduce_flat_(t)
    assert torch.equal(t[0], torch.ones(3)) and torch.equal(t[1], torch.arange(4.0))
