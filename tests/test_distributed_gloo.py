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


def _worker(rank, world, port, name, structure, out_dir, class_units=False):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import laplace_gnn_amd as lg
    from oracle_backend import OracleBackend, OracleClassBackend
    from test_host_logic import _cpu_model

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", structure,
                    backend=OracleClassBackend if class_units else OracleBackend)
    la.fit(loader)
    batches = [tuple(loader.indices[s:s + loader.batch_size].tolist())
               for s in range(0, len(loader.dataset), loader.batch_size)]
    if structure == "diag":  # sample-additive: every rank takes its slice of every batch
        mine = [c[1] for c in la.backend.calls]
        expect = [b[len(b) * rank // world: len(b) * (rank + 1) // world] for b in batches]
        assert mine == [e for e in expect if e]
    elif class_units:  # balanced (batch, class-range) units; samples of a batch are never split
        C = la.backend.num_classes
        expect = [(batches[t], (c0, c1)) for t, c0, c1 in lg.units_of_rank(len(batches), C, rank, world)]
        assert [(c[1], c[2]) for c in la.backend.calls] == expect
    else:
        mine = [c[1] for c in la.backend.calls]
        assert mine == [b for t, b in enumerate(batches) if t % world == rank], "whole batches, round-robin"
    if structure == "kron":
        blocks = [h.numpy() for F in la.H_facs.kfacs for h in F]
        # the distributed eigendecomposition (factors dealt to ranks + one all-reduce) must equal the local one
        for Fs, ls, Qs in zip(la.H_facs.kfacs, la.H.eigenvalues, la.H.eigenvectors):
            for Hm, lam, Q in zip(Fs, ls, Qs):
                ref_lam, _ = lg.symeig(Hm)
                assert torch.allclose(lam, ref_lam, rtol=1e-4, atol=1e-5 * float(ref_lam.abs().max()))
                assert torch.allclose(Q @ torch.diag(lam) @ Q.T, Hm, rtol=1e-3, atol=1e-4 * float(Hm.abs().max()))
    else:
        blocks = [la.H.numpy()]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=float(la.loss), n_data=la.n_data,
             **{f"b{i}": b for i, b in enumerate(blocks)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,class_units", [(2, False), (4, False), (2, True), (4, True), (8, True)])
@pytest.mark.parametrize("structure,name", [("kron", "gcn_small_3batch_s0"), ("diag", "gcn_small_3batch_s1"),
                                            ("kron", "sage_small_3batch_s1")])
def test_multi_rank_fit_equals_single_process(tmp_path, world, class_units, structure, name):
    if structure == "diag" and class_units:
        pytest.skip("class units only exist for kron")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, structure, str(tmp_path), class_units), nprocs=world, join=True)
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


def _worker_override(rank, world, port, name, out_dir):
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
    out = {}
    for key, (sub, struct) in {"diag": ("all", "diag"), "ll": ("last_layer", "full"), "full": ("all", "full"),
                               "kron": ("all", "kron")}.items():
        la = lg.Laplace(model, "classification", sub, struct, backend=OracleBackend)
        la.fit(loader)
        la.fit(loader, override=False)
        la.fit(loader, override=False)
        H = la.H_facs.to_matrix() if key == "kron" else la.H
        out[key], out[key + "_loss"], out[key + "_n"] = H.numpy(), float(la.loss), la.n_data
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_override_false_in_a_distributed_job_counts_earlier_fits_once(tmp_path):
    """fit; fit(override=False) x 2 at world 2: the earlier, already all-reduced curvature must not go through
    the all-reduce again (it would be multiplied by the world size).  diag / last-layer full / full: H == 3 x the
    single fit's H (tests/test_baselaplace.py:408-426 pattern); kron: equal to the single-process sequence."""
    name = "sage_small_3batch_s1"
    port = _free_port()
    mp.spawn(_worker_override, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import laplace_gnn_amd as lg
    from oracle_backend import OracleBackend
    from test_host_logic import _cpu_model

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    for key, (sub, struct) in {"diag": ("all", "diag"), "ll": ("last_layer", "full"), "full": ("all", "full"),
                               "kron": ("all", "kron")}.items():
        la = lg.Laplace(model, "classification", sub, struct, backend=OracleBackend)
        la.fit(loader)
        H1, loss1 = (None if key == "kron" else la.H.clone()), float(la.loss)
        la.fit(loader, override=False)
        la.fit(loader, override=False)
        ref = la.H_facs.to_matrix().numpy() if key == "kron" else la.H.numpy()
        if key != "kron":
            assert np.allclose(ref, 3 * H1.numpy(), rtol=1e-5, atol=1e-6 * np.abs(ref).max())
        for r in range(2):
            out = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
            err = np.linalg.norm(out[key].astype(np.float64) - ref) / np.linalg.norm(ref)
            assert err < 1e-5, (key, r, err)
            assert abs(float(out[key + "_loss"]) - 3 * loss1) < 1e-5 * abs(3 * loss1)
            assert int(out[key + "_n"]) == 3 * len(loader.dataset)


def test_all_reduce_flat_single_process_is_noop():
    import laplace_gnn_amd as lg

    t = [torch.ones(3), torch.arange(4.0)]
    lg.all_reduce_flat_(t)
    assert torch.equal(t[0], torch.ones(3)) and torch.equal(t[1], torch.arange(4.0))
