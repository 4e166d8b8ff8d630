"""Worker of tests/test_gpu_frontend.py::test_two_ranks_share_one_gpu: one rank of a 2-rank job on ONE GPU (gloo all-reduce
of CUDA tensors; RCCL refuses two ranks on the same device).  Fits kron / diag / full through the HIP backend with the
(batch, class) and sample sharding of an N-GPU run and checks the all-reduced result against the reference golden."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import laplace_gnn_amd as lg
    from test_gpu_frontend import model_from_golden

    for name in ("gcn_mid_3batch_sym_s1", "sage_small_3batch_s1"):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        model = model_from_golden(g)
        loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                      batch_size=int(g["batch_size"]))
        la = lg.Laplace(model, "classification", "all", "kron")
        la.fit(loader)  # every rank accumulates its units, one all-reduce, every rank decomposes
        for i, Fs in enumerate(la.H_facs.kfacs):
            for j, Hm in enumerate(Fs):
                assert rel(Hm.cpu().numpy(), g[f"kron_{i}_{j}"]) < 1e-4, (name, rank, i, j)
        assert abs(float(la.loss) - float(g["kron_loss"])) < 1e-4 * float(g["kron_loss"])
        assert abs(float(la.log_marginal_likelihood()) - float(g["kron_marglik_pp1"])) < 3e-4 * abs(float(g["kron_marglik_pp1"]))
        ld = lg.Laplace(model, "classification", "all", "diag")
        ld.fit(loader)
        assert rel(ld.H.cpu().numpy(), g["diag_H"]) < 1e-4, (name, rank)
        assert la.n_data == int(g["n_data"]) and ld.n_data == int(g["n_data"])
        # last-layer full GGN: every rank's sample slices accumulate pair-major, ONE in-place all-reduce of the pair
        # buffers, one placement; compared with the single-call result of the whole training set
        ll = lg.Laplace(model, "classification", "last_layer", "full")
        ll.fit(loader)
        eng = model.engine
        p_ll = ll.n_params
        Href = torch.zeros(p_ll, p_ll, device="cuda")
        lref = torch.zeros(1, device="cuda")
        eng.lastlayer_full_accumulate(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(), Href, lref)
        assert rel(ll.H.cpu().numpy(), Href.cpu().numpy()) < 1e-5, (name, rank)
        assert abs(float(ll.loss) - float(lref)) < 1e-5 * abs(float(lref))
        ll.fit(loader, override=False)  # the earlier, already reduced H must not pass through the all-reduce again
        assert rel(ll.H.cpu().numpy(), 2 * Href.cpu().numpy()) < 1e-5, (name, rank)
        # empirical Fisher backend: whole batches round-robin through backend.kron
        le = lg.Laplace(model, "classification", "all", "kron", backend=lg.HipEF)
        le.fit(loader)
        le1 = lg.KronLaplace(model, "classification", backend=lg.HipEF)
        le1.fit(loader, process_group=dist.new_group([rank]))  # a one-rank group: the single-process fit
        for Fa, Fb in zip(le.H_facs.kfacs, le1.H_facs.kfacs):
            for a_, b_ in zip(Fa, Fb):
                assert rel(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-5, (name, rank, "ef")
        if "adjgrad_vals" in g:  # gradient w.r.t. the adjacency: batches dealt round-robin, one all-reduce of the accumulators
            lp = lg.KronLaplace(model, "classification", prior_precision=float(g["adjgrad_prior"]))
            lp.fit(loader)
            cand = torch.from_numpy(np.stack([g["adjgrad_ne_row"], g["adjgrad_ne_col"]])).cuda()
            val, ei, grad, gc = lp.neg_marglik_adj_grad(loader, candidates=cand)
            assert rel(grad.cpu().numpy(), g["adjgrad_vals"]) < 1e-5 and rel(gc.cpu().numpy(), g["adjgrad_ne_val"]) < 1e-5
        if "adjgrad_diag_vals" in g:  # the same under the diagonal posterior (four accumulators + the candidates', one all-reduce)
            ldg = lg.DiagLaplace(model, "classification", prior_precision=float(g["adjgrad_prior"]))
            ldg.fit(loader)
            cand = torch.from_numpy(np.stack([g["adjgrad_ne_row"], g["adjgrad_ne_col"]])).cuda()
            val, ei, grad, gc = ldg.neg_marglik_adj_grad(loader, candidates=cand)
            assert abs(float(val) - float(g["adjgrad_diag_neg_marglik"])) <= 5e-6 * abs(float(g["adjgrad_diag_neg_marglik"]))
            assert rel(grad.cpu().numpy(), g["adjgrad_diag_vals"]) < 1e-5 and rel(gc.cpu().numpy(), g["adjgrad_diag_ne_val"]) < 1e-5
        model.engine.check_async_errors()
    # the path routes (hidden width > 128) cut a batch by DESTINATION NODES (lgnn_kfac_accumulate_share): the ranks' node ranges
    # of every batch must add up to the single-process fit, for the GCN's two-hop and GraphSAGE's one-hop paths
    solo = [dist.new_group([r]) for r in range(world)][rank]  # (every rank creates every group, in the same order)
    for kind in ("gcn", "sage"):
        N, F, H, C, E = 3000, 40, 256, 10, 12000
        gen = torch.Generator().manual_seed(31)
        ei = torch.randint(0, N, (2, E), generator=gen)
        X = torch.randn(N, F, generator=gen)
        torch.manual_seed(4)
        model = (lg.GCN if kind == "gcn" else lg.GraphSAGE)(F, H, C, 2, X, ei, symmetric=True).to("cuda").eval()
        idx = torch.randperm(N, generator=gen)[:700].cuda()
        y = torch.randint(0, C, (700,), generator=gen).cuda()
        loader = lg.TensorBatchLoader(idx, y, batch_size=300)  # 300 / 300 / 100: 30 (batch, part) units over the ranks
        la = lg.KronLaplace(model, "classification")
        la.fit(loader)
        assert model.engine.last_kfac_used_paths, kind
        la1 = lg.KronLaplace(model, "classification")
        la1.fit(loader, process_group=solo)
        for Fa, Fb in zip(la.H_facs.kfacs, la1.H_facs.kfacs):
            for a_, b_ in zip(Fa, Fb):
                assert rel(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-5, (kind, rank, "node shares")
        assert abs(float(la.loss) - float(la1.loss)) < 1e-5 * abs(float(la1.loss))
        model.engine.check_async_errors()
    dist.barrier()
    if rank == 0:
        print(f"DIST_GPU_OK world={world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
