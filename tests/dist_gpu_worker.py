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
        model.engine.check_async_errors()
    dist.barrier()
    if rank == 0:
        print(f"DIST_GPU_OK world={world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
