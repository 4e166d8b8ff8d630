"""Where the HOST time of a small fit goes (Cora shape: the device work is ~0.2 ms, so Python / ctypes overhead shows).
usage: python tools/host_profile_fit.py [workload] [structure] [fits]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import laplace_gnn_amd as lg  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
structure = sys.argv[2] if len(sys.argv) > 2 else bench.DEFAULT_STRUCTURE[wl]
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dev = torch.device("cuda", 0)
w, ei, X, train_idx, train_y = bench.make_workload(wl, dev)
torch.manual_seed(0)
cls = lg.GraphSAGE if w.get("kind") == "sage" else lg.GCN
model = cls(w["F"], w["H"], w["C"], w.get("layers", 2), X, ei, symmetric=True).to(dev)
loader = lg.TensorBatchLoader(train_idx.to(dev), train_y.to(dev), batch_size=w["batch"])
la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure=structure)
eng = model.engine


def step():
    eng.invalidate()
    la.fit(loader)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(fits):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{wl}/{structure}: host enqueue {1e3 * t_enq / fits:.4f} ms per fit, with device {1e3 * t_all / fits:.4f} ms per fit")
pr = cProfile.Profile()
pr.enable()
for _ in range(fits):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
