"""dev: per-sample Jacobians and the accumulators built on them (full-weight diagonal EF, MC-free generic diagonal,
GLM predictive with the full covariance) at the arxiv shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg
from laplace_gnn_amd.engine import GraphEngine

w, ei, X, tri, try_ = bench.make_workload("arxiv", "cuda")
torch.manual_seed(0)
for kind, cls in (("gcn", lg.GCN), ("sage", lg.GraphSAGE)):
    model = cls(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
    eng = model.engine
    idx = tri[:64].cuda()
    for route in ("closed form", "planes"):
        if route == "planes":
            os.environ["LGNN_JAC_PLANES"] = "1"
            idx = idx[:4]
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            J, f = eng.jacobians(idx)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        os.environ.pop("LGNN_JAC_PLANES", None)
        print(kind, route, f"M={idx.numel()} {dt * 1e3:.2f} ms  {dt * 1e3 / idx.numel():.3f} ms/sample  "
              f"{J.numel() * 4 / dt / 1e9:.0f} GB/s written")
    # diagonal empirical Fisher over 4096 training nodes
    idx = tri[:4096].cuda(); y = try_[:4096].cuda()
    d = torch.zeros(eng.n_params, device="cuda"); loss = torch.zeros(1, device="cuda")
    for rep in range(2):
        d.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.ef_accumulate(idx, y, y_loss=y, diag=d, loss=loss)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kind, f"diag EF over 4096 nodes: {dt * 1e3:.1f} ms")
