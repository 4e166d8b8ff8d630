"""dev: wall time of la(x) on every node of the arxiv shape, per link approximation (matrix-free routes of csrc/predictive.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

w, ei, X, tri, try_ = bench.make_workload("arxiv", "cuda")
for kind in ("gcn", "sage"):
    torch.manual_seed(0)
    model = (lg.GCN if kind == "gcn" else lg.GraphSAGE)(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
    loader = lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"])
    for cls in (lg.KronLaplace, lg.DiagLaplace):
        la = cls(model, "classification", prior_precision=2.0)
        la.fit(loader)
        every = torch.arange(w["N"], device="cuda")
        for link, kw in (("probit", {}), ("bridge", {}), ("bridge_norm", {}), ("mc", {"diagonal_output": True, "n_samples": 100})):
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = la(every, link_approx=link, **kw)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(kind, cls.__name__, link, kw, f"{dt * 1e3:.1f} ms", float(out.sum(-1).mean()), flush=True)
    model.engine.close()
