"""dev: la(x) (GLM predictive, probit link, matrix free) on ALL nodes of the arxiv-shaped model, GCN and GraphSAGE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

w, ei, X, tri, try_ = bench.make_workload("arxiv", "cuda")
for kind, cls in (("gcn", lg.GCN), ("sage", lg.GraphSAGE)):
    torch.manual_seed(0)
    model = cls(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
    la = lg.KronLaplace(model, "classification", prior_precision=2.0)
    la.fit(lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"]))
    every = torch.arange(w["N"], device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        probs = la(every, link_approx="probit")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(probs).all()) and float((probs.sum(-1) - 1).abs().max()) < 1e-4
    sub = torch.randperm(w["N"], generator=torch.Generator().manual_seed(0))[:16].cuda()
    Js, _ = la.backend.jacobians(sub)
    ref = torch.diagonal(la.functional_variance(Js), dim1=1, dim2=2)
    _, fv = la._glm_variance_matrix_free(sub)
    print(kind, f"{dt * 1e3:.0f} ms for {w['N']} nodes, valid={ok}, vs Jacobian route on 16 nodes: rel {float((fv - ref).norm() / ref.norm()):.2e}")
    del la, model
    torch.cuda.empty_cache()
