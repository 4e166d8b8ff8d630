"""dev: wall time of KronLaplace.neg_marglik_adj_grad at the arxiv shape (with / without candidate pairs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

w, ei, X, tri, try_ = bench.make_workload(sys.argv[2] if len(sys.argv) > 2 else "arxiv", "cuda")
torch.manual_seed(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "gcn"
model = (lg.GCN if kind == "gcn" else lg.GraphSAGE)(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
loader = lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"])
la = lg.KronLaplace(model, "classification")
la.fit(loader)
t0 = time.perf_counter(); la.fit(loader); torch.cuda.synchronize(); print(kind, f"fit {(time.perf_counter() - t0) * 1e3:.2f} ms")
cand_all = torch.randint(0, w["N"], (2, 100_000), generator=torch.Generator().manual_seed(1)).cuda()
sr, sc = model.engine.export_adj()  # stored entries are not candidates (KronLaplace.neg_marglik_adj_grad refuses them)
skey = sr * w["N"] + sc
ckey = cand_all[0] * w["N"] + cand_all[1]
pos = torch.searchsorted(skey, ckey).clamp(max=skey.numel() - 1)
cand_all = cand_all[:, skey[pos] != ckey]
for cand in (None, cand_all):
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = la.neg_marglik_adj_grad(loader, candidates=cand)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kind, "candidates" if cand is not None else "stored only", f"{dt * 1e3:.1f} ms", float(out[0]), float(out[2].abs().sum()))
