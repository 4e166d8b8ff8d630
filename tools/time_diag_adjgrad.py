"""dev: wall time of DiagLaplace.fit and DiagLaplace.neg_marglik_adj_grad at the Cora shape (BASELINE configs[1]: what the shipped
STE-GCN configuration runs per hyper-step, gnn/marglik_training.py:197-216) and at the arxiv shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

for wl in sys.argv[1:] or ["cora", "arxiv"]:
    w, ei, X, tri, try_ = bench.make_workload(wl, "cuda")
    torch.manual_seed(0)
    model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
    loader = lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"])
    la = lg.DiagLaplace(model, "classification")
    la.fit(loader)
    torch.cuda.synchronize(); t0 = time.perf_counter(); la.fit(loader); torch.cuda.synchronize()
    print(wl, f"fit {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    cand_all = torch.randint(0, w["N"], (2, 20_000), generator=torch.Generator().manual_seed(1)).cuda()
    sr, sc = model.engine.export_adj()
    skey, ckey = sr * w["N"] + sc, cand_all[0] * w["N"] + cand_all[1]
    pos = torch.searchsorted(skey, ckey).clamp(max=skey.numel() - 1)
    cand_all = cand_all[:, (skey[pos] != ckey) & (cand_all[0] != cand_all[1])]
    for cand in (None, cand_all):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = la.neg_marglik_adj_grad(loader, candidates=cand)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(wl, "candidates %d" % cand.shape[1] if cand is not None else "stored only", f"{dt * 1e3:.1f} ms", float(out[0]),
              float(out[2].abs().sum()), flush=True)
    model.engine.check_async_errors()
    model.engine.close()
