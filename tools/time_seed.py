"""Dev tool: time of the arxiv-shaped top-layer kernel (seed SpMM + Gram) under the DEV ablation switches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import laplace_gnn_amd as lg
w, ei, X, ti, ty = bench.make_workload("arxiv", "cuda")
torch.manual_seed(0)
# single layer model: the batch is the top layer only (forward + seeds + seed SpMM/Gram)
for L in (1, 2):
    model = lg.GCN(w["F"], w["H"], w["C"], L, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    idx, y = ti[:10000].cuda(), ty[:10000].cuda()
    _, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, w["n_train"], views, loss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.kfac_accumulate(idx, y, w["n_train"], views, loss)
    torch.cuda.synchronize()
    print(f"L={L} LGNN_SEED_DEBUG={os.environ.get('LGNN_SEED_DEBUG')}: kfac_accumulate per batch {(time.perf_counter()-t0)/10*1e3:.3f} ms", flush=True)
