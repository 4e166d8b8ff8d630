"""Dev tool: per-kernel average time of one arxiv-shaped KFAC batch via torch profiler-free HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import laplace_gnn_amd as lg
w, ei, X, ti, ty = bench.make_workload("arxiv", "cuda")
torch.manual_seed(0)
model = lg.GCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda").eval()
eng = model.engine
idx, y = ti[:10000].cuda(), ty[:10000].cuda()
_, views, loss = eng.new_kfac_buffers()
eng.kfac_accumulate(idx, y, w["n_train"], views, loss)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.kfac_accumulate(idx, y, w["n_train"], views, loss)
torch.cuda.synchronize()
print(f"kfac_accumulate per batch: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
