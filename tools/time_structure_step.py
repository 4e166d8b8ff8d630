"""One structure-learning step at a BASELINE shape (gnn/marglik_training.py:211-224): fit, adj_backward (= neg_marglik.backward()),
clip, SGD step, re-binarise + lgnn_update_adjacency.  usage: python tools/time_structure_step.py [cora|arxiv|webkb] [diag|kron] [candidates]
(webkb: a Wisconsin-sized graph -- 251 nodes, 1703 features, 5 classes, 120 training nodes, hidden 64 -- with res=True and
norm="layer", every non-edge tracked: the shipped configuration gnn/configs/original/stegcn_config.yaml:54-105)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
structure = sys.argv[2] if len(sys.argv) > 2 else "diag"
ncand = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
g = torch.Generator().manual_seed(1)
extras = {}
if wl == "webkb":
    N = 251
    w = dict(F=1703, H=64, C=5, batch=10000)
    ei = torch.randint(0, N, (2, 515), generator=g)
    X = (torch.rand(N, w["F"], generator=g) < 0.05).float()
    tri = torch.randperm(N, generator=g)[:120]
    try_ = torch.randint(0, w["C"], (120,), generator=g)
    A = torch.zeros(N, N, dtype=torch.bool)
    A[ei[0], ei[1]] = True
    A = A | A.T | torch.eye(N, dtype=torch.bool)
    cand = (~A).nonzero().t().contiguous()
    extras = dict(res=True, norm="layer")
else:
    w, ei, X, tri, try_ = bench.make_workload(wl, "cuda")
    N = X.shape[0]
    cand = torch.randint(0, N, (2, ncand), generator=g)
    cand = cand[:, cand[0] != cand[1]]
torch.manual_seed(0)
model = lg.STEGCN(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True, candidates=cand, **extras).to("cuda").eval()
loader = lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"])
la = (lg.DiagLaplace if structure == "diag" else lg.KronLaplace)(model, "classification")
opt = torch.optim.SGD([model.adj], lr=10.0 if wl == "webkb" else 0.8, weight_decay=5e-4, momentum=0.9)
ts = {k: [] for k in ("fit", "backward", "step", "apply", "flips")}
for it in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    la.fit(loader)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad()
    model.adj_backward(la, loader)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.adj, max_norm=1.0)
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    n = model.apply_adj()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 2:
        for k, v in zip(("fit", "backward", "step", "apply", "flips"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, n)):
            ts[k].append(v)
med = lambda v: sorted(v)[len(v) // 2]
print(f"{wl}/{structure}: tracked pairs {model.adj.numel()}, per step (median of 10): fit {1e3*med(ts['fit']):.3f} ms, adj_backward "
      f"{1e3*med(ts['backward']):.3f} ms, clip + SGD {1e3*med(ts['step']):.3f} ms, re-binarise + update_adjacency {1e3*med(ts['apply']):.3f} ms, "
      f"entries flipped per step {ts['flips']}")
