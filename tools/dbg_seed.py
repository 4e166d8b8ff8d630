import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import test_gpu_scale as T
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
N, F, H, C, E = 3000, 48, 64, 10, 12000
ei, X, Ws, bs = T._make("gcn", N, F, H, C, E, seed=11)
g = torch.Generator().manual_seed(5)
idx = torch.randperm(N, generator=g)[:700].cuda(); y = torch.randint(0, C, (700,), generator=g).cuda()
eng = T._engine("gcn", N, ei, X, Ws, bs)
for r in (None, (0, 3), (3, 4), (4, 10), (0, 10), (2, 10), (0, 8)):
    out = []
    for fuse in (True, False):
        _, v, l = eng.new_kfac_buffers()
        eng.kfac_accumulate(idx[:300], y[:300], 700, v, l, classes=r, fuse=fuse)
        torch.cuda.synchronize()
        out.append([b.cpu().numpy() for _, b in v])
    print(r, "B0 rel", rel(out[0][0], out[1][0]), "B1 rel", rel(out[0][1], out[1][1]), flush=True)
