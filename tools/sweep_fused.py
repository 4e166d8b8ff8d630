"""Dev tool: fused-kernel time per plane vs graph size (is the gather served by the Infinity Cache?)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import laplace_gnn_amd as lg

def run(N, deg=6.9, F=128, H=256, C=40, M=10000):
    g = torch.Generator().manual_seed(0)
    E = int(N * deg)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    torch.manual_seed(0)
    model = lg.GCN(F, H, C, 2, X, ei, symmetric=True).to("cuda").eval()
    eng = model.engine
    idx = torch.randperm(N, generator=g)[:M].cuda()
    y = torch.randint(0, C, (M,), generator=g).cuda()
    _, views, loss = eng.new_kfac_buffers()
    eng.kfac_accumulate(idx, y, M, views, loss)
    torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    for _ in range(3):
        eng.kfac_accumulate(idx, y, M, views, loss)
    torch.cuda.synchronize()
    n, ms, planes = eng.kernel_timing()
    nnz = eng.nnz
    per_plane = ms / planes
    print(f"N={N:7d} nnz={nnz:8d} plane={N*H*4/1e6:6.1f} MB  fused {per_plane*1e3:7.1f} us/plane  "
          f"{nnz*H*4/per_plane/1e9:7.2f} TB/s (all entries)  ", flush=True)
    eng.close()

sizes = [int(a) for a in sys.argv[1:]] or [20000, 40000, 80000, 169343, 340000]
for N in sizes:
    run(N)
