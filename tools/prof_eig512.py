import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from laplace_gnn_amd.matrix import symeig_batched_hip
n=512
G = torch.randn(4000, n, device="cuda", dtype=torch.float64) * torch.logspace(0, -3, n, device="cuda", dtype=torch.float64)
H = (G.T @ G / 4000).float()
for rep in range(5):
    symeig_batched_hip([H])
torch.cuda.synchronize()
