"""dev: lgnn_symeig_batched -- hand-written tridiagonalisation path vs the library's strided-batched syevd (LGNN_EIGH_LIBRARY=1)
on factors of the arxiv-shaped fit (256, 256, 128, 40; rank-deficient ones included): time and accuracy against fp64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laplace_gnn_amd.matrix import symeig_batched_hip

torch.manual_seed(0)
dev = "cuda"
mats = []
for n, rank in ((256, 256), (256, 200), (128, 128), (40, 39)):
    G = torch.randn(4000, n, device=dev, dtype=torch.float64) * torch.logspace(0, -3, n, device=dev, dtype=torch.float64)
    if rank < n:
        G[:, rank:] = G[:, :n - rank] * 0.5 + G[:, 1:n - rank + 1]  # exact linear dependence: zero eigenvalues
    mats.append((G.T @ G / 4000).float())
for path in ("hand-written, side streams", "hand-written, one stream", "library"):
    if path == "library":
        os.environ["LGNN_EIGH_LIBRARY"] = "1"
    if "one stream" in path:
        os.environ["LGNN_EIGH_ONE_STREAM"] = "1"
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = symeig_batched_hip(mats)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    errs = []
    for H, (lam, Q) in zip(mats, out):
        ref = torch.linalg.eigvalsh(H.double()).clamp(min=0)
        rec = (Q * lam) @ Q.T
        errs.append((float((lam.double() - ref).abs().max() / ref.max()), float((rec - H).norm() / H.norm()),
                     float((Q.T @ Q - torch.eye(H.shape[0], device=dev)).abs().max())))
    print(path, f"{dt * 1e3:.2f} ms", " ".join(f"[lam {a:.1e} rec {b:.1e} orth {c:.1e}]" for a, b, c in errs))


# ---- round 4: 256 < n <= 512 (streaming tridiagonalisation + the register-resident kernel) vs the library, one factor per call
os.environ.pop("LGNN_EIGH_LIBRARY", None)
os.environ.pop("LGNN_EIGH_ONE_STREAM", None)
for n in (512, 384, 1433):
    G = torch.randn(4000, n, device=dev, dtype=torch.float64) * torch.logspace(0, -3, n, device=dev, dtype=torch.float64)
    H = (G.T @ G / 4000).float()
    for path in ("hand-written", "library"):
        if path == "library":
            os.environ["LGNN_EIGH_LIBRARY"] = "1"
        else:
            os.environ.pop("LGNN_EIGH_LIBRARY", None)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            (lam, Q), = symeig_batched_hip([H])
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ref = torch.linalg.eigvalsh(H.double()).clamp(min=0)
        rec = (Q * lam) @ Q.T
        print(f"n = {n} {path}: {1e3 * sorted(ts)[len(ts) // 2]:.2f} ms  lam {float((lam.double() - ref).abs().max() / ref.max()):.1e} "
              f"rec {float((rec - H).norm() / H.norm()):.1e} orth {float((Q.T @ Q - torch.eye(n, device=dev)).abs().max()):.1e}")
