"""Which stage owns the adjacency gradient's device error?  Runs ``KronLaplace.neg_marglik_adj_grad``
on the golden fixtures that hold the reference's ``adj.grad`` and on seeded mid-size cases against the CPU oracle, three
ways: (V0) as shipped in rounds 1-2 -- the factor gradients from the fit's fp32 eigenpairs (batched device solver); (V1,
shipped since round 3: KronLaplace._logdet_factor_gradients) the factor-gradient matrices Gamma_B / Gamma_A
(d logdet P / d factor) formed in fp64 from an fp64 eigendecomposition of the SAME fp32 factors, cast to fp32 -- isolates
the eigenpairs + the 1 / (f lB_i lA_j + delta) contraction; (V2) V1 on the ORACLE's factors (CPU, another summation order) -- shows
how much of the rest is the fp32 accumulation of the factors themselves.  What remains after V2 is the sparse chain (SDDMM / SpMM per class plane).
usage: python tools/adjgrad_attribution.py"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gnn_laplace_oracle as O  # noqa: E402
import laplace_gnn_amd as lg  # noqa: E402
from gpu_utils import oracle_from_arrays, rel  # noqa: E402
from test_gpu_frontend import model_from_golden  # noqa: E402


def gamma64(la, factors=None):
    """_logdet_factor_gradients in fp64 (from the device's fp32 factors, or from `factors` = [[B, A], [Bb], ...] fp64)."""
    f = float(la._H_factor)
    pp = torch.as_tensor(la.prior_precision, dtype=torch.float64, device=la._device).reshape(-1)
    deltas = pp.expand(la.n_layers) if pp.numel() == 1 else pp
    kf = factors if factors is not None else [[t.double() for t in F] for F in la.H_facs.kfacs]
    gB, gA = [], []
    for l in range(len(kf) // 2):
        lB, QB = torch.linalg.eigh(kf[2 * l][0])
        lA, QA = torch.linalg.eigh(kf[2 * l][1])
        lBb, QBb = torch.linalg.eigh(kf[2 * l + 1][0])
        den = f * torch.outer(lB, lA) + deltas[2 * l]
        cB = (f * lA.unsqueeze(0) / den).sum(dim=1)
        cA = (f * lB.unsqueeze(1) / den).sum(dim=0)
        cBb = f / (f * lBb + deltas[2 * l + 1])
        gB.append(((QB * cB) @ QB.T + (QBb * cBb) @ QBb.T).float())
        gA.append(((QA * cA) @ QA.T).float())
    return gB, gA


def gamma32(la):
    """the round-1/2 formulation: the fit's own fp32 eigenpairs"""
    f = la._H_factor
    pp = torch.as_tensor(la.prior_precision, dtype=torch.float32, device=la._device).reshape(-1)
    deltas = pp.expand(la.n_layers) if pp.numel() == 1 else pp
    gB, gA = [], []
    ev, qs = la.H.eigenvalues, la.H.eigenvectors
    for l in range(len(ev) // 2):
        (lB, lA), (QB, QA) = ev[2 * l], qs[2 * l]
        (lBb,), (QBb,) = ev[2 * l + 1], qs[2 * l + 1]
        den = f * torch.outer(lB, lA) + deltas[2 * l]
        cB = (f * lA.unsqueeze(0) / den).sum(dim=1)
        cA = (f * lB.unsqueeze(1) / den).sum(dim=0)
        cBb = f / (f * lBb + deltas[2 * l + 1])
        gB.append((QB * cB) @ QB.T + (QBb * cBb) @ QBb.T)
        gA.append((QA * cA) @ QA.T)
    return gB, gA


def run(la, loader, override=None):
    if override is not None:
        la._logdet_factor_gradients = lambda: override
    else:
        la.__dict__.pop("_logdet_factor_gradients", None)
    val, _, grad = la.neg_marglik_adj_grad(loader)
    return float(val), grad.cpu().numpy()


def gamma_err(la, ref):
    gB, gA = gamma32(la)
    return max(rel(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(gB + gA, ref[0] + ref[1]))


print(f"{'case':44s} {'V0 fp32 eig':>12s} {'V1 fp64 eig':>12s} {'V2 orc. fac':>12s} {'|Gamma32-64|':>12s} {'value err':>10s} {'cand. err':>10s}")
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    g = np.load(path)
    if "adjgrad_vals" not in g:
        continue
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    la = lg.KronLaplace(model, "classification", prior_precision=float(g["adjgrad_prior"]))
    la.fit(loader)
    v0, g0 = run(la, loader, gamma32(la))
    g64 = gamma64(la)
    ge = gamma_err(la, g64)
    _, g1 = run(la, loader, g64)
    la.__dict__.pop("_logdet_factor_gradients", None)
    cand = torch.from_numpy(np.stack([g["adjgrad_ne_row"], g["adjgrad_ne_col"]])).cuda()
    gc = la.neg_marglik_adj_grad(loader, candidates=cand)[3].cpu().numpy()
    ve = abs(v0 - float(g["adjgrad_neg_marglik"])) / abs(float(g["adjgrad_neg_marglik"]))
    print(f"{os.path.basename(path)[:-4]:44s} {rel(g0, g['adjgrad_vals']):12.2e} {rel(g1, g['adjgrad_vals']):12.2e} {'':>12s} {ge:12.2e} {ve:10.2e} {rel(gc, g['adjgrad_ne_val']):10.2e}")

for kind, fork_exact, sym, H, C in [("gcn", True, True, 64, 10), ("gcn", True, False, 256, 12), ("sage", True, True, 64, 10),
                                    ("sage", False, False, 32, 7), ("sage", True, False, 256, 12)]:
    N, F, E, M = 2500, 24, 9000, 500
    gen = torch.Generator().manual_seed(13)
    ei = torch.randint(0, N, (2, E), generator=gen)
    X = torch.randn(N, F, generator=gen)
    torch.manual_seed(1)
    model = (lg.GCN if kind == "gcn" else lg.GraphSAGE)(F, H, C, 2, X, ei, symmetric=sym).to("cuda").eval()
    idx = torch.randperm(N, generator=gen)[:M]
    y = torch.randint(0, C, (M,), generator=gen)
    loader = lg.TensorBatchLoader(idx.cuda(), y.cuda(), batch_size=200)
    la = lg.KronLaplace(model, "classification", prior_precision=0.5, backend_kwargs=dict(fork_exact_seed=fork_exact))
    la.fit(loader)
    Ws = [c.lin.weight.detach().cpu().numpy() for c in model.convs]
    bs = [c.lin.bias.detach().cpu().numpy() for c in model.convs]
    om = oracle_from_arrays(kind, N, ei.numpy(), X.numpy(), Ws, bs, sym)
    oval, rows, cols, og = O.kron_marglik_adj_grad(om, idx.numpy(), y.numpy(), 200, 0.5, fork_exact, sym)
    v0, g0 = run(la, loader, gamma32(la))
    g64 = gamma64(la)
    ge = gamma_err(la, g64)
    _, g1 = run(la, loader, g64)
    # fp64 factors from the oracle's fit (same batches)
    _, oH = O.fit_kron(om, idx.numpy(), y.numpy(), 200, fork_exact)
    fac = [[torch.from_numpy(np.asarray(t, np.float64)).cuda() for t in F_] for F_ in oH]
    _, g2 = run(la, loader, gamma64(la, fac))
    name = f"midsize {kind} H={H} C={C} fork={int(fork_exact)} sym={int(sym)}"
    print(f"{name:44s} {rel(g0, og):12.2e} {rel(g1, og):12.2e} {rel(g2, og):12.2e} {ge:12.2e} {abs(v0 - oval) / abs(oval):10.2e}")
