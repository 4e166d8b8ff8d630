"""Drop-in front-end on the GPU: Laplace(...).fit(loader) with the HipGGN backend reproduces the
reference's fitted attributes (golden vectors from the reference's KronLaplace / DiagLaplace)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from golden_utils import constructor_extras
from gpu_utils import rel

pytestmark = pytest.mark.gpu
RTOL = 1e-4
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if "c1_" not in p and "steloop_" not in p)
IDS = [os.path.basename(p)[:-4] for p in CASES]


def model_from_golden(g, device="cuda"):
    import laplace_gnn_amd as lg

    kind = str(g["kind"])
    L = int(g["num_layers"])
    X = torch.from_numpy(g["X"])
    ei = torch.from_numpy(g["edge_index"])
    F, C = X.shape[1], g[f"W{L - 1}"].shape[0]
    Hd = g["W0"].shape[0]
    cls = lg.GCN if kind == "gcn" else lg.GraphSAGE
    model = cls(F, Hd, C, L, X, ei, symmetric=bool(g["symmetric"]), **constructor_extras(g))
    with torch.no_grad():
        for l, conv in enumerate(model.convs):
            conv.lin.weight.copy_(torch.from_numpy(g[f"W{l}"]))
            conv.lin.bias.copy_(torch.from_numpy(g[f"b{l}"]))
        for l, lin in enumerate(model.res):  # res=True fixtures (gnn/models/base_gnn.py:100-113)
            lin.weight.copy_(torch.from_numpy(g[f"Wr{l}"]))
            lin.bias.copy_(torch.from_numpy(g[f"br{l}"]))
        if model.norm_kind is not None:  # norm fixtures: LayerNorm / eval-mode BatchNorm1d state
            for l, nm in enumerate(model.norms):
                nm.weight.copy_(torch.from_numpy(g[f"norm_w{l}"]))
                nm.bias.copy_(torch.from_numpy(g[f"norm_b{l}"]))
                if model.norm_kind == "batch":
                    nm.running_mean.copy_(torch.from_numpy(g[f"norm_rm{l}"]))
                    nm.running_var.copy_(torch.from_numpy(g[f"norm_rv{l}"]))
    model.eval()
    return model.to(device) if device else model


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_kron_laplace_fit(path):
    import laplace_gnn_amd as lg

    g = np.load(path)
    model = model_from_golden(g)
    names = [k for k, _ in model.named_parameters() if "norms" not in k]  # the fork's filter, laplace/curvature/curvature.py:74-79
    assert names == [f"convs.{l}.lin.{w}" for l in range(int(g["num_layers"])) for w in ("weight", "bias")] + [
        f"res.{l}.{w}" for l in range(len(model.res)) for w in ("weight", "bias")]
    if "param_names" in g.files:  # res / norm fixtures record the reference's own parameter order
        assert names == [str(k) for k in g["param_names"]]
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure="kron")
    assert isinstance(la, lg.KronLaplace) and la._backend_cls is lg.HipGGN
    la.fit(loader)
    assert len(la.H_facs.kfacs) == int(g["kron_n_blocks"])
    for i, Fs in enumerate(la.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            assert rel(Hm.cpu().numpy(), g[f"kron_{i}_{j}"]) < RTOL, (i, j)
    assert abs(float(la.loss) - float(g["kron_loss"])) <= RTOL * abs(float(g["kron_loss"]))
    assert la.n_data == int(g["n_data"]) and la.n_outputs == int(g["n_outputs"]) and la.n_params == int(g["n_params"])
    assert model.output_size == la.n_outputs
    # the eigendecomposition fit() ends with (eigenvalues are basis independent)
    for i, ls in enumerate(la.H.eigenvalues):
        for j, lam in enumerate(ls):
            ref = g[f"kron_eig_{i}_{j}"]
            # (measured over all fixtures, tools/tolerance_survey.py: <= 8.2e-7 of the largest eigenvalue; marglik <= 3.7e-7)
            assert np.abs(lam.cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30), (i, j)
    # "next" row 8(f)-1: marginal likelihood on the device from the decomposed factors (eigh + logdet)
    for pp, key in ((None, "kron_marglik_pp1"), (torch.tensor(0.7), "kron_marglik_pp07"),
                    (torch.from_numpy(g["kron_prior_layerwise"]), "kron_marglik_layerwise")):
        got = float(la.log_marginal_likelihood(prior_precision=pp))
        assert abs(got - float(g[key])) <= 5e-6 * abs(float(g[key])), key
    la.prior_precision = 1.0
    # torch.utils.data.DataLoader (the reference's loader) gives the same batches
    from torch.utils.data import DataLoader, TensorDataset
    dl = DataLoader(TensorDataset(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"])),
                    batch_size=int(g["batch_size"]), shuffle=False)
    la2 = lg.KronLaplace(model, "classification", backend=lg.HipGGN)
    la2.fit(dl)
    for Fa, Fb in zip(la.H_facs.kfacs, la2.H_facs.kfacs):
        for a, b in zip(Fa, Fb):
            assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_diag_laplace_fit(path):
    import laplace_gnn_amd as lg

    g = np.load(path)
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure="diag")
    la.fit(loader)
    assert rel(la.H.cpu().numpy(), g["diag_H"]) < RTOL
    assert abs(float(la.loss) - float(g["diag_loss"])) <= RTOL * abs(float(g["diag_loss"]))
    assert la.mean.shape[0] == int(g["n_params"])
    for pp, key in ((None, "diag_marglik_pp1"), (torch.tensor(0.7), "diag_marglik_pp07")):
        got = float(la.log_marginal_likelihood(prior_precision=pp))
        assert abs(got - float(g[key])) <= 5e-6 * abs(float(g[key])), key


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_posterior_samples_and_sampling_predictive(path):
    """"next" row 8(f)-2 on the GPU: reference-generated samples / Monte-Carlo predictive on fixed draws; every
    sample's forward runs through the HIP engine, which must follow the parameter storage swaps of
    torch.nn.utils.vector_to_parameters."""
    import laplace_gnn_amd as lg

    g = np.load(path)
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    eps, idx = torch.from_numpy(g["pred_eps"]).cuda(), torch.from_numpy(g["pred_idx"]).cuda()
    logits_before = model(idx).clone()
    for structure in ("kron", "diag"):
        la = lg.Laplace(model, "classification", "all", structure)
        la.fit(loader)
        assert rel(la.sample(eps=eps).cpu().numpy(), g[structure + "_samples"]) < 5e-5
        py = la(idx, pred_type="nn", link_approx="mc", n_samples=len(eps), eps=eps)
        assert np.abs(py.cpu().numpy() - g[structure + "_nn_py"]).max() < 5e-5
        # mean restored, engine follows it back
        assert torch.equal(model(idx), logits_before)
        la.fit(loader)  # and a refit after predicting still sees the MAP weights
        py2 = la(idx, pred_type="nn", link_approx="mc", n_samples=len(eps), eps=eps)
        assert torch.allclose(py, py2, atol=5e-5)  # refit sums with float atomics: last-bit differences
    model.engine.check_async_errors()


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_jacobians_and_glm_predictive(path):
    """"next" row 8(f)-3 on the GPU: lgnn_jacobians against the reference's torch.func Jacobians (goldens of the
    1-batch small cases) and against the oracle everywhere; the GLM predictive (default la(x): glm + probit, plus
    bridge, bridge_norm, mc on fixed draws) against the reference's own output."""
    import gnn_laplace_oracle as O
    import laplace_gnn_amd as lg
    from gpu_utils import oracle_model_from_golden

    g = np.load(path)
    model = model_from_golden(g)
    idx_all = torch.from_numpy(g["train_idx"]).cuda()
    Js, f = model.engine.jacobians(idx_all)
    oJ, of = O.jacobians_batch(oracle_model_from_golden(g), g["train_idx"])
    assert rel(Js.cpu().numpy(), oJ) < RTOL and rel(f.cpu().numpy(), of) < RTOL
    if "jac_first_batch" in g.files:
        assert rel(Js.cpu().numpy(), g["jac_first_batch"]) < RTOL and rel(f.cpu().numpy(), g["f_first_batch"]) < RTOL
    # a node listed twice gets the same Jacobian twice; an empty request is fine
    twice = torch.cat([idx_all[:3], idx_all[:3]])
    J2, _ = model.engine.jacobians(twice)
    assert torch.equal(J2[:3], J2[3:]) and model.engine.jacobians(idx_all[:0])[0].shape[0] == 0

    loader = lg.TensorBatchLoader(idx_all, torch.from_numpy(g["train_y"]).cuda(), batch_size=int(g["batch_size"]))
    idx = torch.from_numpy(g["pred_idx"]).cuda()
    for structure in ("kron", "diag"):
        la = lg.Laplace(model, "classification", "all", structure)
        la.fit(loader)
        f_mu, f_var = la._glm_predictive_distribution(idx)
        assert rel(f_mu.cpu().numpy(), g[structure + "_glm_fmu"]) < RTOL
        # (J P^-1 J^T divides by eigenvalue products + prior precision: inverted from fp64 eigenpairs of the fitted factors,
        #  KronLaplace._eigh64 -- with the fit's fp32 pairs the small eigenvalues' absolute error, different from run to run
        #  through the accumulation's float atomics, gave 8e-7 ... 2e-4 on the same fixture; now 3e-7 ... 2.4e-5 over 24 refits of
        #  the worst fixture, <= 4e-6 on every other one (profiles/r04_tolerance_survey.log).  What is left is the problem's own
        #  conditioning: the factors' last bits, ~1e-7 of their norm, times ||P^-1|| ||H|| ~ 2e2 at that fixture.)
        assert rel(f_var.cpu().numpy(), g[structure + "_glm_fvar"]) < 1e-4
        assert np.abs(la(idx).cpu().numpy() - g[structure + "_glm_probit"]).max() < 5e-5
        for link in ("bridge", "bridge_norm"):
            got = la(idx, pred_type="glm", link_approx=link).cpu().numpy()
            assert np.abs(got - g[f"{structure}_glm_{link}"]).max() < 2e-5, link
        mc = la(idx, pred_type="glm", link_approx="mc", n_samples=6, eps=torch.from_numpy(g["glm_eps"]).cuda())
        assert np.abs(mc.cpu().numpy() - g[structure + "_glm_mc"]).max() < 2e-5
        if structure == "kron":  # post-hoc prior tuning on the marginal likelihood (reference's Adam loop, 25 steps)
            la.optimize_prior_precision(method="marglik", n_steps=25, lr=0.1, prior_structure="scalar")
            assert rel(la.prior_precision.cpu().numpy(), g["kron_opt_pp_scalar"]) < 2e-4
            la.optimize_prior_precision(method="marglik", n_steps=25, lr=0.1, prior_structure="layerwise")
            assert rel(la.prior_precision.cpu().numpy(), g["kron_opt_pp_layerwise"]) < 2e-4
    model.engine.check_async_errors()


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "sage_small_1batch_s0"])
def test_full_laplace_all_weights(name):
    """hessian_structure="full" over all weights on the GPU: the GGN from the HIP Jacobians against the reference's
    FullLaplace (GGNInterface backend) -- H, marginal likelihood, samples, GLM predictive; a three-batch fit adds up."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    idx_all, y_all = torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda()
    loader = lg.TensorBatchLoader(idx_all, y_all, batch_size=int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", "full")
    la.fit(loader)
    assert rel(la.H.cpu().numpy(), g["fullla_H"]) < RTOL
    assert abs(float(la.loss) - float(g["full_loss"])) < RTOL * abs(float(g["full_loss"]))
    for pp, key in ((None, "fullla_marglik_pp1"), (torch.tensor(0.7), "fullla_marglik_pp07")):
        got = float(la.log_marginal_likelihood(prior_precision=pp))
        assert abs(got - float(g[key])) <= 3e-4 * abs(float(g[key])), key
    la.prior_precision = 1.0
    eps, idx = torch.from_numpy(g["pred_eps"]).cuda(), torch.from_numpy(g["pred_idx"]).cuda()
    assert rel(la.sample(eps=eps).cpu().numpy(), g["fullla_samples"]) < 1e-4
    _, f_var = la._glm_predictive_distribution(idx)
    assert rel(f_var.cpu().numpy(), g["fullla_glm_fvar"]) < 1e-4
    assert np.abs(la(idx).cpu().numpy() - g["fullla_glm_probit"]).max() < 1e-4
    la3 = lg.FullLaplace(model, "classification")
    la3.fit(lg.TensorBatchLoader(idx_all, y_all, batch_size=max(1, len(idx_all) // 3 + 1)))
    assert rel(la3.H.cpu().numpy(), la.H.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "sage_small_1batch_s0"])
def test_regression_likelihood(name):
    """likelihood="regression" on the GPU: KFAC with sqrt(2) I seeds and the MSE loss on the device (fused and unfused,
    whole batch and three batches), diagonal / full GGN through the HIP Jacobians, marginal likelihood with sigma_noise
    and the GLM predictive -- against the reference's regression goldens."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    idx_all, y_all = torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["reg_y"]).cuda()
    loader = lg.TensorBatchLoader(idx_all, y_all, batch_size=int(g["batch_size"]))
    la = lg.Laplace(model, "regression", "all", "kron", sigma_noise=0.7)
    la.fit(loader)
    for i, Fs in enumerate(la.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            assert rel(Hm.cpu().numpy(), g[f"reg_kron_{i}_{j}"]) < RTOL, (i, j)
    assert abs(float(la.loss) - float(g["reg_kron_loss"])) < RTOL * float(g["reg_kron_loss"])
    assert abs(float(la.log_marginal_likelihood()) - float(g["reg_kron_marglik"])) < 3e-4 * abs(float(g["reg_kron_marglik"]))
    f_mu, f_var = la(torch.from_numpy(g["pred_idx"]).cuda(), pred_type="glm")
    assert rel(f_mu.cpu().numpy(), g["reg_kron_glm_fmu"]) < RTOL and rel(f_var.cpu().numpy(), g["reg_kron_glm_fvar"]) < 5e-5
    # unfused kernels and a three-batch fit give the same factors
    be = lg.HipGGN(model, "regression")
    _, views, loss = be.engine.new_kfac_buffers()
    third = max(1, len(idx_all) // 3 + 1)
    for s0 in range(0, len(idx_all), third):
        be.kron_accumulate_(views, loss, idx_all[s0:s0 + third], y_all[s0:s0 + third], len(idx_all), fuse=False)
    k3 = be.pack_kron(views)
    la3 = lg.KronLaplace(model, "regression")
    la3.fit(lg.TensorBatchLoader(idx_all, y_all, batch_size=third))
    for Fa, Fb in zip(k3.kfacs, la3.H_facs.kfacs):
        for a, b in zip(Fa, Fb):
            assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    ld = lg.Laplace(model, "regression", "all", "diag", sigma_noise=0.7)
    ld.fit(loader)
    assert rel(ld.H.cpu().numpy(), g["reg_diag_H"]) < RTOL
    assert abs(float(ld.loss) - float(g["reg_diag_loss"])) < RTOL * float(g["reg_diag_loss"])
    assert abs(float(ld.log_marginal_likelihood()) - float(g["reg_diag_marglik"])) < 3e-4 * abs(float(g["reg_diag_marglik"]))
    lf = lg.Laplace(model, "regression", "all", "full")
    lf.fit(loader)
    assert rel(torch.diagonal(lf.H).cpu().numpy(), g["reg_diag_H"]) < RTOL  # diag(J^T J) == sum J^2
    # classification afterwards still works on the same engine (the likelihood is re-bound)
    lc = lg.Laplace(model, "classification", "all", "kron")
    lc.fit(lg.TensorBatchLoader(idx_all, torch.from_numpy(g["train_y"]).cuda(), batch_size=int(g["batch_size"])))
    assert rel(lc.H_facs.kfacs[0][0].cpu().numpy(), g["kron_0_0"]) < RTOL
    model.engine.check_async_errors()


def test_config1_regression_mlp_on_the_gpu():
    """BASELINE configs[0] (examples/regression_example.py: 1-50-1 tanh MLP, 150 points, regression) through the HIP
    path: an MLP is a GCN on a graph without edges (self loops only, propagation = identity).  diag and kron fits against
    the golden the reference's CPU curvlinops backend produced."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, "c1_regression_mlp.npz"))
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    n = X.shape[0]
    model = lg.GCN(1, 50, 1, 2, X, torch.zeros(2, 0, dtype=torch.int64), act="tanh")
    with torch.no_grad():
        for l, conv in enumerate(model.convs):
            conv.lin.weight.copy_(torch.from_numpy(g[f"W{l}"]))
            conv.lin.bias.copy_(torch.from_numpy(g[f"b{l}"]))
    model = model.eval().cuda()
    loader = lg.TensorBatchLoader(torch.arange(n).cuda(), y.cuda(), batch_size=n)
    ld = lg.Laplace(model, "regression", "all", "diag")
    ld.fit(loader)
    assert rel(ld.H.cpu().numpy(), g["diag_H"]) < RTOL
    assert abs(float(ld.loss) - float(g["diag_loss"])) < RTOL * float(g["diag_loss"])
    lk = lg.Laplace(model, "regression", "all", "kron")
    lk.fit(loader)
    assert abs(float(lk.loss) - float(g["kron_loss"])) < RTOL * float(g["kron_loss"])
    assert len(lk.H_facs.kfacs) == int(g["kron_n_blocks"])
    for i, Fs in enumerate(lk.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            assert rel(Hm.cpu().numpy(), g[f"kron_{i}_{j}"]) < RTOL, (i, j)
    f_mu, f_var = lk(torch.arange(5).cuda())  # regression predictive: mean and variance of the linearised model
    assert f_mu.shape == (5, 1) and f_var.shape == (5, 1, 1) and (f_var > 0).all()


def test_two_ranks_share_one_gpu():
    """The N > 1 path with the real HIP backend: two ranks (gloo, CUDA tensors, one device) shard a kron fit into
    (batch, class) units and a diag fit into sample slices, all-reduce one flat buffer and must reproduce the
    reference goldens on every rank (tests/dist_gpu_worker.py)."""
    import socket
    import subprocess
    import sys

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(os.path.dirname(__file__), "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DIST_GPU_OK world=2" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("structure", ["kron", "diag"])
def test_bench_two_rank_line_is_self_evidencing(structure):
    """``bench.py --gpus 2`` rehearsed on this one GPU (gloo; RCCL refuses two ranks per device): the line must carry the
    evidence a SCALE record needs -- backend name, the rank count an all-reduce of ones returns, the all-reduce's own device
    time and the spread of the per-rank accumulate times."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LGNN_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cora", "--structure", structure,
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    d = line["dist"]
    assert d["dist_backend"] == "gloo" and d["ranks_seen"] == 2
    assert d["allreduce_ms"] > 0 and 0 < d["accumulate_ms_min"] <= d["accumulate_ms_max"]
    if structure == "kron":  # [A_0|B_0|A_1|B_1|loss] + the fit's loss scalar
        assert d["allreduce_floats"] == 1433 ** 2 + 64 ** 2 + 64 ** 2 + 7 ** 2 + 2


def test_batched_symeig_matches_float64_eigh():
    """lgnn_symeig_batched behind Kron.decompose: factors of different sizes go through one padded batched solver
    call; eigenvalues against float64 LAPACK, eigenvectors through gauge-free properties (orthonormal,
    reconstruct the factor), rank-deficient and 1 x 1 factors included (symeig clamp semantics, utils.py:193-226)."""
    import laplace_gnn_amd as lg
    from laplace_gnn_amd.matrix import symeig_batched_hip

    gen = torch.Generator().manual_seed(0)
    mats = []
    for n, rank in ((256, 256), (128, 128), (40, 40), (7, 3), (1, 1), (256, 100)):
        R = torch.randn(max(rank, 1), n, generator=gen)
        mats.append((R.T @ R / rank).cuda())
    for (lam, Q), H in zip(symeig_batched_hip(mats), mats):
        n = H.shape[0]
        ref = torch.linalg.eigvalsh(H.double().cpu()).clamp(min=0).numpy()
        assert lam.shape == (n,) and Q.shape == (n, n)
        assert np.abs(lam.cpu().numpy() - ref).max() < 2e-5 * ref.max()
        assert (lam >= 0).all() and (lam[1:] >= lam[:-1]).all()
        Qd = Q.double().cpu()
        assert (Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max() < 1e-4
        recon = (Qd * lam.double().cpu()) @ Qd.T
        assert rel(recon.numpy(), H.double().cpu().numpy()) < 1e-5
    # 256 < n <= 512 (round 4): the streaming tridiagonalisation of the first n - 256 columns in front of the register-resident
    # kernel -- full rank, rank deficient (a block of exactly dependent columns), a decoupled diagonal block (tau = 0 columns),
    # n = 260 (a 4-column streaming phase) and a partial last panel (n = 300)
    big = []
    for n, rank in ((512, 512), (512, 300), (384, 384), (260, 260), (300, 120)):
        R = torch.randn(max(rank, 1), n, generator=gen) * torch.logspace(0, -2, n).unsqueeze(0)
        big.append((R.T @ R / rank).cuda())
    Hd = torch.zeros(512, 512)
    Hd[:100, :100] = torch.diag(torch.rand(100, generator=gen) + 0.1)
    Rd = torch.randn(600, 412, generator=gen)
    Hd[100:, 100:] = Rd.T @ Rd / 600
    big.append(Hd.cuda())
    for H in big:
        (lam, Q), = symeig_batched_hip([H])
        n = H.shape[0]
        ref = torch.linalg.eigvalsh(H.double().cpu()).clamp(min=0).numpy()
        assert lam.shape == (n,) and Q.shape == (n, n)
        assert np.abs(lam.cpu().numpy() - ref).max() < 1e-5 * ref.max(), n
        assert (lam >= 0).all() and (lam[1:] >= lam[:-1]).all()
        Qd = Q.double().cpu()
        assert (Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max() < 1e-4, n
        recon = (Qd * lam.double().cpu()) @ Qd.T
        assert rel(recon.numpy(), H.double().cpu().numpy()) < 1e-5, n
    # Kron.decompose uses it for CUDA factors and keeps the [[B, A], [B]] sharing
    K = lg.Kron([[mats[2], mats[1]], [mats[2].clone()], [mats[3], mats[2]], [mats[3].clone()]])
    dec = K.decompose()
    assert dec.eigenvalues[1][0] is dec.eigenvalues[0][0] and dec.eigenvalues[3][0] is dec.eigenvalues[2][0]
    dense = K.to_matrix().double().cpu() + 0.5 * torch.eye(40 * 128 + 40 + 7 * 40 + 7, dtype=torch.float64)
    got = float((dec + torch.tensor(0.5, device="cuda")).logdet())
    assert abs(got - float(torch.logdet(dense))) < 1e-4 * abs(float(torch.logdet(dense)))


def test_backend_kron_returns_fresh_tensors_and_reference_layout():
    """CurvatureInterface contract (laplace/curvature/curvlinops.py:55-108): fresh tensors per call,
    [[B,A],[B]] per Linear, A rescaled by M/N, callers may mutate the result."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = model_from_golden(g)
    be = lg.HipGGN(model, "classification")
    assert be.factor == 1.0 and isinstance(be.lossfunc, torch.nn.CrossEntropyLoss)
    assert list(be.params_dict) == ["convs.0.lin.weight", "convs.0.lin.bias", "convs.1.lin.weight",
                                    "convs.1.lin.bias"]
    x = torch.from_numpy(g["train_idx"]).cuda()
    y = torch.from_numpy(g["train_y"]).cuda()
    loss1, k1 = be.kron(x, y, N=len(x))
    k1.kfacs[0][0].mul_(0.0)  # callers mutate what they get
    loss2, k2 = be.kron(x, y, N=len(x))
    assert rel(k2.kfacs[0][0].cpu().numpy(), g["kron_0_0"]) < RTOL
    assert abs(float(loss1) - float(loss2)) < 1e-6 * abs(float(loss2))
    assert [len(F) for F in k2.kfacs] == [2, 1, 2, 1]
    assert k2.kfacs[0][0].shape == (8, 8) and k2.kfacs[0][1].shape == (12, 12)
    # A scales with M/N, B does not (tests/test_curv_backends_curvlinops.py:308-333 pattern)
    _, k3 = be.kron(x, y, N=7 * len(x))
    assert rel(7 * k3.kfacs[0][1].cpu().numpy(), k2.kfacs[0][1].cpu().numpy()) < 1e-5
    assert rel(k3.kfacs[0][0].cpu().numpy(), k2.kfacs[0][0].cpu().numpy()) < 1e-5
    loss_f, H_f = be.full(x, y)  # all weights: GGN from the HIP Jacobians (same matrix as GGNInterface.full)
    assert H_f.shape == (be.engine.n_params,) * 2 and torch.allclose(H_f, H_f.T, atol=1e-5)
    assert abs(float(loss_f) - float(loss2)) < 1e-5 * abs(float(loss2))
    with pytest.raises(NotImplementedError):
        be.jacobians(x, enable_backprop=True)


def test_fit_override_false_accumulates_like_reference():
    """fit twice with override=False: Kron discounting of laplace/baselaplace.py:1589-1607 (B: old + new;
    A: old * n_old/(n_old+n_new) + new * n_new/(n_old+n_new)); loss and n_data add up."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, "gcn_small_3batch_s0.npz"))
    model = model_from_golden(g)
    idx, y = torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=int(g["batch_size"]))
    la = lg.KronLaplace(model, "classification")
    la.fit(loader)
    la.fit(loader, override=False)
    assert la.n_data == 2 * len(idx)
    assert abs(float(la.loss) - 2 * float(g["kron_loss"])) < 1e-4 * 2 * float(g["kron_loss"])
    for i, Fs in enumerate(la.H_facs.kfacs):
        ref = [g[f"kron_{i}_{j}"] for j in range(len(Fs))]
        assert rel(Fs[0].cpu().numpy(), 2 * ref[0]) < RTOL
        if len(Fs) == 2:
            assert rel(Fs[1].cpu().numpy(), ref[1]) < RTOL


def test_weight_update_invalidates_cache():
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = model_from_golden(g)
    idx = torch.from_numpy(g["train_idx"]).cuda()
    out1 = model(idx).clone()
    with torch.no_grad():
        model.convs[1].lin.weight.mul_(2.0)  # in-place update bumps the version counter
    out2 = model(idx)
    assert not torch.equal(out1, out2)


def test_missing_gpu_model_fails_loudly():
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = model_from_golden(g, device=None)  # stays on the CPU
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.arange(3))


def test_invalid_labels_and_indices_are_reported_at_the_end_of_fit():
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = model_from_golden(g)
    idx = torch.from_numpy(g["train_idx"]).cuda()
    y = torch.from_numpy(g["train_y"]).cuda().clone()
    y[3] = 99  # label outside [0, C)
    la = lg.KronLaplace(model, "classification")
    with pytest.raises(lg._lib.HipLibraryError, match="label"):
        la.fit(lg.TensorBatchLoader(idx, y, 10000))
    bad_idx = idx.clone()
    bad_idx[5] = 10_000  # node id outside [0, N)
    ld = lg.DiagLaplace(model, "classification")
    with pytest.raises(lg._lib.HipLibraryError, match="node index"):
        ld.fit(lg.TensorBatchLoader(bad_idx, torch.from_numpy(g["train_y"]).cuda(), 10000))
    # the context stays usable afterwards
    la.fit(lg.TensorBatchLoader(idx, torch.from_numpy(g["train_y"]).cuda(), 10000))
    assert rel(la.H_facs.kfacs[0][0].cpu().numpy(), g["kron_0_0"]) < RTOL


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "gcn_small_3batch_s1", "gcn_mid_1batch_s0", "gcn_mid_3batch_sym_s1",
                                  "gcn_small_isolated_s0", "sage_small_1batch_s0", "sage_small_3batch_s1", "sage_mid_2batch_s2",
                                  "sage_small_3batch_sym_s3"])
def test_matrix_free_glm_variance_matches_the_jacobian_route_and_the_reference(name):
    """csrc/predictive.hip: diag(J P^-1 J^T) per node without Jacobians, Kronecker (scalar and per-block prior) and diagonal
    posterior, against (a) this package's Jacobian route (KronDecomposed.inv_square_form / the diagonal einsum) and (b) the
    reference's f_var golden."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    x = torch.from_numpy(g["pred_idx"]).cuda()
    for cls, key in ((lg.KronLaplace, "kron"), (lg.DiagLaplace, "diag")):
        la = cls(model, "classification")
        la.fit(loader)
        for pp in (1.0, torch.tensor([0.5, 2.0, 1.5, 0.25]) if cls is lg.KronLaplace else 0.7):
            la.prior_precision = pp
            fast = la._glm_variance_matrix_free(x)
            assert fast is not None
            f_mu, f_vd = fast
            Js, f_j = la.backend.jacobians(x)
            ref = torch.diagonal(la.functional_variance(Js), dim1=1, dim2=2)
            assert rel(f_mu.cpu().numpy(), f_j.cpu().numpy()) < 1e-6
            assert rel(f_vd.cpu().numpy(), ref.cpu().numpy()) < 1e-4, (key, pp)
        la.prior_precision = 1.0
        f_mu, f_vd = la._glm_variance_matrix_free(x)
        gold = np.diagonal(g[key + "_glm_fvar"], axis1=1, axis2=2)
        assert rel(f_vd.cpu().numpy(), gold) < 1e-4, key
        assert rel(la(x, link_approx="probit").cpu().numpy(), g[key + "_glm_probit"]) < 1e-4
    model.engine.check_async_errors()


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "gcn_mid_3batch_sym_s1", "gcn_small_isolated_s0",
                                  "sage_small_3batch_s1", "sage_mid_2batch_s2"])
def test_laplace_bridge_without_jacobians_matches_the_reference(name):
    """link_approx "bridge" / "bridge_norm" read the diagonal, the row sums and the total of the C x C predictive covariance
    (laplace/baselaplace.py:637-661): one matrix-free pass over 2 C + 1 linear outputs (lgnn_glm_variance_mapped) gives them.
    Against (a) the reference's f_var golden reduced the same way, (b) the reference's bridge / bridge_norm outputs, (c) this
    package's Jacobian route with a per-block prior.  fp32 <= 1e-4 relative."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    x = torch.from_numpy(g["pred_idx"]).cuda()
    for cls, key in ((lg.KronLaplace, "kron"), (lg.DiagLaplace, "diag")):
        la = cls(model, "classification")
        la.fit(loader)
        moments = la._bridge_moments_matrix_free(x)
        assert moments is not None
        f_mu, diag, rows, total = (t.cpu().numpy() for t in moments)
        gold = g[key + "_glm_fvar"]
        assert rel(f_mu, g[key + "_glm_fmu"]) < 1e-5
        assert rel(diag, np.diagonal(gold, axis1=1, axis2=2)) < 1e-4, key
        assert rel(rows, gold.sum(-1)) < 1e-4 and rel(total, gold.sum((1, 2))) < 1e-4, key
        for link in ("bridge", "bridge_norm"):
            assert rel(la(x, link_approx=link).cpu().numpy(), g[f"{key}_glm_{link}"]) < 1e-4, (key, link)
        # per-block / per-parameter prior: against the Jacobian route (what a backend without the mapped call would take)
        la.prior_precision = torch.tensor([0.5, 2.0, 1.5, 0.25]) if cls is lg.KronLaplace else 0.3
        fast = {link: la(x, link_approx=link) for link in ("bridge", "bridge_norm")}
        la._bridge_moments_matrix_free = lambda x: None
        for link, out in fast.items():
            assert rel(out.cpu().numpy(), la(x, link_approx=link).cpu().numpy()) < 1e-4, (key, link)
        # samples of N(f_mu, diag f_var) read the diagonal only: the same draws through both routes
        eps = torch.from_numpy(g["glm_eps"]).cuda()
        a = la(x, link_approx="mc", diagonal_output=True, eps=eps)
        f_mu_j, f_var_j = la._glm_predictive_distribution(x)
        b = la._glm_predictive_samples(f_mu_j, f_var_j, eps.shape[1], True, None, eps).mean(dim=0)
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    model.engine.check_async_errors()


@pytest.mark.parametrize("name,rows", [("gcn_mid_1batch_s0", 1), ("gcn_small_3batch_s1", 100), ("sage_mid_2batch_s2", 19)])
def test_variances_of_an_arbitrary_linear_map_of_the_logits(name, rows):
    """lgnn_glm_variance_mapped with a random E [rows, C] (fewer and more rows than classes): diag(E S E^T) against the Jacobian
    route's full covariance S, Kronecker (per-block prior) and diagonal posterior; the logits it returns stay the model's."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    x = torch.from_numpy(g["pred_idx"]).cuda()
    for cls in (lg.KronLaplace, lg.DiagLaplace):
        la = cls(model, "classification")
        la.fit(loader)
        la.prior_precision = torch.tensor([0.5, 2.0, 1.5, 0.25]) if cls is lg.KronLaplace else 0.7
        E = torch.randn(rows, la.n_outputs, generator=torch.Generator().manual_seed(rows)).cuda()
        f_mu, v = la._glm_variance_matrix_free(x, out_map=E)
        Js, f_j = la.backend.jacobians(x)
        S = la.functional_variance(Js).double()
        ref = torch.einsum("rc,mcd,rd->mr", E.double(), S, E.double())
        assert v.shape == (x.numel(), rows)
        assert rel(v.cpu().numpy(), ref.cpu().numpy()) < 1e-4, cls.__name__
        assert rel(f_mu.cpu().numpy(), f_j.cpu().numpy()) < 1e-6
    with pytest.raises(ValueError, match="out_map"):
        model.engine.glm_variance(x, **la._matrix_free_operands(), out_map=torch.ones(3, la.n_outputs + 1).cuda())
    model.engine.check_async_errors()


@pytest.mark.parametrize("name", ["gcn_mid_3batch_sym_s1", "sage_mid_2batch_s2", "gcn_small_isolated_s0"])
def test_full_predictive_covariance_without_jacobians(name, monkeypatch):
    """``link_approx="mc"`` samples N(f_mu, f_var) with the full C x C covariance: beyond the size where Jacobians fit, f_var comes
    from C (C + 1) / 2 polarised variances of one matrix-free pass.  Against the reference's f_var golden and, with the golden's own
    draws, the reference's ``mc`` output; evaluation nodes in several chunks."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda(),
                                  batch_size=int(g["batch_size"]))
    x = torch.from_numpy(g["pred_idx"]).cuda()
    eps = torch.from_numpy(g["glm_eps"]).cuda()
    for cls, key in ((lg.KronLaplace, "kron"), (lg.DiagLaplace, "diag")):
        la = cls(model, "classification")
        la.fit(loader)
        f_mu, f_var = la._glm_covariance_matrix_free(x, budget_floats=20_000)  # a few evaluation nodes per chunk
        assert rel(f_mu.cpu().numpy(), g[key + "_glm_fmu"]) < 1e-5
        # against this fit's own Jacobian route (same factors) and the reference's golden (polarised off-diagonals carry the
        # rounding of three variances: 1e-4)
        assert rel(f_var.cpu().numpy(), la._glm_predictive_distribution(x)[1].cpu().numpy()) < 1e-4, key
        assert rel(f_var.cpu().numpy(), g[key + "_glm_fvar"]) < 1e-4, key
        assert float((f_var - f_var.transpose(1, 2)).abs().max()) == 0.0
        monkeypatch.setattr(type(la), "_JACOBIAN_BYTES_MAX", 0)  # every call counts as "too large for Jacobians"
        mc = la(x, link_approx="mc", n_samples=eps.shape[1], eps=eps)
        assert np.abs(mc.cpu().numpy() - g[key + "_glm_mc"]).max() < 5e-5, key
        monkeypatch.undo()
    model.engine.check_async_errors()


def test_decomposition_groups_small_factors_and_caches_large_ones():
    """Kron.decompose on the GPU: factors of up to 256 rows share one call of the hand-written path, larger ones (a
    Cora-shaped model's 1 433 x 1 433 input covariance) get a call of their own and are served from the CALLER's cache under
    the caller's exact key the second time; every eigenpair against fp64."""
    from laplace_gnn_amd import matrix as mx

    g = torch.Generator().manual_seed(1)
    mats = []
    for n, rank in ((300, 300), (64, 64), (7, 6), (257, 200), (256, 256)):
        G = torch.randn(700, n, generator=g, dtype=torch.float64)
        if rank < n:
            G[:, rank:] = G[:, : n - rank] + G[:, 1: n - rank + 1]
        mats.append((G.T @ G / 700).float().cuda())
    cache, keys = {}, [("k", b) if H.shape[0] > 256 else None for b, H in enumerate(mats)]
    first = mx.symeig_batched_hip(mats, cache, keys)
    assert len(cache) == 2  # the 300 and the 257, under the caller's exact keys
    second = mx.symeig_batched_hip(mats, cache, keys)
    for H, (lam, Q), (lam2, Q2) in zip(mats, first, second):
        ref = torch.linalg.eigvalsh(H.double()).clamp(min=0)
        assert float((lam.double() - ref).abs().max()) <= 2e-6 * float(ref.max())
        assert float(((Q * lam) @ Q.T - H).norm() / H.norm()) < 1e-5
        assert float((Q.T @ Q - torch.eye(H.shape[0], device="cuda")).abs().max()) < 1e-4
        if H.shape[0] > 256:
            assert torch.equal(lam, lam2) and torch.equal(Q, Q2)  # served from the cache
    # no key, no cache: an unkeyed call neither reads nor writes it
    third = mx.symeig_batched_hip(mats, cache, [None] * len(mats))
    assert len(cache) == 2 and float((third[0][0] - first[0][0]).abs().max()) <= 1e-5 * float(first[0][0].max())
    # more factors than side streams (8), sizes 1 and 2 included
    many = []
    for i, n in enumerate((1, 2, 3, 5, 17, 40, 64, 100, 128, 200, 255, 256)):
        G = torch.randn(300, n, generator=g, dtype=torch.float64)
        many.append((G.T @ G / 300).float().cuda())
    for H, (lam, Q) in zip(many, mx.symeig_batched_hip(many)):
        ref = torch.linalg.eigvalsh(H.double()).clamp(min=0)
        assert float((lam.double() - ref).abs().max()) <= 2e-6 * float(ref.max()), H.shape
        assert float(((Q * lam) @ Q.T - H).norm() / H.norm()) < 1e-5, H.shape


def test_kron_laplace_reuses_a_gcn_first_input_covariance_decomposition_under_an_exact_key():
    """F = 300 > 256: A_0 = (T / N) X^T X takes the library solver; a second fit of the same model (weights changed, same X)
    is served from the cache kept on the model object; an in-place change of X, another batch count or
    ``cache_decompositions=False`` are not."""
    import laplace_gnn_amd as lg

    g = torch.Generator().manual_seed(3)
    N, F, H, C = 400, 300, 16, 4
    X = torch.randn(N, F, generator=g)
    ei = torch.randint(0, N, (2, 1500), generator=g)
    torch.manual_seed(0)
    model = lg.GCN(F, H, C, 2, X, ei, symmetric=True).cuda().eval()
    idx = torch.randperm(N, generator=g)[:120].cuda()
    y = torch.randint(0, C, (120,), generator=g).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=50)
    la = lg.KronLaplace(model, "classification")
    la.fit(loader)
    cache = model.__dict__["_lgnn_eig_cache"]
    assert len(cache) == 1
    lamA = la.H.eigenvalues[0][1].clone()
    ref = torch.linalg.eigvalsh(la.H_facs.kfacs[0][1].double()).clamp(min=0)
    assert float((lamA.double() - ref).abs().max()) <= 2e-6 * float(ref.max())
    with torch.no_grad():
        model.convs[0].lin.weight.mul_(1.1)  # the weights do not enter A_0
    la2 = lg.KronLaplace(model, "classification")
    la2.fit(loader)
    assert len(cache) == 1 and torch.equal(la2.H.eigenvalues[0][1], lamA)  # the cached eigenpairs, bit for bit
    assert not torch.equal(la2.H.eigenvalues[0][0], la.H.eigenvalues[0][0])  # B_0 did change
    lg.KronLaplace(model, "classification").fit(lg.TensorBatchLoader(idx, y, batch_size=30))
    assert len(cache) == 2  # another batch count: another matrix, another key
    lg.KronLaplace(model, "classification", cache_decompositions=False).fit(loader)
    assert len(cache) == 2
    # an in-place change of the bound feature tensor is seen by the engine (the X-only caches used to survive)
    before = model(idx).clone()
    model.X.mul_(2.0)
    after = model(idx)
    assert float((after - before).abs().max()) > 1e-3
    la3 = lg.KronLaplace(model, "classification")
    la3.fit(loader)
    assert len(cache) == 3
    ref3 = torch.linalg.eigvalsh(la3.H_facs.kfacs[0][1].double()).clamp(min=0)
    assert float((la3.H.eigenvalues[0][1].double() - ref3).abs().max()) <= 2e-6 * float(ref3.max())
    assert float(ref3.max()) > 3.9 * float(ref.max())


def test_rccl_backend_initialises_and_reduces_on_this_box():
    """The N > 1 path uses ``dist.init_process_group("nccl", device_id=...)`` (RCCL) and one flat all-reduce.  The pool has
    one GPU per box and RCCL refuses two ranks on one device, so the collective itself can only be exercised at world
    size 1 here: init, an all-reduce of the KFAC flat buffer and of a 64 MiB tensor (the in-place branch of
    all_reduce_flat_), a fit inside the group, destroy -- in a child process."""
    import subprocess
    import sys

    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import numpy as np
import laplace_gnn_amd as lg
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(0)
N, F, H, C = 400, 16, 32, 5
model = lg.GCN(F, H, C, 2, torch.randn(N, F, generator=g), torch.randint(0, N, (2, 1500), generator=g), symmetric=True).to(dev)
idx = torch.randperm(N, generator=g)[:200].to(dev); y = torch.randint(0, C, (200,), generator=g).to(dev)
la = lg.KronLaplace(model, "classification"); la.fit(lg.TensorBatchLoader(idx, y, 80), process_group=dist.group.WORLD)
flat, views, loss = model.engine.new_kfac_buffers()
model.engine.kfac_accumulate(idx[:80], y[:80], 200, views, loss)
before = flat.clone(); dist.all_reduce(flat); torch.cuda.synchronize()
assert torch.equal(before, flat)
big = torch.ones(1 << 24, device=dev); small = torch.ones(3, device=dev)
dist.all_reduce(big); dist.all_reduce(small); torch.cuda.synchronize()
assert float(big.sum()) == float(1 << 24) and float(small.sum()) == 3.0
dist.destroy_process_group(); print("rccl ok")
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "sage_small_1batch_s0", "sage3_small_1batch_s0"])
def test_last_layer_jacobians_are_the_last_block_of_the_full_jacobians(name):
    """CurvatureInterface.last_layer_jacobians (laplace/curvature/curvature.py:132-167): J_n = [I_C (x) phi_n^T | s_n I_C] equals
    the last layer's columns of the reference's all-weights Jacobian golden; the GLM predictive of the last-layer Laplace
    runs on them."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = model_from_golden(g)
    be = lg.HipGGN(model, "classification", last_layer=True)
    x = torch.from_numpy(g["train_idx"]).cuda()
    Js, f = be.last_layer_jacobians(x)
    L = int(g["num_layers"])
    p_ll = g[f"W{L - 1}"].size + g[f"b{L - 1}"].size
    assert rel(Js.cpu().numpy(), g["jac_first_batch"][:, :, -p_ll:]) < RTOL
    assert rel(f.cpu().numpy(), g["f_first_batch"]) < RTOL
    la = lg.Laplace(model, "classification", subset_of_weights="last_layer", hessian_structure="full")
    la.fit(lg.TensorBatchLoader(x, torch.from_numpy(g["train_y"]).cuda(), batch_size=10000))
    probs = la(x[:8], link_approx="probit")
    assert probs.shape == (8, f.shape[1]) and float((probs.sum(-1) - 1).abs().max()) < 1e-5


def test_early_decomposition_of_a_large_input_factor_matches_the_late_one():
    """A GraphSAGE's A_1 is 2 H wide: above 256 rows it is decomposed by a library call, started on a side stream behind the
    first batch (KronLaplace._snapshot_large_input_factors / _finish_accumulate: the input covariances do not depend on the batch).  Same
    eigenvalues, marginal likelihood and predictive as the decomposition after the loop (cache_decompositions=False)."""
    import laplace_gnn_amd as lg

    N, F, H, C, E = 2000, 40, 160, 6, 9000  # A_1: 320 x 320
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, N, (2, E), generator=g)
    X = torch.randn(N, F, generator=g)
    torch.manual_seed(2)
    model = lg.GraphSAGE(F, H, C, 2, X, ei, symmetric=True).to("cuda").eval()
    idx, y = torch.randperm(N, generator=g)[:700].cuda(), torch.randint(0, C, (700,), generator=g).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=300)  # 300 / 300 / 100
    early = lg.KronLaplace(model, "classification", prior_precision=0.7)
    early.fit(loader)
    late = lg.KronLaplace(model, "classification", prior_precision=0.7, cache_decompositions=False)
    late.fit(loader)
    assert early._early_ok and not late._early_ok
    for Fe, Fl in zip(early.H_facs.kfacs, late.H_facs.kfacs):
        for a, b in zip(Fe, Fl):
            assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    for le, ll in zip(early.H.eigenvalues, late.H.eigenvalues):
        for a, b in zip(le, ll):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    for (Qe, le), (Ql, ll), Fs in zip(zip(early.H.eigenvectors, early.H.eigenvalues), zip(late.H.eigenvectors, late.H.eigenvalues),
                                      early.H_facs.kfacs):
        for Q, lam, Hm in zip(Qe, le, Fs):  # the early eigenpairs reconstruct the FINAL factor
            assert rel(((Q * lam) @ Q.T).cpu().numpy(), Hm.cpu().numpy()) < 2e-5
    assert abs(float(early.log_marginal_likelihood()) - float(late.log_marginal_likelihood())) <= 5e-6 * abs(float(late.log_marginal_likelihood()))
    x = torch.arange(50).cuda()
    assert rel(early(x).cpu().numpy(), late(x).cpu().numpy()) < 1e-4  # (two fp32 decompositions: the small eigenvalues' noise)
    # override=False adds to fitted factors: nothing is known early, the late path runs
    early.fit(loader, override=False)
    assert not early._early_ok


def test_diag_fit_as_one_graph_replay_equals_the_ordinary_fit():
    """``DiagLaplace.fit_graph``: a repeated, unchanged fit is captured into a hipGraph on its second call and replayed after:
    H / loss / marginal likelihood equal the ordinary path's; an in-place weight update is seen by the next replay (the kernels
    read the weights; the forward pass is part of the graph); a different loader or re-bound parameters fall back."""
    import laplace_gnn_amd as lg

    g = np.load(os.path.join(GOLDEN, "gcn_mid_3batch_sym_s1.npz"))
    model = model_from_golden(g)
    idx, y = torch.from_numpy(g["train_idx"]).cuda(), torch.from_numpy(g["train_y"]).cuda()
    loader = lg.TensorBatchLoader(idx, y, batch_size=int(g["batch_size"]))
    ref = lg.DiagLaplace(model, "classification")
    ref.fit(loader)
    H0, l0 = ref.H.clone(), float(ref.loss)
    la = lg.DiagLaplace(model, "classification")
    la.fit_graph = True
    for k in range(4):  # ordinary, capture + replay, replay, replay
        la.fit(loader)
        assert rel(la.H.cpu().numpy(), g["diag_H"]) < 1e-4, k
        assert rel(la.H.cpu().numpy(), H0.cpu().numpy()) < 1e-6 and abs(float(la.loss) - l0) <= 1e-6 * abs(l0), k
        assert la.n_data == int(g["n_data"])
    assert la._fit_graph_state["graph"] is not None and not la._fit_graph_state["off"]
    assert abs(float(la.log_marginal_likelihood()) - float(ref.log_marginal_likelihood())) <= 1e-6 * abs(float(ref.log_marginal_likelihood()))
    with torch.no_grad():
        model.convs[0].lin.weight.mul_(1.5)  # in place: same storage, the replay reads the new values
    la.fit(loader)
    ref.fit(loader)
    assert rel(la.H.cpu().numpy(), ref.H.cpu().numpy()) < 1e-6 and rel(la.H.cpu().numpy(), H0.cpu().numpy()) > 1e-3
    assert rel(la.mean.cpu().numpy(), ref.mean.cpu().numpy()) == 0.0
    other = lg.TensorBatchLoader(idx[:100].clone(), y[:100].clone(), batch_size=50)
    la.fit(other)  # a different key: ordinary path (and a fresh warm-up count)
    ref.fit(other)
    assert rel(la.H.cpu().numpy(), ref.H.cpu().numpy()) < 1e-6 and la.n_data == 100
    model.engine.check_async_errors()


def test_symeig_sizes_around_the_hand_written_ranges():
    """n = 257 .. 515 around the edges of the streaming route (256 < n <= 512, n % 4 == 0): sizes that are not a multiple of 4 and
    n > 512 take the library; every size returns the same decomposition semantics (ascending, clamped, orthonormal)."""
    from laplace_gnn_amd.matrix import symeig_batched_hip

    gen = torch.Generator().manual_seed(3)
    for n in (257, 258, 260, 264, 288, 508, 512, 513, 516):
        R = torch.randn(n + 50, n, generator=gen) * torch.logspace(0, -2, n).unsqueeze(0)
        H = (R.T @ R / (n + 50)).cuda()
        (lam, Q), = symeig_batched_hip([H])
        ref = torch.linalg.eigvalsh(H.double().cpu()).clamp(min=0).numpy()
        assert np.abs(lam.cpu().numpy() - ref).max() < 1e-5 * ref.max(), n
        Qd = Q.double().cpu()
        assert (Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max() < 1e-4, n
        assert rel(((Qd * lam.double().cpu()) @ Qd.T).numpy(), H.double().cpu().numpy()) < 1e-5, n
    # several mid-size factors in one fit-like call (one C call each), twice: nothing left over between calls
    mats = [((torch.randn(600, n, generator=gen).T @ torch.randn(600, n, generator=gen)) / 600) for n in (320, 512, 384)]
    mats = [(M + M.T).cuda() @ (M + M.T).cuda().T / 4 for M in mats]
    a = symeig_batched_hip(mats)
    b = symeig_batched_hip(mats)
    for (la_, Qa), (lb, Qb), H in zip(a, b, mats):
        assert torch.equal(la_, lb) and torch.equal(Qa, Qb)
        ref = torch.linalg.eigvalsh(H.double().cpu()).clamp(min=0).numpy()
        assert np.abs(la_.cpu().numpy() - ref).max() < 1e-5 * ref.max()


def test_graphsage_neighbour_sampling_with_an_explicit_seed():
    """``num_sampled_nodes_per_hop`` (gnn/models/models.py:54-58, gnn/models/utils.py:115-131): the reference draws an unseeded
    random subgraph at every forward call -- parity unpinned by construction.  The seeded draw here: every row keeps
    min(deg, k) of its own neighbours, the same seed gives the same graph, another seed another one, and the model equals one
    built directly on the sampled edge list (forward, KFAC factors)."""
    import laplace_gnn_amd as lg

    gen = torch.Generator().manual_seed(5)
    N, F, H, C, E, k = 300, 10, 16, 4, 2500, 3
    ei = torch.randint(0, N, (2, E), generator=gen)
    X = torch.randn(N, F, generator=gen)
    with pytest.raises(NotImplementedError, match="sample_seed"):
        lg.GraphSAGE(F, H, C, 2, X, ei, num_sampled_nodes_per_hop=k)
    torch.manual_seed(0)
    full = lg.GraphSAGE(F, H, C, 2, X, ei, symmetric=True).cuda().eval()
    fr, fc = full.engine.export_adj()
    deg = torch.bincount(fr, minlength=N)
    torch.manual_seed(0)
    m1 = lg.GraphSAGE(F, H, C, 2, X, ei, symmetric=True, num_sampled_nodes_per_hop=k, sample_seed=7).cuda().eval()
    r1, c1 = m1.engine.export_adj()
    assert torch.equal(torch.bincount(r1, minlength=N), deg.clamp(max=k))  # min(deg, k) per row
    fkeys = set((fr * N + fc).cpu().tolist())
    assert all(int(q) in fkeys for q in (r1 * N + c1).cpu().tolist())       # a subgraph of the full adjacency
    torch.manual_seed(0)
    m2 = lg.GraphSAGE(F, H, C, 2, X, ei, symmetric=True, num_sampled_nodes_per_hop=k, sample_seed=7).cuda().eval()
    r2, c2 = m2.engine.export_adj()
    assert torch.equal(r1, r2) and torch.equal(c1, c2)
    m2.resample(8)
    r3, c3 = m2.engine.export_adj()
    assert torch.equal(torch.bincount(r3, minlength=N), deg.clamp(max=k)) and not (torch.equal(r1, r3) and torch.equal(c1, c3))
    # every neighbour is kept about k / deg of the time: over 40 seeds the kept fraction of the rows with deg = 2 k is ~ 1/2
    rows2k = (deg == 2 * k).nonzero().squeeze(1)
    if rows2k.numel() >= 5:
        kept = 0
        for s in range(40):
            m2.resample(100 + s)
            rs, _ = m2.engine.export_adj()
            kept += int(torch.isin(rs, rows2k).sum())
        frac = kept / (40 * rows2k.numel() * 2 * k)
        assert abs(frac - 0.5) < 1e-9  # exactly k of 2 k per row, always
    torch.manual_seed(0)
    direct = lg.GraphSAGE(F, H, C, 2, X, torch.stack([r1.cpu(), c1.cpu()]), symmetric=False).cuda().eval()
    idx = torch.arange(N).cuda()
    assert torch.equal(m1(idx), direct(idx))
    y = torch.randint(0, C, (120,), generator=gen).cuda()
    loader = lg.TensorBatchLoader(idx[:120], y, batch_size=50)
    la, lb = lg.KronLaplace(m1, "classification"), lg.KronLaplace(direct, "classification")
    la.fit(loader)
    lb.fit(loader)
    for Fa, Fb in zip(la.H_facs.kfacs, lb.H_facs.kfacs):
        for a_, b_ in zip(Fa, Fb):
            assert rel(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-5
