"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
ctypes table matches the header, and the host logic (Kron container, loaders, Laplace front with an
injected oracle backend) behaves like the reference's caller."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle_backend import CpuForwardGCN, OracleBackend

import laplace_gnn_amd as lg

RTOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ---- C ABI --------------------------------------------------------------------------------------
def _header_symbols():
    text = open(os.path.join(ROOT, "include", "laplace_gnn_hip.h")).read()
    return sorted(set(re.findall(r"LGNN_API[^;(]*?\b(lgnn_[a-z_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_header_symbol():
    if not os.path.exists(lg._lib.LIB_PATH):
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True)
    lib = lg._lib.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/laplace_gnn_hip.h but not exported"
    assert sorted(lg._lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.lgnn_abi_version() == 1
    assert isinstance(lib.lgnn_last_error(), bytes)
    out = subprocess.run(["nm", "-D", "--defined-only", lg._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(re.findall(r" T (lgnn_[a-z_]+)", out))
    assert exported == syms, "the .so exports exactly the header's entry points (hidden visibility otherwise)"


def test_header_cites_reference_for_every_compute_entry_point():
    text = open(os.path.join(ROOT, "include", "laplace_gnn_hip.h")).read()
    for needle in ("curvlinops.py:77-108", "curvature.py:412-432", "curvature.py:132-167", "base_gnn.py:136-161",
                   "gnn/utils.py:325-330", "gnn/utils.py:333-336", "utils.py:106-112"):
        assert needle in text, needle


def test_kfac_plan_decisions_without_a_gpu():
    """lgnn_kfac_plan is a host-only query: the kernel choices of the KFAC path per shape, incl. the 4 GiB plane bound of
    the 256-wide fused kernel (a products-shaped GraphSAGE plane is 5 GB: SpMM + Gram through HBM, pong planes needed)
    -- decided without allocating anything."""
    from laplace_gnn_amd.engine import kfac_plan

    arxiv = kfac_plan("gcn", [128, 256, 40], 169_343, 2_500_000, paths=False)
    assert arxiv == {"seeds_on_the_fly": True, "sage_compact": False, "need_pong": False, "classes_per_chunk": 40,
                     "fused": [False, True], "backgemm": [False, True], "paths": False}
    # the headline shape's default: B_0 from the batch's two-hop paths (csrc/paths.hip) -- no planes, no fused kernel
    assert kfac_plan("gcn", [128, 256, 40], 169_343, 2_500_000, paths=True) == {
        "seeds_on_the_fly": True, "sage_compact": False, "need_pong": False, "classes_per_chunk": 40,
        "fused": [False, False], "backgemm": [False, False], "paths": True}
    for dims in ([128, 128, 40], [128, 256, 70], [128, 256, 256, 40], [128, 254, 40]):  # width, classes, depth, alignment
        assert not kfac_plan("gcn", dims, 169_343, 2_500_000, paths=True)["paths"], dims
    # GraphSAGE, same widths: the same fused kernel over ONE-hop paths (no compact top level, no fused LIST kernel)
    assert kfac_plan("sage", [128, 256, 40], 169_343, 2_500_000) == {
        "seeds_on_the_fly": False, "sage_compact": False, "need_pong": False, "classes_per_chunk": 40,
        "fused": [False, False], "backgemm": [False, False], "paths": True}
    assert not kfac_plan("gcn", [128, 256, 40], 169_343, 2_500_000, act="tanh", paths=True)["paths"]
    assert kfac_plan("gcn", [128, 256, 40], 169_343, 2_500_000, fuse=False)["need_pong"]
    sage = kfac_plan("sage", [128, 256, 40], 169_343, 2_500_000, paths=False)
    assert sage["sage_compact"] and sage["fused"] == [False, True] and not sage["need_pong"]
    prod = kfac_plan("sage", [100, 256, 256, 47], 2_449_029, 123_000_000)
    assert prod["fused"] == [False, False, False] and prod["need_pong"] and not prod["sage_compact"]
    assert 1 <= prod["classes_per_chunk"] < 47  # 2 x 5 GB per class plane pair under the 32 GiB default cap
    # the same widths on a graph whose planes stay below 4 GiB are fused
    mid = kfac_plan("sage", [100, 256, 256, 47], 1_000_000, 50_000_000)
    assert mid["fused"] == [False, True, True] and mid["sage_compact"]
    # 128-wide planes use 64-bit addressing: no size bound
    assert kfac_plan("sage", [100, 128, 47], 9_000_000, 10_000_000)["fused"] == [False, True]
    # width % 4 != 0 and C > 64 leave the fused / on-the-fly paths
    odd = kfac_plan("gcn", [16, 30, 70], 1000, 5000)
    assert odd["fused"] == [False, False] and odd["need_pong"] and not odd["seeds_on_the_fly"]
    with pytest.raises(lg._lib.HipLibraryError):
        kfac_plan("gcn", [4] * 11, 10, 10)  # more than 8 layers


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(lg._lib, "_lib", None)
    monkeypatch.setattr(lg._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lg._lib.HipLibraryError, match="no CPU fallback"):
        lg._lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "laplace-gnn_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*"), recursive=True):
        if path.endswith((".py", ".hip", ".h", ".cpp")):
            src = open(path).read()
            assert "gnn_laplace_oracle" not in src and "oracle/" not in src, path


# ---- Kron container (laplace tests/test_matrix.py patterns) ------------------------------------------
def _rand_kron(seed=0):
    g = torch.Generator().manual_seed(seed)

    def psd(n):
        a = torch.randn(n, n + 3, generator=g)
        return a @ a.T

    return lg.Kron([[psd(4), psd(6)], [psd(4)], [psd(3), psd(4)], [psd(3)]])


def test_kron_add_mul_diag_to_matrix():
    k1, k2 = _rand_kron(0), _rand_kron(1)
    s = k1 + k2
    for Fs, Fa, Fb in zip(s.kfacs, k1.kfacs, k2.kfacs):
        for h, a, b in zip(Fs, Fa, Fb):
            assert torch.allclose(h, a + b)
    m = k1 * 0.25
    assert torch.allclose(m.kfacs[0][0], 0.5 * k1.kfacs[0][0])  # sqrt(0.25) per factor of a 2-block
    assert torch.allclose(m.kfacs[1][0], 0.25 * k1.kfacs[1][0])
    assert torch.allclose(m.to_matrix(), 0.25 * k1.to_matrix(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(k1.diag(), k1.to_matrix().diag(), rtol=1e-5)
    assert torch.allclose(k1.logdet(), torch.logdet(k1.to_matrix()), rtol=1e-4)
    with pytest.raises(ValueError):
        k1 + 3
    with pytest.raises(ValueError):
        k1 * "x"


def test_kron_decompose_matches_dense():
    k = _rand_kron(2)
    kd = k.decompose()
    assert torch.allclose(kd.to_matrix(), k.to_matrix(), rtol=1e-3, atol=1e-3)
    delta = torch.tensor(0.7)
    dense = k.to_matrix() + 0.7 * torch.eye(k.to_matrix().shape[0])
    assert torch.allclose((kd + delta).logdet(), torch.logdet(dense), rtol=1e-4)
    assert torch.allclose((kd * 2.0 + delta).logdet(), torch.logdet(2 * k.to_matrix() + 0.7 * torch.eye(dense.shape[0])),
                          rtol=1e-4)
    # symeig clamps at zero like laplace/utils/utils.py:221-224
    L, W = lg.symeig(torch.tensor([[1.0, 0.0], [0.0, -1e-3]]))
    assert float(L.min()) == 0.0


def test_kron_init_from_model_layout():
    lin = torch.nn.Linear(5, 3)
    k = lg.Kron.init_from_model([lin.weight, lin.bias], "cpu")
    assert [tuple(h.shape) for F in k.kfacs for h in F] == [(3, 3), (5, 5), (3, 3)]


def test_kron_tied_hint_follows_the_arithmetic(monkeypatch):
    """A bias block repeats its weight block's B factor; ``decompose`` may skip comparing the tensors (a device sync on the
    GPU) only where the container KNOWS they are equal: zero init, sums of hinted containers, scaling by exactly 1."""
    l0, l1 = torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)
    params = [l0.weight, l0.bias, l1.weight, l1.bias]
    z = lg.Kron.init_from_model(params, "cpu")
    assert z._tied == {1, 3}
    B0, B1 = torch.eye(3) * 2, torch.eye(2) * 3
    hinted = lg.Kron([[B0.clone(), torch.eye(5)], [B0.clone()], [B1.clone(), torch.eye(3)], [B1.clone()]], tied=[1, 3])
    plain = lg.Kron([[B0.clone(), torch.eye(5)], [B0.clone()], [B1.clone(), torch.eye(3)], [B1.clone()]])
    assert (z + hinted)._tied == {1, 3} and (z + plain)._tied == frozenset() and (hinted + plain)._tied == frozenset()
    assert (hinted * 1.0)._tied == {1, 3} and (hinted * 0.5)._tied == frozenset()  # sqrt(0.5) B vs 0.5 B
    calls = []
    real_equal = torch.equal
    monkeypatch.setattr(torch, "equal", lambda a, b: calls.append(1) or real_equal(a, b))
    d_h, w_h = (z + hinted)._distinct_factors()
    assert not calls and len(d_h) == 4 and w_h == [[0, 1], [0], [2, 3], [2]]
    d_p, w_p = plain._distinct_factors()
    assert len(calls) == 2 and len(d_p) == 4 and w_p == w_h  # equal contents are still found, by comparison
    kd = (z + hinted).decompose()
    assert torch.allclose(kd.to_matrix(), plain.to_matrix(), atol=1e-5)


def test_decomposition_cache_is_keyed_exactly_and_owned_by_the_caller():
    """The decomposition of a large factor is reused only under an EXACT key the caller supplies (KronLaplace: identity and
    version of the feature tensor, batch count, N_train, likelihood factor) -- never by comparing matrix contents, never
    from module-global state."""
    from laplace_gnn_amd import engine as en
    from laplace_gnn_amd import matrix as mx

    assert not hasattr(mx, "_LARGE_EIG_CACHE") and not hasattr(mx, "_cached_large_symeig")
    X, Y = torch.zeros(3, 2), torch.zeros(3, 2)
    k1, k2 = en._IdentityKey(X), en._IdentityKey(X)
    assert k1 == k2 and hash(k1) == hash(k2) and k1 != en._IdentityKey(Y)  # identity, not (elementwise) equality
    store = {("gcn_A0", (k1, X._version, (3, 2)), 3, 33, 1.0): "pair"}
    assert ("gcn_A0", (k2, X._version, (3, 2)), 3, 33, 1.0) in store
    X.add_(1.0)  # an in-place change bumps the version: the old key no longer matches
    assert ("gcn_A0", (k2, X._version, (3, 2)), 3, 33, 1.0) not in store
    # without an engine that can name its feature tensor (here: the CPU stand-in) nothing is cached
    g = np.load(os.path.join(GOLDEN, "gcn_small_3batch_s1.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]), int(g["batch_size"]))
    la = lg.KronLaplace(model, "classification", backend=lg.HipGGN)
    la.fit(loader)
    assert la._decompose_cache(loader, True) == (None, None) and "_lgnn_eig_cache" not in model.__dict__
    assert lg.KronLaplace(model, "classification", cache_decompositions=False).cache_decompositions is False


def test_assigning_kfacs_drops_the_tied_hint_and_regression_state_round_trips(tmp_path):
    """``load_state_dict`` used to keep init_from_model's 'zeros equal zeros' hint and reuse the weight
    block's eigenpairs of B for the bias block; for a regression fit the two differ (sqrt(.5) B vs .5 B)."""
    k = lg.Kron.init_from_model([torch.zeros(3, 2), torch.zeros(3)], "cpu")
    assert k._tied == frozenset({1})
    k.kfacs = [[torch.eye(3), torch.eye(2)], [2 * torch.eye(3)]]
    assert k._tied == frozenset()
    kd = k.decompose()
    assert torch.allclose(kd.eigenvalues[1][0], torch.full((3,), 2.0))
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["reg_y"]), int(g["batch_size"]))
    la = lg.Laplace(model, "regression", "all", "kron", backend=OracleBackend, sigma_noise=0.7)
    la.fit(loader)
    path = tmp_path / "reg_kron.pt"
    torch.save(la.state_dict(), path)
    lb = lg.Laplace(model, "regression", "all", "kron", backend=OracleBackend, sigma_noise=0.7)
    lb.load_state_dict(torch.load(path, weights_only=True))
    assert abs(float(lb.log_marginal_likelihood()) - float(la.log_marginal_likelihood())) < 1e-5 * abs(float(la.log_marginal_likelihood()))
    for la_l, lb_l in zip(la.H.eigenvalues, lb.H.eigenvalues):
        for a, b in zip(la_l, lb_l):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert abs(float(la.log_marginal_likelihood()) - float(g["reg_kron_marglik"])) < 2e-4 * abs(float(g["reg_kron_marglik"]))


# ---- loaders ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,bs", [(33, 12), (10, 10), (7, 100), (25, 1)])
def test_tensor_batch_loader_has_dataloader_boundaries(n, bs):
    from torch.utils.data import DataLoader, TensorDataset

    idx, y = torch.arange(n) * 3, torch.arange(n) % 4
    ours = list(lg.TensorBatchLoader(idx, y, bs))
    ref = list(DataLoader(TensorDataset(idx, y), batch_size=bs, shuffle=False))
    assert len(ours) == len(ref) == len(lg.TensorBatchLoader(idx, y, bs))
    for (a, b), (c, d) in zip(ours, ref):
        assert torch.equal(a, c) and torch.equal(b, d)
    assert len(lg.TensorBatchLoader(idx, y, bs).dataset) == n


def test_batches_of_rank_partition():
    for world in (1, 2, 3, 8, 16):
        owned = [lg.batches_of_rank(10, r, world) for r in range(world)]
        assert sorted(t for o in owned for t in o) == list(range(10))
    assert [len(lg.batches_of_rank(10, r, 8)) for r in range(8)] == [2, 2, 1, 1, 1, 1, 1, 1]


def test_units_of_rank_is_a_balanced_partition():
    for T, C, world in [(10, 40, 8), (10, 40, 1), (3, 3, 8), (1, 7, 4), (10, 40, 3)]:
        seen = []
        sizes = []
        for r in range(world):
            units = lg.units_of_rank(T, C, r, world)
            sizes.append(sum(c1 - c0 for _, c0, c1 in units))
            assert len({t for t, _, _ in units}) == len(units)  # at most one class range per batch and rank
            seen += [(t, c) for t, c0, c1 in units for c in range(c0, c1)]
        assert sorted(seen) == [(t, c) for t in range(T) for c in range(C)]
        assert max(sizes) - min(sizes) <= 1
    assert [sum(c1 - c0 for _, c0, c1 in lg.units_of_rank(10, 40, r, 8)) for r in range(8)] == [50] * 8


def test_oracle_class_ranges_are_additive():
    import gnn_laplace_oracle as O
    g = np.load(os.path.join(GOLDEN, "sage_small_1batch_s0.npz"))
    for name in ("sage_small_1batch_s0", "gcn_small_1batch_s0"):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        m = _cpu_model(g).oracle_model()
        idx, y, n = g["train_idx"], g["train_y"], len(g["train_idx"])
        full_loss, full = O.kfac_batch(m, idx, y, n)
        parts = [O.kfac_batch(m, idx, y, n, classes=r) for r in ((0, 1), (1, 3))]
        assert abs(sum(float(p[0]) for p in parts) - float(full_loss)) < 1e-6 * abs(float(full_loss))
        for i, Fs in enumerate(full):
            for j, Hm in enumerate(Fs):
                assert rel(sum(p[1][i][j] for p in parts), Hm) < 1e-6


# ---- front-end with an injected oracle backend ------------------------------------------------------------
def _cpu_model(g):
    kind, L = str(g["kind"]), int(g["num_layers"])
    X, ei = torch.from_numpy(g["X"]), torch.from_numpy(g["edge_index"])
    cls = lg.GCN if kind == "gcn" else lg.GraphSAGE
    from golden_utils import constructor_extras
    m = cls(X.shape[1], g["W0"].shape[0], g[f"W{L - 1}"].shape[0], L, X, ei, symmetric=bool(g["symmetric"]),
            **constructor_extras(g))
    with torch.no_grad():
        for l, conv in enumerate(m.convs):
            conv.lin.weight.copy_(torch.from_numpy(g[f"W{l}"]))
            conv.lin.bias.copy_(torch.from_numpy(g[f"b{l}"]))
        for l, lin in enumerate(m.res):
            lin.weight.copy_(torch.from_numpy(g[f"Wr{l}"]))
            lin.bias.copy_(torch.from_numpy(g[f"br{l}"]))
        if m.norm_kind is not None:
            for l, nm in enumerate(m.norms):
                nm.weight.copy_(torch.from_numpy(g[f"norm_w{l}"]))
                nm.bias.copy_(torch.from_numpy(g[f"norm_b{l}"]))
                if m.norm_kind == "batch":
                    nm.running_mean.copy_(torch.from_numpy(g[f"norm_rm{l}"]))
                    nm.running_var.copy_(torch.from_numpy(g[f"norm_rv{l}"]))
    return CpuForwardGCN(m.eval())


@pytest.mark.parametrize("name", ["gcn_small_3batch_s1", "sage_small_3batch_s1", "gcn_small_isolated_s0",
                                  "gcn_resln_small_3batch_s1", "sage_resln_small_3batch_s1", "gcn_bn_small_3batch_s3",
                                  "gcn3_resln_small_3batch_s5"])
def test_fit_loop_with_oracle_backend_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    la = lg.Laplace(model, "classification", subset_of_weights="all", hessian_structure="kron", backend=OracleBackend)
    la.fit(loader)
    assert [c[0] for c in la.backend.calls] == ["kron"] * len(loader)
    for i, Fs in enumerate(la.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            assert rel(Hm.numpy(), g[f"kron_{i}_{j}"]) < RTOL
    assert abs(float(la.loss) - float(g["kron_loss"])) < RTOL * float(g["kron_loss"])
    assert (la.n_data, la.n_outputs, la.n_params) == (int(g["n_data"]), int(g["n_outputs"]), int(g["n_params"]))
    assert isinstance(la.H, lg.KronDecomposed)
    ld = lg.Laplace(model, "classification", "all", "diag", backend=OracleBackend)
    ld.fit(loader)
    assert rel(ld.H.numpy(), g["diag_H"]) < RTOL
    # "next" row 8(f)-1: the reference's log marginal likelihood from the decomposed factors
    def close(a, b):
        return abs(float(a) - float(b)) <= 2e-4 * abs(float(b))
    assert close(la.log_marginal_likelihood(), g["kron_marglik_pp1"])
    assert close(la.log_marginal_likelihood(prior_precision=torch.tensor(0.7)), g["kron_marglik_pp07"])
    assert close(la.log_marginal_likelihood(prior_precision=torch.from_numpy(g["kron_prior_layerwise"])),
                 g["kron_marglik_layerwise"])
    assert close(ld.log_marginal_likelihood(prior_precision=1.0), g["diag_marglik_pp1"])
    assert close(ld.log_marginal_likelihood(prior_precision=torch.tensor(0.7)), g["diag_marglik_pp07"])
    la.prior_precision, ld.prior_precision = 1.0, 1.0
    # marginal likelihood identity (laplace tests/test_baselaplace.py:308-384 pattern), diag case
    pp = 0.7
    ld.prior_precision = pp
    expect = -float(ld.loss) - 0.5 * (float(torch.log(ld.H + pp).sum()) - ld.n_params * np.log(pp)
                                      + pp * float(ld.mean @ ld.mean))
    assert abs(float(ld.log_marginal_likelihood()) - expect) < 1e-3 * abs(expect)
    la.prior_precision = pp
    dense = la.H_facs.to_matrix() + pp * torch.eye(la.n_params)
    expect = -float(la.loss) - 0.5 * (float(torch.logdet(dense.double())) - la.n_params * np.log(pp)
                                      + pp * float(la.mean @ la.mean))
    assert abs(float(la.log_marginal_likelihood()) - expect) < 1e-3 * abs(expect)


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "gcn_mid_3batch_sym_s1", "sage_small_3batch_s1"])
def test_posterior_samples_and_sampling_predictive_match_reference(name):
    """"next" row 8(f)-2: theta = mean + P^-1/2 eps and the Monte-Carlo softmax average the driver evaluates with
    (gnn/marglik_training.py:338-352 -> laplace/baselaplace.py:1183-1199, :1646-1655, :1912-1919); the golden
    values were computed by the reference's own code on the same fixed draws ``eps``."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    eps, idx = torch.from_numpy(g["pred_eps"]), torch.from_numpy(g["pred_idx"])
    before = [p.detach().clone() for p in model.parameters()]
    for structure, key in (("kron", "kron"), ("diag", "diag")):
        la = lg.Laplace(model, "classification", "all", structure, backend=OracleBackend)
        la.fit(loader)
        assert rel(la.sample(eps=eps).numpy(), g[key + "_samples"]) < 2e-4
        py = la(idx, pred_type="nn", link_approx="mc", n_samples=len(eps), eps=eps)
        assert np.abs(py.numpy() - g[key + "_nn_py"]).max() < 2e-5
        assert torch.allclose(py.sum(-1), torch.ones(len(idx)), atol=1e-5)
        # the mean is restored and the parameters do not alias it
        for p, b in zip(model.parameters(), before):
            assert torch.equal(p.detach(), b) and p.data_ptr() != la.mean.data_ptr()
        # generator path: reproducible, right shape, centred on the mean for many draws
        gen = torch.Generator().manual_seed(3)
        s1 = la.sample(64, generator=gen)
        s2 = la.sample(64, generator=torch.Generator().manual_seed(3))
        assert s1.shape == (64, la.n_params) and torch.equal(s1, s2)
        # "next" row 8(f)-3: the GLM predictive (the reference's default la(x)), every link approximation, against
        # the reference's own output on the same inputs (mc: the same fixed draws)
        f_mu, f_var = la._glm_predictive_distribution(idx)
        assert rel(f_mu.numpy(), g[key + "_glm_fmu"]) < 1e-5 and rel(f_var.numpy(), g[key + "_glm_fvar"]) < 2e-4
        assert np.abs(la(idx).numpy() - g[key + "_glm_probit"]).max() < 2e-5  # defaults: glm + probit
        for link in ("bridge", "bridge_norm"):
            assert np.abs(la(idx, pred_type="glm", link_approx=link).numpy() - g[f"{key}_glm_{link}"]).max() < 5e-5
        mc = la(idx, pred_type="glm", link_approx="mc", n_samples=6, eps=torch.from_numpy(g["glm_eps"]))
        assert np.abs(mc.numpy() - g[key + "_glm_mc"]).max() < 5e-5
        with pytest.raises(ValueError):
            la(idx, pred_type="nn", link_approx="probit")
        with pytest.raises(ValueError):
            la(idx, pred_type="gp")


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "sage_small_3batch_s1"])
def test_optimize_prior_precision_matches_reference(name):
    """Post-hoc prior tuning (laplace/baselaplace.py:342-560).  marglik: 25 Adam steps on the log prior precision,
    scalar and per-block, against what the reference's loop returns from the same fitted state; the marginal
    likelihood is differentiable in the prior precision; gridsearch picks the grid value with the lowest
    validation NLL of the predictive."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", "kron", backend=OracleBackend)
    la.fit(loader)
    pp = torch.tensor([0.7, 1.3, 0.9, 2.0], requires_grad=True)
    ml = la.log_marginal_likelihood(prior_precision=pp)
    (grad,) = torch.autograd.grad(ml, pp)
    eps = 1e-2
    for i in range(4):  # central differences
        d = torch.zeros(4); d[i] = eps
        fd = (float(la.log_marginal_likelihood(prior_precision=(pp + d).detach()))
              - float(la.log_marginal_likelihood(prior_precision=(pp - d).detach()))) / (2 * eps)
        assert abs(float(grad[i]) - fd) < 2e-2 * max(1.0, abs(fd))
    la.optimize_prior_precision(method="marglik", n_steps=25, lr=0.1, prior_structure="scalar")
    assert rel(la.prior_precision.numpy(), g["kron_opt_pp_scalar"]) < 1e-3
    la.optimize_prior_precision(method="marglik", n_steps=25, lr=0.1, prior_structure="layerwise")
    assert rel(la.prior_precision.numpy(), g["kron_opt_pp_layerwise"]) < 1e-3
    # the tuned value improves the marginal likelihood over the initial one
    tuned = float(la.log_marginal_likelihood())
    assert tuned > float(la.log_marginal_likelihood(prior_precision=1.0))
    # gridsearch on a validation loader: result lies on the grid and is the argmin of the validation NLL
    val = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"][:12]), torch.from_numpy(g["train_y"][:12]), 6)
    la.optimize_prior_precision(pred_type="glm", method="gridsearch", val_loader=val, grid_size=7,
                                log_prior_prec_min=-2, log_prior_prec_max=2)
    grid = torch.logspace(-2, 2, 7)
    best = float(la.prior_precision)
    assert min(abs(best - float(v)) for v in grid) < 1e-6

    def nll(p):
        la.prior_precision = p
        probs = la(torch.from_numpy(g["train_idx"][:12]))
        return float(torch.nn.functional.nll_loss(probs.log(), torch.from_numpy(g["train_y"][:12])))
    assert abs(nll(best) - min(nll(float(v)) for v in grid)) < 1e-6
    with pytest.raises(ValueError):
        la.optimize_prior_precision(method="gridsearch")
    with pytest.raises(ValueError):
        la.optimize_prior_precision(method="cv")


def test_full_laplace_all_weights_matches_reference():
    """hessian_structure="full" over all weights (laplace/baselaplace.py:1380-1510) against the reference's FullLaplace
    run with its backend-free GGNInterface: H, marginal likelihood, samples, GLM predictive."""
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", "full", backend=OracleBackend)
    assert isinstance(la, lg.FullLaplace)
    la.fit(loader)
    assert rel(la.H.numpy(), g["fullla_H"]) < RTOL
    assert abs(float(la.log_marginal_likelihood()) - float(g["fullla_marglik_pp1"])) < 2e-4 * abs(float(g["fullla_marglik_pp1"]))
    got = float(la.log_marginal_likelihood(prior_precision=torch.tensor(0.7)))
    assert abs(got - float(g["fullla_marglik_pp07"])) < 2e-4 * abs(float(g["fullla_marglik_pp07"]))
    la.prior_precision = 1.0
    eps, idx = torch.from_numpy(g["pred_eps"]), torch.from_numpy(g["pred_idx"])
    assert rel(la.sample(eps=eps).numpy(), g["fullla_samples"]) < 5e-4
    _, f_var = la._glm_predictive_distribution(idx)
    assert rel(f_var.numpy(), g["fullla_glm_fvar"]) < 5e-4
    assert np.abs(la(idx).numpy() - g["fullla_glm_probit"]).max() < 5e-5


def test_state_dict_round_trip(tmp_path):
    """laplace/baselaplace.py:1314-1368, 1664-1677: a fitted Laplace saved with torch.save and loaded into a fresh
    object (safe loader) predicts and scores identically; wrong class / size / likelihood are refused."""
    g = np.load(os.path.join(GOLDEN, "gcn_small_3batch_s1.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]),
                                  int(g["batch_size"]))
    idx = torch.from_numpy(g["pred_idx"])
    for structure in ("kron", "diag", "full"):
        la = lg.Laplace(model, "classification", "all", structure, backend=OracleBackend, prior_precision=0.8)
        la.fit(loader)
        path = tmp_path / f"{structure}.pt"
        torch.save(la.state_dict(), path)
        lb = lg.Laplace(model, "classification", "all", structure, backend=OracleBackend)
        lb.load_state_dict(torch.load(path, weights_only=True))
        assert float(lb.prior_precision) == pytest.approx(0.8) and lb.n_data == la.n_data
        assert abs(float(lb.log_marginal_likelihood()) - float(la.log_marginal_likelihood())) < 1e-4
        assert torch.allclose(lb(idx), la(idx), atol=1e-6)
        other = lg.Laplace(model, "classification", "all", "diag" if structure != "diag" else "kron", backend=OracleBackend)
        with pytest.raises(ValueError):
            other.load_state_dict(la.state_dict())
    with pytest.raises(AttributeError):
        lg.Laplace(model, "classification", "all", "diag", backend=OracleBackend).state_dict()


@pytest.mark.parametrize("name", ["gcn_small_1batch_s0", "sage_small_1batch_s0"])
def test_regression_likelihood_matches_reference(name):
    """likelihood="regression" (MSE, Hessian square root sqrt(2) I, interface factor 0.5): KFAC factors, diagonal
    GGN, marginal likelihood with sigma_noise and the GLM predictive against the reference's KronLaplace /
    DiagLaplace run on float targets."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["reg_y"]), int(g["batch_size"]))
    la = lg.Laplace(model, "regression", "all", "kron", backend=OracleBackend, sigma_noise=0.7)
    la.fit(loader)
    for i, Fs in enumerate(la.H_facs.kfacs):
        for j, Hm in enumerate(Fs):
            assert rel(Hm.numpy(), g[f"reg_kron_{i}_{j}"]) < RTOL
    assert abs(float(la.loss) - float(g["reg_kron_loss"])) < RTOL * float(g["reg_kron_loss"])
    assert abs(float(la.log_marginal_likelihood()) - float(g["reg_kron_marglik"])) < 2e-4 * abs(float(g["reg_kron_marglik"]))
    f_mu, f_var = la(torch.from_numpy(g["pred_idx"]), pred_type="glm")
    assert rel(f_mu.numpy(), g["reg_kron_glm_fmu"]) < 1e-5 and rel(f_var.numpy(), g["reg_kron_glm_fvar"]) < 3e-4
    m_nn, v_nn = la(torch.from_numpy(g["pred_idx"]), pred_type="nn", link_approx="mc", n_samples=8,
                    generator=torch.Generator().manual_seed(0))
    assert m_nn.shape == f_mu.shape and v_nn.shape == f_mu.shape and (v_nn >= 0).all()
    ld = lg.Laplace(model, "regression", "all", "diag", backend=OracleBackend, sigma_noise=0.7)
    ld.fit(loader)
    assert rel(ld.H.numpy(), g["reg_diag_H"]) < RTOL
    assert abs(float(ld.log_marginal_likelihood()) - float(g["reg_diag_marglik"])) < 2e-4 * abs(float(g["reg_diag_marglik"]))


def test_kron_decomposed_bmm_matches_dense_power():
    """laplace tests/test_matrix.py bmm pattern: (Q diag(l + delta)^e Q^T) W against the dense matrix, for
    1-, 2- and 3-dimensional W and exponents -1, -1/2, 1."""
    K = _rand_kron(2)
    dec = K.decompose() + torch.tensor(0.3)
    P = dec.to_matrix().shape[0]
    gen = torch.Generator().manual_seed(0)
    for e in (-1.0, -0.5, 1.0):
        M = dec.to_matrix(exponent=e).double()
        W2 = torch.randn(5, P, generator=gen)
        assert rel(dec.bmm(W2, exponent=e).numpy(), (W2.double() @ M.T).float().numpy()) < 1e-4
        W3 = torch.randn(3, 4, P, generator=gen)
        assert rel(dec.bmm(W3, exponent=e).numpy(), (W3.double() @ M.T).float().numpy()) < 1e-4
        w1 = torch.randn(P, generator=gen)
        assert rel(dec.bmm(w1, exponent=e).numpy(), (M @ w1.double()).float().numpy()) < 1e-4
    with pytest.raises(ValueError):
        dec.bmm(torch.zeros(1, 1, 1, P))


def test_override_false_triples_and_override_true_is_idempotent():
    """tests/test_baselaplace.py:387-426 pattern: fit, fit(override=False), fit(override=False)."""
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]), 10000)
    ld = lg.DiagLaplace(model, "classification", backend=OracleBackend)
    ld.fit(loader)
    H1 = ld.H.clone()
    ld.fit(loader)
    assert torch.allclose(ld.H, H1)
    ld.fit(loader, override=False)
    ld.fit(loader, override=False)
    assert torch.allclose(ld.H, 3 * H1, rtol=1e-5) and ld.n_data == 3 * len(loader.dataset)


def test_factory_keys_and_errors():
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    model = _cpu_model(g)
    assert isinstance(lg.Laplace(model, "classification", "all", "kron", backend=OracleBackend), lg.KronLaplace)
    assert isinstance(lg.Laplace(model, "classification", "all", "diag", backend=OracleBackend), lg.DiagLaplace)
    ll = lg.Laplace(model, "classification", "last_layer", "full", backend=OracleBackend)
    assert isinstance(ll, lg.FullLLLaplace) and ll.n_params == 3 * 8 + 3
    with pytest.raises(NotImplementedError):
        lg.Laplace(model, "classification", "all", "lowrank")
    with pytest.raises(ValueError):
        lg.Laplace(model, "classification", "subnetwork", "kron")
    with pytest.raises(ValueError):
        lg.Laplace(model, "poisson", "all", "kron")
    with pytest.raises(ValueError):
        lg.KronLaplace(model, "classification", sigma_noise=2.0)
    with pytest.raises(TypeError, match="engine"):
        lg.HipGGN(torch.nn.Linear(2, 2), "classification")


def test_stochastic_last_layer_backend_is_refused_not_ignored():
    """HipGGN(stochastic=True, last_layer=True) used to fall through to the deterministic last-layer GGN."""
    g = np.load(os.path.join(GOLDEN, "sage_small_1batch_s0.npz"))
    model = _cpu_model(g)
    with pytest.raises(NotImplementedError, match="stochastic"):
        lg.HipGGN(model, "classification", last_layer=True, stochastic=True)
    lg.HipGGN(model, "classification", last_layer=True)
    lg.HipGGN(model, "classification", stochastic=True)


def test_lastlayer_full_front_with_oracle_backend():
    g = np.load(os.path.join(GOLDEN, "sage_small_1batch_s0.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]), 10000)
    ll = lg.Laplace(model, "classification", "last_layer", "full", backend=OracleBackend)
    ll.fit(loader)
    p = ll.n_params
    assert rel(ll.H.numpy(), g["full_H"][-p:, -p:]) < RTOL
    assert ll.mean.shape[0] == p


def test_model_constructor_validation():
    X = torch.randn(6, 4)
    adj = torch.zeros(6, 6)
    adj[0, 1] = adj[2, 3] = 1
    m = lg.GCN(4, 5, 3, 2, X, adj)
    assert m.edge_index.tolist() == [[0, 2], [1, 3]]
    assert [tuple(p.shape) for p in m.parameters()] == [(5, 4), (5,), (3, 5), (3,)]
    s = lg.GraphSAGE(4, 5, 3, 2, X, adj, None)
    assert [tuple(p.shape) for p in s.parameters()] == [(5, 8), (5,), (3, 10), (3,)]
    with pytest.raises(AssertionError):
        lg.GCN(4, 5, 3, 2, X, adj * 2)
    for kw in (dict(act="gelu"), dict(update_adj=True)):
        with pytest.raises(NotImplementedError):
            lg.GCN(4, 5, 3, 2, X, adj, **kw)
    with pytest.raises(ValueError):
        lg.GCN(4, 5, 3, 2, X, adj, norm="group")  # gnn/models/base_gnn.py:94-95
    # res / norm: the reference's module layout and parameter order (base_gnn.py:86-113; convs.* before res.*)
    r = lg.GraphSAGE(4, 5, 3, 3, X, adj, None, norm="layer", res=True)
    assert [k for k, _ in r.named_parameters()] == [
        "norms.0.weight", "norms.0.bias", "norms.1.weight", "norms.1.bias",
        "convs.0.lin.weight", "convs.0.lin.bias", "convs.1.lin.weight", "convs.1.lin.bias", "convs.2.lin.weight",
        "convs.2.lin.bias", "res.0.weight", "res.0.bias", "res.1.weight", "res.1.bias"]
    assert [tuple(p.shape) for p in r.res.parameters()] == [(5, 4), (5,), (5, 5), (5,)]
    b = lg.GCN(4, 5, 3, 2, X, adj, norm="batch")
    assert isinstance(b.norms[0], torch.nn.BatchNorm1d) and len(b.res) == 0
    with pytest.raises(NotImplementedError):
        lg.GraphSAGE(4, 5, 3, 2, X, adj, 5)


def test_the_flattened_mean_follows_every_kind_of_parameter_write():
    """fit() flattens the parameters into ``mean`` on EVERY call (laplace/baselaplace.py:800): writes through ``p.data``
    bump no version counter, and ``load_state_dict`` / ``la.mean = ...`` replace the vector -- none of them may leave a
    stale or foreign posterior mean behind."""
    g = np.load(os.path.join(GOLDEN, "gcn_small_3batch_s1.npz"))
    model = _cpu_model(g)
    loader = lg.TensorBatchLoader(torch.from_numpy(g["train_idx"]), torch.from_numpy(g["train_y"]), int(g["batch_size"]))
    la = lg.Laplace(model, "classification", "all", "diag", backend=OracleBackend)
    flat = lambda: torch.nn.utils.parameters_to_vector(la.params).detach()  # noqa: E731
    la.fit(loader)
    assert torch.equal(la.mean, flat())
    v0 = [p._version for p in la.params]
    for p in la.params:
        p.data.add_(1.0)  # invisible to the version counters
    assert [p._version for p in la.params] == v0
    la.fit(loader)
    assert torch.equal(la.mean, flat())
    with torch.no_grad():
        model.convs[0].lin.bias.add_(0.25)  # in place: the version counter moves
    la.fit(loader)
    assert torch.equal(la.mean, flat())
    m1 = la.mean
    torch.nn.utils.vector_to_parameters(m1.clone() * 0.5, la.params)  # storage replaced
    la.fit(loader)
    assert torch.equal(la.mean, m1 * 0.5)
    la.mean = torch.zeros_like(la.mean)  # a foreign vector
    la.fit(loader)
    assert torch.equal(la.mean, flat())
    sd = la.state_dict()
    sd["mean"] = sd["mean"] + 3.0
    la.load_state_dict(sd)
    la.fit(loader)
    assert torch.equal(la.mean, flat())


def test_a_reference_checkpoint_with_the_dense_adj_key_loads():
    """VERDICT r3 missing #5: the reference's ``state_dict`` carries the graph as the dense ``adj`` parameter
    (gnn/models/base_gnn.py:75-76); loading it replaces the module's edge list (a GCN's stored self loops included, the
    engine deduplicates them), keeps ``strict`` happy and leaves the Linear parameters where they belong."""
    g = np.load(os.path.join(GOLDEN, "gcn_small_1batch_s0.npz"))
    N = int(g["X"].shape[0])
    adj = np.zeros((N, N), np.float32)
    adj[g["adj_nz_row"], g["adj_nz_col"]] = 1.0  # the reference's model.adj (with the GCN's self loops)
    X = torch.from_numpy(g["X"])
    other = torch.zeros(N, N)
    other[0, 1] = 1.0
    m = lg.GCN(X.shape[1], int(g["W0"].shape[0]), int(g["W1"].shape[0]), 2, X, other)
    sd = {"adj": torch.from_numpy(adj), "convs.0.lin.weight": torch.from_numpy(g["W0"]), "convs.0.lin.bias": torch.from_numpy(g["b0"]),
          "convs.1.lin.weight": torch.from_numpy(g["W1"]), "convs.1.lin.bias": torch.from_numpy(g["b1"])}
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    got = set(map(tuple, m.edge_index.t().tolist()))
    assert got == set(zip(g["adj_nz_row"].tolist(), g["adj_nz_col"].tolist()))
    assert torch.equal(m.convs[0].lin.weight.detach(), torch.from_numpy(g["W0"]))
    # a sparse adjacency and an STE model's continuous values (propagated as adj > threshold) load the same way
    m2 = lg.GCN(X.shape[1], int(g["W0"].shape[0]), int(g["W1"].shape[0]), 2, X, other)
    sd["adj"] = torch.from_numpy(adj * 0.9 + 0.05 * (adj == 0)).to_sparse()
    m2.threshold = 0.5
    m2.load_state_dict(sd, strict=True)
    assert set(map(tuple, m2.edge_index.t().tolist())) == got
    with pytest.raises(RuntimeError):
        lg.GCN(X.shape[1], int(g["W0"].shape[0]), int(g["W1"].shape[0]), 2, X, other).load_state_dict(
            {**sd, "adj": torch.zeros(N + 1, N + 1)})
