"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *reference's own* hot-path Python files from ``/root/reference`` by path so
that golden vectors can be generated from the reference's arithmetic.  It only works
in the build container (``/root/reference`` does not exist on the GPU box); nothing is
copied out of the reference and nothing here ships.

Recipe:
  * the package ``__init__`` files of ``curvlinops`` / ``laplace`` / ``gnn`` pull in
    third-party engines that are not installed here (backpack, asdl, torchmetrics,
    torch_geometric, einconv, opt_einsum).  They are bypassed by pre-seeding
    ``sys.modules`` with empty namespace shells that only carry ``__path__``.
  * names that the GCN/GraphSAGE ``kron``/``diag`` path never *calls* but that are
    imported at module import time get import-only placeholders.
  * the real files (``curvlinops/kfac.py``, ``laplace/curvature/curvature.py`` ...) are
    then imported normally, so every number comes from the reference's own code.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("LAPLACE_GNN_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "curvlinops"))


def _shell(name: str, path: str | None = None) -> types.ModuleType:
    mod = types.ModuleType(name)
    if path is not None:
        mod.__path__ = [path]  # namespace shell: sub-modules resolve, __init__ never runs
    sys.modules[name] = mod
    return mod


def _placeholder_class(name: str):
    return type(name, (), {"__doc__": "import-only placeholder (absent third-party)"})


_LOADED = None


def load():
    """Return a namespace with the reference modules of the hot path."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # never drop __pycache__ into the reference tree
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    import torch
    from torch import nn

    # ---- import-only placeholders for absent third-party names -------------------
    einconv = _shell("einconv", path=None)
    einconv.index_pattern = lambda *a, **k: (_ for _ in ()).throw(
        NotImplementedError("einconv absent (Conv2d KFAC is out of scope)")
    )
    einconv_utils = _shell("einconv.utils")
    einconv_utils.get_conv_paddings = einconv.index_pattern
    einconv.utils = einconv_utils

    oe = _shell("opt_einsum")
    oe.contract = lambda *a, **k: torch.einsum(*a)

    tm = _shell("torchmetrics")

    class Metric(nn.Module):  # only subclassed at import time (RunningNLLMetric)
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, *a, **k):
            pass

    tm.Metric = Metric
    tm.MeanSquaredError = _placeholder_class("MeanSquaredError")

    tg = _shell("torch_geometric")
    tgnn = _shell("torch_geometric.nn")
    tgres = _shell("torch_geometric.nn.resolver")

    def activation_resolver(act, **kw):
        if callable(act) and not isinstance(act, str):
            return act
        table = {"relu": nn.ReLU, "tanh": nn.Tanh, "elu": nn.ELU, None: nn.Identity}
        return table[act](**kw)

    tgres.activation_resolver = activation_resolver
    tg.nn = tgnn
    tgnn.resolver = tgres

    ipdb = _shell("ipdb")
    ipdb.set_trace = lambda *a, **k: None  # live breakpoint at gnn/models/base_gnn.py:109

    # ---- namespace shells for the reference's own packages -----------------------
    cl = _shell("curvlinops", os.path.join(REF_ROOT, "curvlinops"))
    lap = _shell("laplace", os.path.join(REF_ROOT, "laplace"))
    lap_utils = _shell("laplace.utils", os.path.join(REF_ROOT, "laplace", "utils"))
    lap_curv = _shell("laplace.curvature", os.path.join(REF_ROOT, "laplace", "curvature"))
    gnn = _shell("gnn", os.path.join(REF_ROOT, "gnn"))
    gnn_models = _shell("gnn.models", os.path.join(REF_ROOT, "gnn", "models"))
    lap.utils, lap.curvature, gnn.models = lap_utils, lap_curv, gnn_models

    # alternative backends that are never instantiated on this path
    for mod, names in {
        "laplace.curvature.asdfghjkl": ["AsdfghjklHessian"],
        "laplace.curvature.asdl": ["AsdlGGN"],
        "laplace.curvature.backpack": ["BackPackGGN"],
    }.items():
        m = _shell(mod)
        for n in names:
            setattr(m, n, _placeholder_class(n))

    # ---- the real reference files ---------------------------------------------------
    imp = importlib.import_module
    cl_base = imp("curvlinops._base")
    cl_kfac_utils = imp("curvlinops.kfac_utils")
    cl_kfac = imp("curvlinops.kfac")
    cl.KFACLinearOperator = cl_kfac.KFACLinearOperator
    cl.FisherType = cl_kfac.FisherType
    cl.KFACType = cl_kfac.KFACType
    for n in (
        "EFLinearOperator",
        "FisherMCLinearOperator",
        "GGNLinearOperator",
        "HessianLinearOperator",
    ):  # need backpack.hessianfree -> only reachable through hessian_structure="full"
        setattr(cl, n, _placeholder_class(n))

    enums = imp("laplace.utils.enums")
    for n in (
        "HessianStructure Likelihood LinkApprox PredType PriorStructure "
        "SubsetOfWeights TuningMethod"
    ).split():
        setattr(lap_utils, n, getattr(enums, n))
    utils = imp("laplace.utils.utils")
    for n in utils.__all__:
        if hasattr(utils, n):
            setattr(lap_utils, n, getattr(utils, n))
    lap_utils._is_valid_scalar = utils._is_valid_scalar
    lap_utils.SoDSampler = utils.SoDSampler
    matrix = imp("laplace.utils.matrix")
    lap_utils.Kron, lap_utils.KronDecomposed = matrix.Kron, matrix.KronDecomposed
    metrics = imp("laplace.utils.metrics")
    lap_utils.RunningNLLMetric = metrics.RunningNLLMetric

    curvature = imp("laplace.curvature.curvature")
    lap_curv.CurvatureInterface = curvature.CurvatureInterface
    lap_curv.GGNInterface = curvature.GGNInterface
    lap_curv.EFInterface = curvature.EFInterface
    curv_cl = imp("laplace.curvature.curvlinops")
    baselaplace = imp("laplace.baselaplace")

    g_utils = imp("gnn.models.utils")
    g_layers = imp("gnn.models.layers")
    g_base = imp("gnn.models.base_gnn")
    g_models = imp("gnn.models.models")

    ns = types.SimpleNamespace(
        curvlinops_base=cl_base,
        kfac_utils=cl_kfac_utils,
        kfac=cl_kfac,
        enums=enums,
        utils=utils,
        matrix=matrix,
        curvature=curvature,
        curvature_curvlinops=curv_cl,
        baselaplace=baselaplace,
        gnn_utils=g_utils,
        gnn_layers=g_layers,
        gnn_base=g_base,
        gnn_models=g_models,
    )
    _LOADED = ns
    return ns


# ---- the two integer helpers of gnn/utils.py, which imports torch_geometric / sklearn /
# GPUtil at module level and therefore cannot be imported as a file.  They are pinned by
# *behaviour* in make_golden.py: dense adjacency built with plain torch indexing the way
# gnn/utils.py:325-330 + gnn/marglik_training.py:405 do (scipy COO -> toarray sums
# duplicates, then clamp to 1), and adj_to_edge_index re-stated from gnn/utils.py:333-336.
