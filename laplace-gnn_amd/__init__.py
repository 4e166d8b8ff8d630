"""MI355X-native curvature accumulation for ``Laplace(...).fit()`` on GCN / GraphSAGE models.

Only the hot path of anitasyang/Laplace-GNN is implemented: graph ingest,
the cached sparse forward, KFAC / diagonal / last-layer GGN accumulation as hand-written gfx950
kernels behind a C ABI (``include/laplace_gnn_hip.h``), a ``laplace.curvature``-compatible
backend class and the thin ``Laplace`` front that drives it.
"""
from . import _lib  # noqa: F401
from .curvature import HipCurvatureInterface, HipEF, HipGGN  # noqa: F401
from .data import TensorBatchLoader, batches_of_rank, units_of_rank  # noqa: F401
from .engine import GraphEngine  # noqa: F401
from .laplace import (BaseLaplace, DiagLaplace, FullLaplace, FullLLLaplace, KronLaplace, Laplace,  # noqa: F401
                      ParametricLaplace, all_reduce_flat_)
from .matrix import Kron, KronDecomposed, symeig  # noqa: F401
from .models import GCN, STEGCN, GraphSAGE  # noqa: F401

__all__ = ["GraphEngine", "HipGGN", "HipEF", "HipCurvatureInterface", "Laplace", "BaseLaplace", "ParametricLaplace",
           "KronLaplace", "DiagLaplace", "FullLaplace", "FullLLLaplace", "Kron", "KronDecomposed", "symeig", "GCN", "STEGCN", "GraphSAGE",
           "TensorBatchLoader", "batches_of_rank", "units_of_rank", "all_reduce_flat_"]
