"""MI355X-native curvature accumulation for ``Laplace(...).fit()`` on GCN / GraphSAGE models.

Only the hot path of anitasyang/Laplace-GNN is implemented: graph ingest,
the cached sparse forward, KFAC / diagonal / last-layer GGN accumulation as hand-written gfx950
kernels behind a C ABI (``include/laplace_gnn_hip.h``), a ``laplace.curvature``-compatible
backend class and the thin ``Laplace`` front that drives it.
"""
from . import _lib  # noqa: F401
from .engine import GraphEngine  # noqa: F401

__all__ = ["GraphEngine"]
