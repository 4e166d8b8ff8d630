"""Helpers shared by the CPU and GPU tests: the optional ``res`` / ``norm`` state of a golden fixture
(gnn/models/base_gnn.py:86-113) as keyword arguments of the oracle's ``GnnModel`` and of ``GraphEngine.bind``."""
import numpy as np


def model_extras(g):
    """GnnModel keyword arguments (numpy arrays); {} for fixtures of plain models."""
    if "norm" not in g.files:
        return {}
    L = int(g["num_layers"])
    kw = {}
    if bool(g["res"]):
        kw["res_weights"] = [g[f"Wr{l}"] for l in range(L - 1)]
        kw["res_biases"] = [g[f"br{l}"] for l in range(L - 1)]
    norm = str(g["norm"])
    if norm in ("layer", "batch"):
        kw.update(norm=norm, norm_eps=float(g["norm_eps"]), norm_weight=[g[f"norm_w{l}"] for l in range(L - 1)],
                  norm_bias=[g[f"norm_b{l}"] for l in range(L - 1)])
        if norm == "batch":
            kw.update(norm_mean=[g[f"norm_rm{l}"] for l in range(L - 1)], norm_var=[g[f"norm_rv{l}"] for l in range(L - 1)])
    return kw


def has_extras(g):
    return "norm" in g.files


def constructor_extras(g):
    """``norm=`` / ``res=`` constructor arguments of the model classes."""
    if "norm" not in g.files:
        return {}
    norm = str(g["norm"])
    return dict(norm=None if norm == "None" else norm, res=bool(g["res"]))


def conv_offset(g, layer):
    """Offset of ``convs.{layer}.lin.weight`` in the flat parameter vector (convs.* precede res.*)."""
    return int(sum(g[f"W{l}"].size + g[f"b{l}"].size for l in range(layer)))
