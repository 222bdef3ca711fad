"""Seeded weight generator shared by the engine, the oracle fixtures and the benchmark.

Mirrors HF `_init_weights` as used by the reference models (`self.post_init()` at
modeling_pretrain.py:117 / modeling_finetune.py:105): Linear/Embedding ~ N(0, initializer_range),
embedding pad row = 0, RMSNorm weights = 1, LayerScale lambdas = init value
(utils_graphgpt.py:95-104), gated stacking weight = kaiming-uniform(a=sqrt(5)) i.e.
U(-1/sqrt(d), 1/sqrt(d)) (modeling_common.py:116-125).  The random stream is NumPy's so the same
seed gives the same tensors in the fixture generator (CPU container) and on the GPU box.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .spec import ModelSpec


def make_state_dict(spec: ModelSpec, seed: int = 0, std: float = 0.02, head_std: float | None = None
                    ) -> "OrderedDict[str, np.ndarray]":
    """Return name -> float32 ndarray for every parameter of `spec` (reference key names)."""
    rng = np.random.RandomState(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in spec.param_table().items():
        if name.endswith("layernorm.weight") or name == "model.norm.weight":      # (incl. embed_layernorm.weight)
            w = np.ones(shape, np.float32)
        elif name.endswith("lambda_1") or name.endswith("lambda_2"):
            w = np.full(shape, spec.layer_scale_init, np.float32)
        elif name == "stacked_feat_agg.weight":
            bound = 1.0 / np.sqrt(shape[1])
            w = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        elif name == "score.bias" or (name.startswith("score.mlp_modules.") and name.endswith(".bias")):
            w = np.zeros(shape, np.float32)
        else:
            s = std
            if head_std is not None and (name in ("lm_head.weight", "n_token_proj.weight", "score.weight")
                                         or name.startswith("score.mlp_modules.")):
                s = head_std
            w = (rng.standard_normal(size=shape) * s).astype(np.float32)
            if name == "model.embed_tokens.weight":
                w[spec.pad_token_id] = 0.0
        out[name] = w
    return out
