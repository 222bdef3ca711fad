"""Architecture description of one Graph Eulerian Transformer and its parameter table.

The hot path is fully determined by a handful of integers; `ModelSpec` carries
them between the Python surface (`GraphGPTConfig`), the C-ABI (`gget_config_t` in include/gget.h)
and the test oracle.  Parameter names and shapes are the reference's state-dict keys
(reference: src/models/graphgpt/modeling_pretrain.py:57-117, modeling_finetune.py:64-105,
modeling_common.py:105-169, utils_graphgpt.py:83-104) so checkpoints interchange by name.
"""
from __future__ import annotations

import dataclasses
from collections import OrderedDict
from typing import Dict, List, Tuple

KIND_PRETRAIN = 0   # GraphGPTPretrainBase: SMTP/NTP head (n_token_proj + lm_head + CE)
KIND_TASK = 1       # GraphGPTTaskModel: `score` head on the pooled ("last") row


@dataclasses.dataclass
class ModelSpec:
    kind: int = KIND_PRETRAIN
    vocab_size: int = 756
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_layers: int = 12
    num_heads: int = 12
    head_dim: int = 64
    stacked_feat: int = 13          # F
    next_n_token: int = 13          # pre-train: F ; fine-tune: unused
    gated_agg: bool = False         # stacked_feat_agg_method == "gated"
    causal: bool = False            # causal_attention
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position: int = 1024
    layer_scale_init: float = 0.0   # >0 => lambda_1/lambda_2 per layer
    num_labels: int = 2             # fine-tune head width
    score_bias: bool = False        # problem_type == "regression"
    pad_token_id: int = 0
    path_pdrop: float = 0.0         # stochastic depth (DropPath) rate of the last layer; 0 = feature not allocated
    mlp_pdrop: float = 0.0          # MLP dropouts (utils_graphgpt.py:69-80); 0 = feature not allocated
    embed_pdrop: float = 0.0        # dropout on the gathered token embeddings (modeling_helpers.py:96-98)
    head_mlp: Tuple[int, ...] = ()  # fine-tune: hidden widths of the `MLP` score head (config.mlp, src/utils/modules_utils.py:8-34); () = Linear
    head_pdrop: float = 0.0         # dropout inside that head (config.dropout)
    rope_range: float = 0.0         # > 0: position ids passed to a forward are rescaled to [0, rope_range) per row (utils_graphgpt.py:574-581)
    embed_dim: int = 0              # > 0: raw-embedding inputs [B,S,embed_dim] (embed_layernorm, emb_mask_token (pre-train), embed_proj)

    def __post_init__(self):
        assert self.hidden_size == self.num_heads * self.head_dim, "no GQA / odd head dims on this path"
        assert self.head_dim == 64, "reference fixes head_dim = 64 (src/utils/modules_utils.py:37-42)"

    # ------------------------------------------------------------------ parameter table
    def param_table(self) -> "OrderedDict[str, Tuple[int, ...]]":
        d, ff, V, F = self.hidden_size, self.intermediate_size, self.vocab_size, self.stacked_feat
        t: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        t["model.embed_tokens.weight"] = (V, d)
        if self.gated_agg:
            t["stacked_feat_agg.weight"] = (F, d)
        if self.embed_dim > 0:    # modeling_pretrain.py:69-84 / modeling_finetune.py:76-85
            t["embed_layernorm.weight"] = (self.embed_dim,)
            if self.kind == KIND_PRETRAIN:
                t["emb_mask_token"] = (1, 1, self.embed_dim)
            t["embed_proj.weight"] = (d, self.embed_dim)
        for i in range(self.num_layers):
            p = f"model.layers.{i}."
            t[p + "input_layernorm.weight"] = (d,)
            t[p + "self_attn.q_proj.weight"] = (d, d)
            t[p + "self_attn.k_proj.weight"] = (d, d)
            t[p + "self_attn.v_proj.weight"] = (d, d)
            t[p + "self_attn.o_proj.weight"] = (d, d)
            if self.layer_scale_init > 0:
                t[p + "lambda_1"] = (d,)
            t[p + "post_attention_layernorm.weight"] = (d,)
            t[p + "mlp.gate_proj.weight"] = (ff, d)
            t[p + "mlp.up_proj.weight"] = (ff, d)
            t[p + "mlp.down_proj.weight"] = (d, ff)
            if self.layer_scale_init > 0:
                t[p + "lambda_2"] = (d,)
        t["model.norm.weight"] = (d,)
        if self.kind == KIND_PRETRAIN:
            if self.next_n_token > 1:
                t["n_token_proj.weight"] = (self.next_n_token * d, d)
            t["lm_head.weight"] = (V, d)
        elif len(self.head_mlp) > 0:
            dims = [d] + [int(x) for x in self.head_mlp] + [self.num_labels]
            for i in range(len(dims) - 1):
                t[f"score.mlp_modules.{i}.weight"] = (dims[i + 1], dims[i])
                if self.score_bias:
                    t[f"score.mlp_modules.{i}.bias"] = (dims[i + 1],)
        else:
            t["score.weight"] = (self.num_labels, d)
            if self.score_bias:
                t["score.bias"] = (self.num_labels,)
        return t

    def num_params(self) -> int:
        n = 0
        for shp in self.param_table().values():
            k = 1
            for s in shp:
                k *= s
            n += k
        return n

    # parameters that are exempt from weight decay in *no* reference path: torch AdamW / DS Adam apply
    # weight decay to every tensor (opt_utils.py:18-24 passes model.parameters() as one group).
    def as_c_ints(self) -> List[int]:
        """Field order of `gget_config_t` (include/gget.h)."""
        return [self.kind, self.vocab_size, self.hidden_size, self.intermediate_size, self.num_layers,
                self.num_heads, self.stacked_feat, self.next_n_token, int(self.gated_agg), int(self.causal),
                self.max_position, self.num_labels, int(self.score_bias), self.pad_token_id]


# Model-size table of the reference launch scripts (examples/graph_lvl/pcqm4m_v2_pretrain.sh:159-202)
MODEL_SIZES: Dict[str, Dict[str, int]] = {
    "tiny": dict(hidden_size=128, num_layers=2),
    "mini": dict(hidden_size=256, num_layers=4),
    "small": dict(hidden_size=512, num_layers=4),
    "medium": dict(hidden_size=512, num_layers=8),
    "base": dict(hidden_size=768, num_layers=12),
    "base24": dict(hidden_size=768, num_layers=24),
    "large": dict(hidden_size=1024, num_layers=24),
}


def spec_from_size(name: str, **kw) -> ModelSpec:
    sz = MODEL_SIZES[name]
    d = sz["hidden_size"]
    return ModelSpec(hidden_size=d, intermediate_size=4 * d, num_layers=sz["num_layers"],
                     num_heads=d // 64, head_dim=64, **kw)
