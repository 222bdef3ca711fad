"""Python operator surface of the reference for the hot path, backed by the HIP engine.

Mirrors (same names, argument meaning, return fields, error behaviour):
  GraphGPTConfig                 reference src/models/graphgpt/configuration_graphgpt.py:6-207
  DoubleHeadsModelOutput         reference src/models/graphgpt/modeling_common.py:55-99
  GraphGPTPretrainBase.forward   reference src/models/graphgpt/modeling_pretrain.py:152-266
  GraphGPTTaskModel.forward      reference src/models/graphgpt/modeling_finetune.py:236-326
Parameters are `nn.Parameter`s whose storage IS the engine's fp32 master arena, under the reference's
state-dict keys, so `state_dict()/load_state_dict()/named_parameters()` interchange with reference
checkpoints.  The returned loss is autograd-connected: `loss.backward()` runs the HIP backward and
(optionally) fills `.grad`, so the reference's DDP-style step (training_utils.py:53-86) works; the fast
path is `graph-gpt_amd.training.GgetEngine` (DeepSpeed-engine protocol, fused AdamW, overlapped RCCL).
"""
from __future__ import annotations

import dataclasses
import json
import os
from collections import OrderedDict
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .engine import Engine
from .spec import KIND_PRETRAIN, KIND_TASK, ModelSpec
from .weights import make_state_dict


class GraphGPTConfig:
    """Field names AND defaults follow the reference `GraphGPTConfig(LlamaConfig)` (configuration_graphgpt.py:25-200; note
    `hidden_act="silu"`, `use_cache=True`, `causal_attention=True` there - the Hydra `GraphGPTModelConfig` is what defaults to
    gelu).  Only the fields that reach the hot path are interpreted, every other keyword is stored verbatim (so reference config
    dicts round-trip) - EXCEPT Llama switches that would change the arithmetic of the hot path: those raise in `to_spec`
    instead of being swallowed (attention_bias, mlp_bias, rope_scaling, tie_word_embeddings, GQA, pretraining_tp > 1)."""

    model_type = "graphgpt"

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, hidden_act="silu", max_position_embeddings=2048, initializer_range=0.02,
                 rms_norm_eps=1e-6, use_cache=True, pad_token_id=0, tie_word_embeddings=False, pooling_method="last",
                 causal_attention=True, rope_range=0, embed_pdrop=0.0, path_pdrop=0.0, mlp_pdrop=0.0,
                 layer_scale_init_value=0.0, stacked_feat=1, stack_method=None, stacked_feat_agg_method="sum",
                 embed_dim=0, next_n_token=1, use_generative=True, use_discriminative=False, focal_gamma=0.0,
                 smtp_inside=False, cls_token_id=None, mlp=None, dropout=0.0, loss_type=None, num_neg=None, num_labels=2,
                 problem_type=None, attention_dropout=0.0, rope_theta=10000.0, head_dim=None, num_key_value_heads=None,
                 attention_bias=False, mlp_bias=False, pretraining_tp=1, **kwargs):
        # the reference passes its own `rope_scaling=None` next to **kwargs (configuration_graphgpt.py:118,185-199): a
        # caller-supplied VALUE is a TypeError there, too.  A null / False entry is what to_dict() / save_pretrained() (and any
        # hf LlamaConfig config.json) write for these two fields: dropped, so saved configs load again.
        if kwargs.pop("rope_scaling", None):
            raise TypeError("GraphGPTConfig() got multiple values for keyword argument 'rope_scaling'")
        if kwargs.pop("rope_3d", False):
            raise NotImplementedError("gget engine: rope_3d (out of the hot-path scope, see DESIGN.md)")
        assert pooling_method in {"last", "sum", "mean"}            # configuration_graphgpt.py:137
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.hidden_act, self.max_position_embeddings = hidden_act, max_position_embeddings
        self.initializer_range, self.rms_norm_eps, self.use_cache = initializer_range, rms_norm_eps, use_cache
        self.pad_token_id, self.tie_word_embeddings, self.pooling_method = pad_token_id, tie_word_embeddings, pooling_method
        self.causal_attention, self.rope_range = causal_attention, rope_range
        self.embed_pdrop, self.path_pdrop, self.mlp_pdrop = embed_pdrop, path_pdrop, mlp_pdrop
        self.layer_scale_init_value = layer_scale_init_value
        self.stacked_feat, self.stack_method, self.stacked_feat_agg_method = stacked_feat, stack_method, stacked_feat_agg_method
        self.embed_dim, self.next_n_token = embed_dim, next_n_token
        self.use_generative, self.use_discriminative, self.focal_gamma = use_generative, use_discriminative, focal_gamma
        self.smtp_inside, self.mlp, self.dropout, self.loss_type = smtp_inside, list(mlp or []), dropout, loss_type
        self.cls_token_id = cls_token_id
        self.num_labels, self.problem_type, self.attention_dropout = num_labels, problem_type, attention_dropout
        self.num_neg = num_neg
        self.rope_theta, self.rope_scaling, self.rope_3d = rope_theta, None, False
        # hf LlamaConfig: head_dim defaults to hidden_size // num_attention_heads, num_key_value_heads to num_attention_heads
        self.head_dim = head_dim if head_dim is not None else hidden_size // max(int(num_attention_heads), 1)
        self.num_key_value_heads = num_key_value_heads if num_key_value_heads is not None else num_attention_heads
        self.attention_bias, self.mlp_bias, self.pretraining_tp = attention_bias, mlp_bias, pretraining_tp
        self.num_params = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    # ---- what the engine supports; anything else fails loudly instead of silently diverging from the reference
    def to_spec(self, kind: int) -> ModelSpec:
        def need(cond, msg):
            if not cond:
                raise NotImplementedError(f"gget engine: {msg} (out of the hot-path scope, see DESIGN.md)")
        need(self.hidden_act == "gelu", f"hidden_act={self.hidden_act!r}; the reference configs use exact-erf 'gelu'")
        need(self.head_dim == 64 and self.hidden_size == 64 * self.num_attention_heads, "head_dim must be 64")
        need(self.num_key_value_heads == self.num_attention_heads, "GQA (num_key_value_heads != num_attention_heads) is not used by the reference configs")
        need(not self.attention_bias, "attention_bias=True (biased q/k/v/o projections)")
        need(not self.mlp_bias, "mlp_bias=True (biased gate/up/down projections)")
        need(not self.tie_word_embeddings, "tie_word_embeddings=True (lm_head sharing embed_tokens)")
        need(not getattr(self, "rope_scaling", None), f"rope_scaling={getattr(self, 'rope_scaling', None)!r}")
        need(not getattr(self, "rope_3d", False), "rope_3d")
        need(int(self.pretraining_tp or 1) == 1, "pretraining_tp > 1 (sliced projections)")
        need(self.embed_dim % 64 == 0 and 0 <= self.embed_dim <= 2048, f"embed_dim={self.embed_dim}: raw-embedding width must be a multiple of 64 (<= 2048)")
        need(self.stack_method in ("short", "long", None), f"stack_method={self.stack_method!r}")
        need(not self.use_discriminative and self.use_generative, "contrastive (pretrain-cl) head")
        need(len(self.mlp) <= 4, "an MLP score head with more than 4 hidden layers")
        need(self.pooling_method == "last", "pooling other than 'last'")
        return ModelSpec(kind=kind, vocab_size=self.vocab_size, hidden_size=self.hidden_size,
                         intermediate_size=self.intermediate_size, num_layers=self.num_hidden_layers,
                         num_heads=self.num_attention_heads, head_dim=64, stacked_feat=self.stacked_feat,
                         next_n_token=self.next_n_token if kind == KIND_PRETRAIN else 1,
                         gated_agg=self.stacked_feat_agg_method == "gated", causal=bool(self.causal_attention),
                         rms_eps=self.rms_norm_eps, rope_theta=self.rope_theta, max_position=self.max_position_embeddings,
                         layer_scale_init=float(self.layer_scale_init_value), num_labels=self.num_labels,
                         score_bias=self.problem_type == "regression", pad_token_id=self.pad_token_id,
                         path_pdrop=float(self.path_pdrop), mlp_pdrop=float(self.mlp_pdrop),
                         embed_pdrop=float(self.embed_pdrop),
                         head_mlp=tuple(int(x) for x in self.mlp) if kind == KIND_TASK else (),
                         head_pdrop=float(self.dropout) if kind == KIND_TASK else 0.0,
                         rope_range=float(self.rope_range or 0), embed_dim=int(self.embed_dim or 0))

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    def save_pretrained(self, save_directory: str):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as fh:
            json.dump(dict(self.to_dict(), model_type=self.model_type), fh, indent=2, default=str)

    @classmethod
    def from_pretrained(cls, directory: str):
        with open(os.path.join(directory, "config.json")) as fh:
            d = json.load(fh)
        d.pop("model_type", None)
        return cls(**d)


@dataclasses.dataclass
class DoubleHeadsModelOutput:
    pretrain_loss: Optional[torch.Tensor] = None
    task_loss: Optional[torch.Tensor] = None
    pretrain_logits: Optional[torch.Tensor] = None
    task_logits: Optional[torch.Tensor] = None
    head1_loss: Optional[torch.Tensor] = None
    head2_loss: Optional[torch.Tensor] = None
    head1_logits: Optional[torch.Tensor] = None
    head2_logits: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Any = None
    task_hidden_states: Optional[torch.Tensor] = None
    attentions: Any = None


class _LazyLogits:
    """head1_logits costs a device->host count read; fetch only when somebody looks at it."""

    def __init__(self, model):
        self._m, self._v = model, None

    def get(self):
        if self._v is None:
            self._v = self._m._engine.head_logits()
        return self._v


class _PretrainOutput(DoubleHeadsModelOutput):
    def __init__(self, loss, lazy):
        super().__init__(head1_loss=loss)
        self._lazy = lazy

    def __getattribute__(self, name):
        if name == "head1_logits":
            return object.__getattribute__(self, "_lazy").get()
        return object.__getattribute__(self, name)


class _LossFn(torch.autograd.Function):
    """Bridges the HIP backward into autograd so `loss.backward()` (reference training_utils.py:66) works."""

    @staticmethod
    def forward(ctx, anchor, model, loss_val):
        ctx.model = model
        return loss_val.detach()      # (an alias of the engine's per-call result tensor: no copy kernel in the step)

    @staticmethod
    def backward(ctx, g):
        ctx.model._autograd_backward(g)
        return None, None, None


class _Namespace(nn.Module):
    pass


class _GgetModel(nn.Module):
    kind = KIND_PRETRAIN

    def __init__(self, config: GraphGPTConfig, seed: int = 0):
        super().__init__()
        self.config = config
        self.spec = config.to_spec(self.kind)
        self.num_labels = config.num_labels
        self._engine: Optional[Engine] = None
        self._anchor = None
        self.dropout_seed, self._drop_step = seed, 0
        self.materialize_grads = True   # fill nn.Parameter.grad (fp32) after backward, like autograd would
        self._token_layout = "auto"
        self._dirty = False             # master weights changed behind the engine's back (external optimizer)
        state = make_state_dict(self.spec, seed=seed, std=config.initializer_range)
        # module tree with the reference's attribute paths: model.embed_tokens / model.layers[i].* / model.norm ...
        self._flat: "OrderedDict[str, nn.Parameter]" = OrderedDict()
        for name, arr in state.items():
            p = nn.Parameter(torch.from_numpy(arr.copy()))
            self._flat[name] = p
            mod = self
            parts = name.split(".")
            for i, part in enumerate(parts[:-1]):
                if part.isdigit():
                    lst = mod
                    while len(lst) <= int(part):
                        lst.append(_Namespace())
                    mod = lst[int(part)]
                else:
                    if not hasattr(mod, part):
                        nxt = nn.ModuleList() if (i + 1 < len(parts) - 1 and parts[i + 1].isdigit()) else _Namespace()
                        setattr(mod, part, nxt)
                    mod = getattr(mod, part)
            mod.register_parameter(parts[-1], p)
        self.config.num_params = sum(p.numel() for p in self._flat.values())

    # ---- nn.Module protocol bits the reference pipeline touches
    @property
    def device(self):
        return next(iter(self._flat.values())).device

    @property
    def dtype(self):
        return torch.bfloat16

    # token layout of this model's forwards - a per-model switch beside the process-wide GGET_VARLEN:
    #   "auto"   (default) the environment decides; without it: var-len whenever the real-token count is known or countable
    #   "padded" every row of the [B,S] grid, as the reference computes it (hidden-state accessors, debugging)
    #   "nosync" var-len only when the count costs no device->host read (host mask or num_tokens=); device masks run padded
    #   "sync"   count on the device even when the caller passed nothing
    _LAYOUTS = {"auto": None, "varlen": "", "padded": "0", "nosync": "nosync", "sync": "sync"}

    @property
    def token_layout(self) -> str:
        return self._token_layout

    @token_layout.setter
    def token_layout(self, value: str):
        if value not in self._LAYOUTS:
            raise ValueError(f"token_layout: one of {sorted(self._LAYOUTS)}, got {value!r}")
        self._token_layout = value
        if self._engine is not None:
            self._engine.varlen_mode = self._LAYOUTS[value]

    def gradient_checkpointing_enable(self, *a, **k):
        return None  # activations for 288 GB HBM are kept; recompute is never needed on this path

    def _apply(self, fn, *a, **k):
        # `.to(device)` / `.cuda()`: parameters move into the engine arena on first GPU use instead.  Once the engine exists
        # the parameters ARE views of its fp32 master arena: a move that would re-allocate them (`.cpu()`, `.half()`,
        # `.to(torch.bfloat16)`) would silently detach `state_dict()` / optimizers from the weights being trained, so it
        # raises; moves that change nothing (`.cuda()`, `.float()`, `.to(same device)`) are no-ops.
        if self._engine is not None:
            probe = fn(torch.zeros(1, device=self._engine.device))
            if probe.device == self._engine.device and probe.dtype == torch.float32:
                return self
            raise RuntimeError(f"GraphGPT engine: parameters live in the engine's fp32 master arena on {self._engine.device}; "
                               f"moving them to {probe.device}/{probe.dtype} is not supported (the bf16 compute copy is kept by "
                               "the engine itself; use state_dict() to export weights)")
        probe = fn(torch.zeros(1))
        if probe.device.type == "cuda":
            with torch.cuda.device(probe.device):
                self._ensure_engine(1, 8)
            return self
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True):
        res = super().load_state_dict(state_dict, strict=strict)
        self._dirty = True
        return res

    def mark_dirty(self):
        self._dirty = True

    # ---- engine management
    def _ensure_engine(self, B: int, S: int):
        need_tok, need_b = B * S, B
        e = self._engine
        if e is not None and need_tok <= e.cfg.max_tokens and need_b <= e.cfg.max_batch:
            return e
        if not torch.cuda.is_available():
            raise L.GgetError("GraphGPT engine: no GPU visible and there is no CPU fallback "
                              "(the CPU statement of this path lives in oracle/ and is test-only)")
        cap_tok = max(need_tok, e.cfg.max_tokens if e else 0)
        cap_b = max(need_b, e.cfg.max_batch if e else 0)
        new = Engine(self.spec, cap_tok, cap_b)
        for name, p in self._flat.items():
            new.view(name, "master").view(p.shape).copy_(p.data.to(new.device, torch.float32))
        if e is not None:
            new.adam_m.copy_(e.adam_m)
            new.adam_v.copy_(e.adam_v)
            new.step_count = e.step_count
            new.comm_adopt(e)       # an RCCL communicator bound through the C ABI moves to the new handle
        for name, p in self._flat.items():
            p.data = new.view(name, "master").view(p.shape)      # (emb_mask_token is [1,1,e] in the reference, flat in the engine)
        new.sync_params()
        new.set_stack_method(getattr(self.config, "stack_method", None) == "long")
        new.set_rope_range(float(getattr(self.config, "rope_range", 0) or 0))
        new.varlen_mode = self._LAYOUTS[self._token_layout]
        self._engine = new
        self._anchor = torch.zeros(1, device=new.device, requires_grad=True)
        self._dirty = False
        return new

    def _pre_forward(self, B, S):
        e = self._ensure_engine(B, S)
        if self._dirty:
            e.sync_params()
            self._dirty = False
        # attention dropout exactly when the reference applies it: module in training mode and attention_dropout > 0
        # (hf eager_attention_forward :210).  A fresh mask every step, different on every rank.
        p = float(self.config.attention_dropout) if self.training else 0.0
        pp = float(self.config.path_pdrop) if self.training else 0.0
        # embed_dropout (modeling_helpers.py:96-98) and the MLP dropouts (utils_graphgpt.py:69-80): nn.Dropout modules, same rule
        pe = float(self.config.embed_pdrop) if self.training else 0.0
        pm = float(self.config.mlp_pdrop) if self.training else 0.0
        ph = float(self.config.dropout) if (self.training and self.kind == KIND_TASK and len(self.config.mlp) > 0) else 0.0
        if p > 0 or pp > 0 or pe > 0 or pm > 0 or ph > 0:
            self._drop_step += 1
            rank = int(os.environ.get("RANK", "0"))
            seed = (self.dropout_seed * 0x9E3779B1 + self._drop_step * 0x85EBCA6B + rank * 0xC2B2AE35) & 0xFFFFFFFF
            e.set_dropout(p, pp, seed)
            self.last_dropout_seed = seed
        else:
            e.set_dropout(0.0, 0.0, 0)
        e.set_dropout_ex(pe, pm, ph)
        if self.kind == KIND_PRETRAIN:
            e.set_focal_gamma(float(getattr(self.config, "focal_gamma", 0.0) or 0.0))
        return e

    def _autograd_backward(self, g):
        """`loss.backward()` of the reference's DDP-style step (training_utils.py:53-86).  The gradients are produced by the
        HIP backward, not by autograd, so DDP's reducer hooks never see them: when a process group is up and this model is
        not driven by a `GgetEngine` (which exchanges the gradient buckets itself), the flat bf16 gradient array is
        sum-all-reduced here and averaged, which is what DDP would have done (opt_utils.py:13)."""
        e = self._engine
        e.backward()
        world = 1
        if not getattr(self, "_managed_by_engine", False) and torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
            if world > 1:
                torch.distributed.all_reduce(e.grad_bf16, op=torch.distributed.ReduceOp.SUM)
        if self.materialize_grads:
            scale = g.to(torch.float32) / world
            for name, p in self._flat.items():
                p.grad = (e.view(name, "grad").to(torch.float32) * scale).view(p.shape)
        self._dirty = True  # an external optimizer will now touch the master weights

    def _check_positions(self, position_ids, S):
        """The RoPE table is precomputed for max_position_embeddings rows (the reference evaluates the rotary embedding on the
        fly, hf LlamaRotaryEmbedding.forward :111-127, and accepts any position): a larger position raises IndexError instead of
        silently diverging.  HOST tensors (what the reference's loops receive from the DataLoader) are checked here, for free.
        DEVICE tensors are NOT read back per step (that would stall the stream on every fine-tune forward): the engine clamps them
        into the table and raises a sticky device flag, which `check_deferred()` turns into the same IndexError at a time of the
        caller's choosing (TrainingMode.run_training: every log step and at the end; evaluate / ft_evaluate: at the end);
        GGET_CHECK_INPUTS=1 checks every forward immediately."""
        maxp = self.spec.max_position
        if position_ids is None:
            if S > maxp:
                raise IndexError(f"sequence length {S} exceeds max_position_embeddings {maxp}")
            return
        if float(getattr(self.config, "rope_range", 0) or 0) > 0 or not position_ids.numel():
            return      # (rope_range: the positions are rescaled per row, angles are evaluated per token - no table to overrun)
        if position_ids.device.type == "cpu" or os.environ.get("GGET_CHECK_INPUTS"):
            hi, lo = int(position_ids.max()), int(position_ids.min())
            if hi >= maxp or lo < 0:
                raise IndexError(f"position_ids must lie in [0, max_position_embeddings = {maxp}); got [{lo}, {hi}] - build the model "
                                 "with a larger max_position_embeddings")

    def check_deferred(self):
        """Raise what the per-step device-side guards recorded since the last call (one stream sync): position_ids outside the RoPE
        table (IndexError, as _check_positions raises for host tensors); a var-len step whose `num_tokens` disagreed with its
        attention mask, or labels != -100 at padded positions on the var-len layout (ValueError - the step's loss was NaN / differed
        from the padded grid's)."""
        e = self._engine
        if e is None:
            return
        clamped, mismatch = e.deferred_status()
        if clamped:
            raise IndexError(f"position_ids outside [0, max_position_embeddings = {self.spec.max_position}) were passed to a forward since "
                             "the last check (the engine clamped them into the RoPE table: results diverge from the reference) - build "
                             "the model with a larger max_position_embeddings")
        if mismatch:
            raise ValueError("a forward since the last check ran on the var-len token layout with a `num_tokens` that differs from "
                             "sum(attention_mask) (its loss was set to NaN), or met labels != -100 at padded positions; pass the right "
                             "count, no count at all (the engine counts a device mask itself), or GGET_VARLEN=0")

    def _validate_inputs(self, input_ids, attention_mask, labels=None):
        """Debug-mode input validation (GGET_CHECK_INPUTS=1; costs device->host reads): the kernels index with the ids / labels
        they are given, where the reference's nn.Embedding / CrossEntropyLoss raise an index error, and a 2-D mask is reduced
        to ONE key length per row, where the reference accepts any 0/1 pattern."""
        if not os.environ.get("GGET_CHECK_INPUTS"):
            return
        check_batch(input_ids, attention_mask, labels, self.config.vocab_size)

    @staticmethod
    def _token_count(attention_mask, num_tokens):
        """Real (un-padded) tokens of the batch for the var-len token layout (Engine.set_token_count): the caller's `num_tokens`;
        sum(attention_mask) when the mask still lives on the HOST (the loops of graph-gpt_amd.training pass CPU batches: counting
        there costs nothing); "auto" for a DEVICE mask - the call shape of the reference's own step, which moves every tensor first
        (training_utils.py:14-26): the engine then sums the key lengths on the device and reads 4 bytes back (one stream
        synchronisation, right behind the ones `.to(device)` just paid)."""
        if num_tokens is not None:
            return int(num_tokens)
        if attention_mask is not None and attention_mask.dim() == 2:
            return int((attention_mask != 0).sum()) if attention_mask.device.type == "cpu" else "auto"
        return None       # (GGET_VARLEN=sync / nosync / 0 are handled by Engine._set_layout)

    def _want_hidden_states(self, output_hidden_states):
        """hf resolve: the call's flag, else config.output_hidden_states (modeling_helpers.resolve_forward_defaults)."""
        if output_hidden_states is None:
            output_hidden_states = getattr(self.config, "output_hidden_states", False)
        return bool(output_hidden_states)

    def _collect_hidden_states(self, B, S):
        """`outputs.hidden_states` of the reference's backbone with output_hidden_states=True (hf LlamaModel.forward
        modeling_llama.py:401-414): a tuple of L + 1 tensors [B,S,d] - the residual stream entering every decoder layer, then the
        FINAL-NORMED output of the last layer.  bf16 copies of the engine's activations in the [B,S,d] layout whatever token layout
        the forward ran on (gget_hidden_states_grid: after a var-len forward the compact rows go back to their positions, the
        positions behind a sample's tokens read as zero - the reference computes values there that nothing uses)."""
        e = self._engine
        return tuple(e.layer_hidden_states(i, B, S) for i in range(self.spec.num_layers)) + (e.hidden_states(B, S),)

    def _wrap_loss(self, loss):
        if loss is None:
            return None
        if torch.is_grad_enabled():
            return _LossFn.apply(self._anchor, self, loss)
        return loss.detach()


class GraphGPTPretrainBase(_GgetModel):
    kind = KIND_PRETRAIN

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                inputs_raw_embeds=None, labels=None, label_mask=None, sample_wgt=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, cache_position=None, num_tokens=None):
        assert inputs_embeds is None, "inputs_embeds is not supported (reference asserts the same, modeling_helpers.py:95)"
        assert (inputs_raw_embeds is not None) == (int(self.config.embed_dim or 0) > 0), \
            "inputs_raw_embeds are given exactly when the model was built with embed_dim > 0 (modeling_pretrain.py:131-132)"
        if input_ids.dim() == 2:
            input_ids = input_ids[:, :, None]
        if getattr(self.config, "smtp_inside", False):
            # reference modeling_pretrain.py:175-189: the batch carries 4 extra columns (pos_deco), column F+2 is the node
            # index; ids are masked and labels produced on the device (HIP kernel behind gget_op_smtp2d)
            from .smtp import smtp2d_mask
            F = self.spec.stacked_feat
            full = input_ids.to(device=self.device if self._engine is not None else torch.device("cuda"), dtype=torch.int64).contiguous()
            self._smtp_step = getattr(self, "_smtp_step", 0) + 1
            seed = (self.dropout_seed * 0x9E3779B1 + self._smtp_step * 0x85EBCA77 + int(os.environ.get("RANK", "0")) * 0xC2B2AE3D) & 0xFFFFFFFF
            input_ids, labels = smtp2d_mask(full, full[:, :, F + 2], F, smtp_2d_rate=1.0,
                                            power=float(getattr(self.config, "smtp_power", 1.0)), replace_rate=0.0,
                                            vocab=self.config.vocab_size, global_2d_mask=False, seed=seed)
        B, S = input_ids.shape[:2]
        assert input_ids.shape[2] == self.spec.stacked_feat, \
            f"stacked_feat: {self.spec.stacked_feat}\nx.shape: {tuple(input_ids.shape)}"  # modeling_common.py:131-133
        want_hs = self._want_hidden_states(output_hidden_states)
        n_real = self._token_count(attention_mask, num_tokens)     # (hidden states: gget_hidden_states_grid spreads the compact rows back)
        if attention_mask is None:
            attention_mask = torch.ones(B, S, dtype=torch.int64)
        assert attention_mask.dim() in (2, 3), "attention_mask is [B,S] (right padding) or [B,S,S] (packed, block-diagonal)"
        if attention_mask.dim() == 3:
            assert not self.spec.causal, "the reference only builds the 3-D mask for bi-directional attention (modeling_pretrain.py:197-198)"
        self._check_positions(position_ids, S)
        self._validate_inputs(input_ids, attention_mask, labels)
        e = self._pre_forward(B, S)
        if inputs_raw_embeds is not None:
            e.set_raw_embeds(inputs_raw_embeds, first_label_only=bool(getattr(self.config, "smtp_inside", False)))
        loss = e.forward_pretrain(input_ids, attention_mask, labels, sample_wgt, position_ids, num_tokens=n_real)
        out = _PretrainOutput(self._wrap_loss(loss), _LazyLogits(self))
        if want_hs:
            out.hidden_states = self._collect_hidden_states(B, S)       # modeling_pretrain.py:264 (outputs.hidden_states)
        return out


class GraphGPTTaskModel(_GgetModel):
    kind = KIND_TASK

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                inputs_raw_embeds=None, task_labels=None, cls_idx=None, sample_wgt=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, num_tokens=None, **kwargs):
        assert inputs_embeds is None
        if kwargs.get("pretrain_labels") is not None:     # (the reference's ft step passes the keyword, None unless finetune.use_aux)
            raise NotImplementedError("pretrain_labels (use_aux: the double-heads fine-tune model) is outside the hot-path scope")
        assert (inputs_raw_embeds is not None) == (int(self.config.embed_dim or 0) > 0), \
            "inputs_raw_embeds are given exactly when the model was built with embed_dim > 0 (modeling_helpers.py:127-139)"
        if input_ids.dim() == 2:
            input_ids = input_ids[:, :, None]
        B, S = input_ids.shape[:2]
        want_hs = self._want_hidden_states(output_hidden_states)
        n_real = self._token_count(attention_mask, num_tokens)
        if attention_mask is None:
            attention_mask = torch.ones(B, S, dtype=torch.int64)
        cfg = self.config
        problem = cfg.problem_type
        if problem is None and task_labels is not None:  # modeling_finetune.py:175-183
            if self.num_labels == 1:
                problem = "regression"
            elif task_labels.dtype in (torch.long, torch.int):
                problem = "single_label_classification"
            else:
                problem = "multi_label_classification"
            cfg.problem_type = problem
        if problem == "regression":
            code = L.PROBLEM_REGRESSION_L1 if cfg.loss_type == "l1" else L.PROBLEM_REGRESSION_MSE
        elif problem in ("single_label_classification", None):
            if cfg.loss_type == "token_ce_intra":
                raise NotImplementedError("loss_type='token_ce_intra' (intra-instance label embeddings) is outside the hot-path scope")
            # "token_ce" (node-level tasks): `score` and the cross-entropy on every row, task_logits [B,S,num_labels]
            code = L.PROBLEM_TOKEN_CE if cfg.loss_type == "token_ce" else (L.PROBLEM_AUC if cfg.loss_type == "auc" else L.PROBLEM_SINGLE_LABEL)
            if code == L.PROBLEM_TOKEN_CE and len(cfg.mlp) > 0:
                raise NotImplementedError("loss_type='token_ce' with an MLP score head is outside the hot-path scope")
        else:
            code = L.PROBLEM_MULTI_LABEL   # BCE-with-logits on the labelled entries (modeling_finetune.py:227-230)
        self._check_positions(position_ids, S)
        self._validate_inputs(input_ids, attention_mask)
        e = self._pre_forward(B, S)
        if code == L.PROBLEM_AUC and task_labels is not None:
            # one seed per call (the reference draws torch.randperm from the global generator every call)
            self._auc_calls = getattr(self, "_auc_calls", 0) + 1
            self.last_auc_seed = (int(getattr(self, "auc_seed", 0)) + 0x9E3779B1 * self._auc_calls) & 0xFFFFFFFF
            e.set_auc(cfg.num_neg or 1, self.last_auc_seed)
        if inputs_raw_embeds is not None:
            e.set_raw_embeds(inputs_raw_embeds)
        loss, logits, hid = e.forward_task(input_ids, attention_mask, position_ids, task_labels, sample_wgt, code, num_tokens=n_real)
        return DoubleHeadsModelOutput(pretrain_loss=None, task_loss=self._wrap_loss(loss), pretrain_logits=None,
                                      task_logits=logits, task_hidden_states=hid,
                                      hidden_states=self._collect_hidden_states(B, S) if want_hs else None)   # modeling_finetune.py:323


def check_batch(input_ids, attention_mask, labels, vocab_size: int):
    """What `GGET_CHECK_INPUTS=1` verifies before a forward (see _GgetModel._validate_inputs); raises like the reference would."""
    lo, hi = int(input_ids.min()), int(input_ids.max())
    if lo < 0 or hi >= vocab_size:
        raise IndexError(f"input_ids must lie in [0, vocab_size = {vocab_size}); got [{lo}, {hi}]")
    if labels is not None:
        lab = labels[labels != -100]
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= vocab_size):
            raise IndexError(f"labels must be -100 or lie in [0, vocab_size = {vocab_size}); got [{int(lab.min())}, {int(lab.max())}]")
    if attention_mask is not None and attention_mask.dim() == 2:
        m = attention_mask != 0
        n = m.sum(-1, keepdim=True)
        want = torch.arange(m.shape[1], device=m.device)[None, :] < n
        if not bool((m == want).all()):
            raise ValueError("a 2-D attention_mask must be right-padded (1 ... 1 0 ... 0): the engine keeps one key length per row; "
                             "pass a [B,S,S] block-diagonal mask for packed rows")


def elem_drop_keep(seed: int, which: str, layer: int, rows: int, cols: int, p: float):
    """Python twin of the element dropouts (csrc/common.h:elem_drop_mul): the keep multiplier (0 or 1/(1-p)) of every element
    of a [rows, cols] tensor.  which = "embed" (rows = B*S*F cells), "mlp_act" ([T, ff]), "mlp_out" ([T, d]) or "head" (input of
    Linear `layer` of the MLP score head, [B, width]); `seed` is the seed handed to set_dropout; `layer` is ignored for "embed"."""
    import numpy as np
    from .smtp import _rng24
    if p <= 0:
        return np.ones((rows, cols), np.float32)
    stream = {"embed": 48, "raw": 60, "mlp_act": 49, "mlp_out": 50, "head": 51 + layer}[which]     # "raw": raw_embed_dropout, [T, embed_dim]
    s = (seed ^ 0x5BD1E995) if which in ("embed", "raw") else ((seed ^ 0x2545F491) if which == "head" else (seed + 0x7F4A7C15 * (layer + 1)))
    s &= 0xFFFFFFFF
    thresh = int(np.float32(p) * np.float32(16777216.0))
    a = np.arange(rows, dtype=np.uint64)[:, None]
    b = np.arange(cols, dtype=np.uint64)[None, :]
    r = _rng24(s, stream, a, b)
    return np.where(r < thresh, np.float32(0), np.float32(1.0) / (np.float32(1.0) - np.float32(p))).astype(np.float32)


def auc_pairs(labels, num_neg: int, seed: int):
    """Python twin of the negative sampling in csrc/kernels.hip:auc_loss_kernel: returns the index (into the ordered list of
    negative samples) paired with each of the P * num_neg (positive, slot) entries - what `torch.randperm(cnt) % n_neg`
    is in the reference (src/utils/loss_utils.py:39-42)."""
    import numpy as np
    from .smtp import _rng24
    y = np.asarray(labels).reshape(-1)
    P = int((y != 0).sum())
    N = int(y.size - P)
    cnt = P * int(num_neg)
    keys = _rng24(seed, 40, np.arange(cnt), 0)
    order = np.lexsort((np.arange(cnt), keys))      # sort by key, ties by index
    rank = np.empty(cnt, dtype=np.int64)
    rank[order] = np.arange(cnt)
    return rank % max(N, 1)


# flat GraphGPTConfig keyword -> attribute path inside the reference's GraphGPTModelConfig (src/conf/model/model_configs.py:247-287
# and its sub-configs :13-244), exactly the map of configuration_graphgpt.py:222-318.  "<section>:" prefixes abbreviate.
_LEGACY_TOP = ("vocab_size hidden_size intermediate_size num_hidden_layers num_attention_heads num_key_value_heads head_dim "
               "attention_bias mlp_bias hidden_act max_position_embeddings initializer_range rms_norm_eps tie_word_embeddings "
               "rope_theta use_cache pad_token_id bos_token_id eos_token_id cls_token_id causal_attention rope_range "
               "layer_scale_init_value").split()
_LEGACY_NESTED = {
    "dropout_settings": dict(embed_pdrop="embed_dropout", path_pdrop="path_dropout", mlp_pdrop="mlp_dropout",
                             attention_dropout="attention_dropout"),
    "graph_input": dict(stacked_feat="stacked_feat", stack_method="stack_method",
                        stacked_feat_agg_method="stacked_feat_agg_method", embed_dim="embed_dim"),
    "geometric_input": dict(pos_agg_method="pos_agg_method", pos_bins="pos_bins"),
    "pt_head": dict(next_n_token="next_n_token", use_generative="use_generative", use_discriminative="use_discriminative",
                    focal_gamma="focal_gamma", smtp_inside="smtp_inside"),
    "ft_head": dict(pooling_method="pooling_method", mlp="mlp", dropout="dropout", loss_type="loss_type", num_neg="num_neg",
                    num_labels="num_labels", problem_type="problem_type"),
    "pos_pt_head": dict(smtp_power="smtp_power", pt_problem_type="problem_type", smtp_3d_power="smtp_3d_power",
                        smtp_3d_noise_scale="smtp_3d_noise_scale", coord_lvl_mask="coord_lvl_mask", pt_num_bins="num_bins",
                        pt_num_bins_line="num_bins_line", pt_num_bins_cube="num_bins_cube", apply_denoise="apply_denoise",
                        label_smoothing="label_smoothing", pt_pos_agg_method="pos_agg_method", use_pos_proj="use_pos_proj",
                        loss_agg="loss_agg", pt_pos_range="pos_range", pt_smtp_2d_rate="smtp_2d_rate",
                        smtp_2d_replace_rate="smtp_2d_replace_rate", sep_2d3d_inputs="sep_2d3d_inputs",
                        global_2d_mask="global_2d_mask", pt_use_discriminative="use_discriminative"),
    "denoise_head": dict(noise_scale="noise_scale", denoise_wgt="denoise_wgt", denoise_schedule_pow="denoise_schedule_pow",
                         bi_causal="bi_causal", r_2d="r_2d", r_3d="r_3d", r_both="r_both", add_pos_type="add_pos_type",
                         inputs_transform="inputs_transform", num_bins_line="num_bins_line", num_bins_cube="num_bins_cube",
                         dn_pos_range="pos_range", dn_use_pos_proj="use_pos_proj", smtp_3d="smtp_3d", smtp_wgt="smtp_wgt",
                         smtp_3d_scheduler_power="smtp_3d_scheduler_power", smtp_denoise="smtp_denoise",
                         smtp_vocab="smtp_vocab", dn_smtp_2d_rate="smtp_2d_rate",
                         smtp_2d_scheduler_power="smtp_2d_scheduler_power"),
}
_ROPE_SCALING_FIELDS = ("rope_type factor original_max_position_embeddings attention_factor beta_fast beta_slow short_factor "
                        "long_factor low_freq_factor high_freq_factor").split()


def legacy_config_kwargs(model_cfg) -> Dict[str, Any]:
    """The flat keyword dict the reference's converter builds (configuration_graphgpt.py:222-335) from a nested
    `GraphGPTModelConfig` (attributes, an OmegaConf node or plain dicts), `None` values dropped like the reference does.
    A section or field the reference's dataclasses always carry and that is MISSING here is an error, not a default: this
    function used to fall back silently and trained a different network."""
    missing = object()

    def get(obj, name, where):
        val = obj.get(name, missing) if isinstance(obj, dict) or (hasattr(obj, "get") and hasattr(obj, "keys")) else getattr(obj, name, missing)
        if val is missing:
            raise AttributeError(f"convert_to_legacy_config: the model config has no field {where}{name!r} "
                                 "(expected the layout of the reference's GraphGPTModelConfig, src/conf/model/model_configs.py)")
        return val

    kw: Dict[str, Any] = {k: get(model_cfg, k, "") for k in _LEGACY_TOP}
    for section, fields in _LEGACY_NESTED.items():
        sub = get(model_cfg, section, "")
        for flat, name in fields.items():
            kw[flat] = get(sub, name, section + ".")
    kw["use_aux"] = get(get(model_cfg, "ft_head", ""), "task_ratio", "ft_head.") < 1      # configuration_graphgpt.py:275
    mlp = kw.get("mlp")
    if mlp is not None and not isinstance(mlp, list):
        kw["mlp"] = list(mlp)                       # (OmegaConf ListConfig)
    rs = get(model_cfg, "rope_scaling", "")
    if rs:
        kw["rope_scaling"] = {f: get(rs, f, "rope_scaling.") for f in _ROPE_SCALING_FIELDS}
    return {k: v for k, v in kw.items() if v is not None}


def convert_to_legacy_config(model_cfg) -> GraphGPTConfig:
    """Counterpart of reference configuration_graphgpt.py:210-342: nested `GraphGPTModelConfig` -> flat `GraphGPTConfig`, every
    keyword of the reference's map (tests/golden/config_convert.json holds the reference's own output for the PCQM4M-v2
    pre-train and ogbl-ppa fine-tune settings; tests/test_host_logic.py compares field by field).  A flat dict of
    GraphGPTConfig keywords is accepted as well.  Like the reference, a non-empty `rope_scaling` reaches GraphGPTConfig as a
    duplicate keyword and raises TypeError."""
    if isinstance(model_cfg, dict) and "hidden_size" in model_cfg and "graph_input" not in model_cfg:
        return GraphGPTConfig(**model_cfg)
    return GraphGPTConfig(**legacy_config_kwargs(model_cfg))
