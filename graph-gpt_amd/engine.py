"""Python host side of the engine: owns the HBM arenas (as torch tensors: PyTorch is the allocator
and stream provider, nothing more), binds them to a `gget_handle_t`, and exposes forward / backward /
optimizer step on raw device pointers.  Mirrors what the reference gets from
`deepspeed.initialize(...)` / DDP+AdamW (SURVEY.md section 8b "Engine protocol")."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .spec import KIND_PRETRAIN, KIND_TASK, ModelSpec


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rope_tables(max_position: int, head_dim: int, theta: float):
    """fp32 cos/sin [max_position, head_dim/2], computed exactly like hf LlamaRotaryEmbedding.forward
    :111-127 does on the host (fp32 inv_freq, fp32 outer product, fp32 cos/sin)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(max_position, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


class Engine:
    def __init__(self, spec: ModelSpec, max_tokens: int, max_batch: int, device: Optional[torch.device] = None,
                 with_optimizer: bool = True):
        if not torch.cuda.is_available():
            raise L.GgetError("the gget engine needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = L.load()
        self.spec = spec
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        cfg = L.GgetConfig()
        (cfg.kind, cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads,
         cfg.stacked_feat, cfg.next_n_token, cfg.gated_agg, cfg.causal, cfg.max_position, cfg.num_labels,
         cfg.score_bias, cfg.pad_token_id) = spec.as_c_ints()
        cfg.rms_eps, cfg.rope_theta, cfg.layer_scale_init = spec.rms_eps, spec.rope_theta, spec.layer_scale_init
        cfg.max_tokens, cfg.max_batch = int(max_tokens), int(max_batch)
        cfg.path_pdrop = float(getattr(spec, "path_pdrop", 0.0))
        cfg.mlp_pdrop = float(getattr(spec, "mlp_pdrop", 0.0))
        hm = [int(x) for x in getattr(spec, "head_mlp", ())]
        assert len(hm) <= 4, "the MLP score head holds at most 4 hidden layers"
        cfg.head_mlp_layers = len(hm)
        for i, x in enumerate(hm):
            cfg.head_mlp[i] = x
        cfg.embed_dim = int(getattr(spec, "embed_dim", 0) or 0)
        self.cfg = cfg
        sz = L.GgetSizes()
        L.check(self.lib.gget_query_sizes(C.byref(cfg), C.byref(sz)))
        self.n_params = int(sz.n_params)
        dev = self.device
        with torch.cuda.device(dev):
            self.param_bf16 = torch.zeros(self.n_params, dtype=torch.bfloat16, device=dev)
            self.grad_bf16 = torch.zeros(self.n_params, dtype=torch.bfloat16, device=dev)
            self.master = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
            if with_optimizer:
                self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
                self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
            else:
                self.adam_m = self.adam_v = None
            self.workspace = torch.zeros(int(sz.workspace_bytes), dtype=torch.uint8, device=dev)
            cos, sin = rope_tables(spec.max_position, spec.head_dim, spec.rope_theta)
            self.rope_cos, self.rope_sin = cos.to(dev), sin.to(dev)
            self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
            self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        bufs = L.GgetBuffers(_ptr(self.param_bf16), _ptr(self.master), _ptr(self.adam_m), _ptr(self.adam_v),
                             _ptr(self.grad_bf16), _ptr(self.workspace), _ptr(self.rope_cos), _ptr(self.rope_sin))
        h = C.c_void_p()
        # gget_create clears parts of the workspace and uploads tables on the NULL stream; the arenas above were zero-filled on
        # torch's current stream (possibly a non-blocking side stream): order the two
        torch.cuda.current_stream(dev).synchronize()
        L.check(self.lib.gget_create(C.byref(cfg), C.byref(bufs), C.byref(h)))
        self.h = h
        self.workspace_bytes = int(sz.workspace_bytes)
        # parameter table
        self.varlen_mode: Optional[str] = None     # per-engine override of GGET_VARLEN (see _set_layout)
        self.params: "OrderedDict[str, dict]" = OrderedDict()
        info = L.GgetParamInfo()
        for i in range(self.lib.gget_param_count(self.h)):
            L.check(self.lib.gget_param_info(self.h, i, C.byref(info)))
            shape = tuple(int(info.shape[k]) for k in range(info.ndim))
            n = int(np.prod(shape))
            off = int(info.offset)
            self.params[info.name.decode()] = dict(shape=shape, offset=off, numel=n, layer=int(info.layer))
        self.buckets = []
        o, c = C.c_uint64(), C.c_uint64()
        for b in range(self.lib.gget_bucket_count(self.h)):
            L.check(self.lib.gget_bucket_range(self.h, b, C.byref(o), C.byref(c)))
            self.buckets.append((int(o.value), int(c.value)))
        self.step_count = 0
        self.comm_world = 0     # > 0 once this handle owns an RCCL communicator (comm_init / comm_adopt)
        self._keep = None  # keeps the last batch tensors alive until backward has consumed them

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.gget_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def view(self, name: str, which: str = "master") -> torch.Tensor:
        p = self.params[name]
        arena = {"master": self.master, "bf16": self.param_bf16, "grad": self.grad_bf16, "m": self.adam_m,
                 "v": self.adam_v}[which]
        return arena[p["offset"]: p["offset"] + p["numel"]].view(p["shape"])

    def load_state_dict(self, state: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        missing = [k for k in self.params if k not in state]
        unexpected = [k for k in state if k not in self.params]
        if strict and (missing or unexpected):
            raise KeyError(f"state dict mismatch: missing {missing[:4]} unexpected {unexpected[:4]}")
        for k, p in self.params.items():
            if k in state:
                t = torch.as_tensor(state[k]).to(torch.float32).reshape(p["shape"])
                self.view(k, "master").copy_(t.to(self.device))
        self.sync_params()
        return missing, unexpected

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, self.view(k, "master").detach().clone()) for k in self.params)

    def sync_params(self):
        L.check(self.lib.gget_sync_params(self.h, _stream()))

    def grads(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, self.view(k, "grad")) for k in self.params)

    # ------------------------------------------------------------------ forward / backward / step
    @staticmethod
    def _i64(t, dev):
        return None if t is None else t.to(device=dev, dtype=torch.int64).contiguous()

    def _set_layout(self, att, num_tokens):
        """Token layout of the forward about to run.  `num_tokens`: a host int (real tokens of the batch, no device traffic), the
        string "auto" (the engine sums the key lengths of the 2-D device mask and reads the total back: 4 bytes + one stream
        synchronisation - what the model classes pass for a device-resident mask) or None (unknown: padded layout, the default of this
        low-level class).  GGET_VARLEN=0 keeps the padded layout whatever the caller passed; =sync counts even when the caller passed
        nothing; =nosync never counts on the device.  `self.varlen_mode` (the same strings; set through the model classes'
        `token_layout` attribute) overrides the environment for this engine."""
        import os
        mode = self.varlen_mode if self.varlen_mode is not None else os.environ.get("GGET_VARLEN", "")
        n = None if mode == "0" else num_tokens
        if n is None and mode == "sync":
            n = "auto"
        if n == "auto" and (mode == "nosync" or att is None or att.dim() != 2):
            n = None
        self.set_token_count(L.TOKENS_AUTO if n == "auto" else n)

    def forward_pretrain(self, input_ids, attention_mask, labels=None, sample_wgt=None, position_ids=None, num_tokens=None):
        dev = self.device
        ids = self._i64(input_ids, dev)
        if ids.dim() == 2:
            ids = ids[:, :, None].contiguous()
        B, S, F = ids.shape
        assert F == self.spec.stacked_feat, f"input_ids has {F} stacked features, model expects {self.spec.stacked_feat}"
        att = self._i64(attention_mask, dev)
        lab = self._i64(labels, dev)
        if lab is not None and lab.dim() == 2:
            lab = lab[:, :, None].contiguous()
        wgt = None if sample_wgt is None else sample_wgt.to(device=dev, dtype=torch.float32).contiguous()
        pos = self._i64(position_ids, dev)
        self._keep = (ids, att, lab, wgt, pos)
        self._set_layout(att, num_tokens)   # (full-logit inference runs var-len too since round 5: the logits keep their [B S F, V] cell order)
        if att is not None and att.dim() == 3:      # packed rows: block-diagonal [B,S,S] mask
            assert att.shape == (B, S, S), f"3-D attention mask must be [B,S,S], got {tuple(att.shape)}"
            fn = self.lib.gget_forward_pretrain_packed
        else:
            fn = self.lib.gget_forward_pretrain
        # (a fresh 1-element result tensor per call - a host-side allocator operation - instead of one buffer whose value every caller
        #  had to copy out with a kernel before the next step overwrote it)
        self._loss = torch.empty(1, dtype=torch.float32, device=dev)
        L.check(fn(self.h, _ptr(ids), _ptr(att), _ptr(lab), _ptr(wgt), _ptr(pos), B, S, _ptr(self._loss), _stream()))
        return self._loss[0] if lab is not None else None

    def forward_task(self, input_ids, attention_mask, position_ids=None, task_labels=None, sample_wgt=None,
                     problem: int = L.PROBLEM_SINGLE_LABEL, num_tokens=None):
        dev = self.device
        ids = self._i64(input_ids, dev)
        if ids.dim() == 2:
            ids = ids[:, :, None]
        ids = ids[:, :, : self.spec.stacked_feat].contiguous()
        B, S, _ = ids.shape
        att = self._i64(attention_mask, dev)
        pos = self._i64(position_ids, dev)
        y = None
        if task_labels is not None:
            if problem in (L.PROBLEM_SINGLE_LABEL, L.PROBLEM_AUC, L.PROBLEM_TOKEN_CE):
                y = task_labels.to(device=dev, dtype=torch.int64).contiguous()
            else:
                y = task_labels.to(device=dev, dtype=torch.float32).contiguous()
        wgt = None if sample_wgt is None else sample_wgt.to(device=dev, dtype=torch.float32).contiguous()
        if problem == L.PROBLEM_TOKEN_CE:   # token-level task: labels and logits per row
            assert y is None or tuple(y.shape) == (B, S), f"token-level labels must be [B,S], got {tuple(y.shape)}"
            logits = torch.empty(B, S, self.spec.num_labels, dtype=torch.float32, device=dev)
        else:
            logits = torch.empty(B, self.spec.num_labels, dtype=torch.float32, device=dev)
        hid = torch.empty(B, self.spec.hidden_size, dtype=torch.bfloat16, device=dev)
        self._keep = (ids, att, pos, y, wgt)
        self._set_layout(att, num_tokens)
        self._loss = torch.empty(1, dtype=torch.float32, device=dev)
        L.check(self.lib.gget_forward_task(self.h, _ptr(ids), _ptr(att), _ptr(pos), _ptr(y), _ptr(wgt), problem, B, S,
                                           _ptr(self._loss), _ptr(logits), _ptr(hid), _stream()))
        return (self._loss[0] if y is not None else None), logits, hid

    def set_dropout(self, attention_p: float = 0.0, path_p: float = 0.0, seed: int = 0):
        """Attention dropout / stochastic depth for the NEXT forward+backward (training mode); zeros = eval."""
        L.check(self.lib.gget_set_dropout(self.h, float(attention_p), float(path_p), int(seed) & 0xFFFFFFFF))

    def set_dropout_ex(self, embed_p: float = 0.0, mlp_p: float = 0.0, head_p: float = 0.0):
        """Embedding dropout, the two MLP dropouts and the MLP score head's dropout for the NEXT forward+backward (masks keyed
        by set_dropout's seed)."""
        L.check(self.lib.gget_set_dropout_ex(self.h, float(embed_p), float(mlp_p), float(head_p)))

    def debug_probe(self, enable: bool):
        """Measurement aid (gget_debug_probe): switch the in-step event probe on / off; returns the mean launch durations in ms
        recorded so far (grouped weight-gradient launch, gate|up + GEGLU launch)."""
        out = (C.c_float * 2)()
        L.check(self.lib.gget_debug_probe(self.h, int(bool(enable)), out))
        return float(out[0]), float(out[1])

    def set_focal_gamma(self, gamma: float = 0.0):
        """config.focal_gamma: > 0 = focal loss on the SMTP head (un-weighted path)."""
        L.check(self.lib.gget_set_focal_gamma(self.h, float(gamma)))

    def set_stack_method(self, stack_long: bool):
        """config.stack_method == "long": per-token 1/nnz embedding ratio + per-feature-level SMTP loss weights."""
        L.check(self.lib.gget_set_stack_method(self.h, int(bool(stack_long))))

    def set_raw_embeds(self, raw_embeds, first_label_only: bool = False):
        """inputs_raw_embeds [B,S,embed_dim] of the NEXT forward (config.embed_dim > 0); first_label_only = the smtp_inside mask rule."""
        if raw_embeds is None:
            self._raw_keep = None
            L.check(self.lib.gget_set_raw_embeds(self.h, None, 0))
            return
        assert raw_embeds.dim() == 3 and raw_embeds.shape[-1] == self.cfg.embed_dim, \
            f"inputs_raw_embeds must be [B,S,{self.cfg.embed_dim}], got {tuple(raw_embeds.shape)} (the [B,S,S,e] form is not supported)"
        self._raw_keep = raw_embeds.to(device=self.device, dtype=torch.float32).contiguous()
        L.check(self.lib.gget_set_raw_embeds(self.h, _ptr(self._raw_keep), int(bool(first_label_only))))

    def set_rope_range(self, rope_range: float):
        """config.rope_range: > 0 rescales the position ids of a forward to [0, rope_range) per row (per-token rotary angles)."""
        L.check(self.lib.gget_set_rope_range(self.h, float(rope_range)))

    def set_token_count(self, n_real_tokens: Optional[int]):
        """sum(attention_mask) of the NEXT forward's batch, when the host knows it: switches that step to the var-len (padding-free)
        token layout (gget_set_token_count).  None / 0 = unknown -> padded layout."""
        L.check(self.lib.gget_set_token_count(self.h, int(n_real_tokens or -1)))   # (L.TOKENS_AUTO = -2 passes through)

    def set_option(self, option: int, value: int):
        L.check(self.lib.gget_set_option(self.h, int(option), int(value)))

    def positions_clamped(self) -> bool:
        """True if a forward since the last call met position_ids outside [0, max_position) (they were clamped into the RoPE table);
        clears the flag; synchronises."""
        out = C.c_int32(0)
        L.check(self.lib.gget_position_status(self.h, C.byref(out), _stream()))
        return bool(out.value)

    def deferred_status(self):
        """(positions clamped, var-len token count / label mismatch) since the last call (gget_deferred_status); clears both; synchronises."""
        out = (C.c_int32 * 2)()
        L.check(self.lib.gget_deferred_status(self.h, out, _stream()))
        return bool(out[0]), bool(out[1])

    def varlen_status(self):
        """(ran var-len, rows, count mismatch) of the last forward; synchronises."""
        out = (C.c_int32 * 3)()
        L.check(self.lib.gget_varlen_status(self.h, out, _stream()))
        return bool(out[0]), int(out[1]), bool(out[2])

    def set_auc(self, num_neg: int = 1, seed: int = 0):
        """Negatives per positive and the sampling seed of the NEXT forward_task(problem=PROBLEM_AUC)."""
        L.check(self.lib.gget_set_auc(self.h, int(num_neg), int(seed) & 0xFFFFFFFF))

    def backward(self):
        L.check(self.lib.gget_backward(self.h, 1.0, _stream()))

    def backward_begin(self):
        L.check(self.lib.gget_backward_begin(self.h, 1.0, _stream()))

    def backward_layer(self, i: int):
        L.check(self.lib.gget_backward_layer(self.h, i, _stream()))

    def backward_end(self):
        L.check(self.lib.gget_backward_end(self.h, _stream()))

    def adamw_step(self, lr, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1, max_grad_norm=1.0, grad_scale=1.0):
        self.step_count += 1
        self._gnorm = torch.empty(1, dtype=torch.float32, device=self.device)     # (fresh per step: nothing to copy out, see forward_pretrain)
        L.check(self.lib.gget_adamw_step(self.h, lr, beta1, beta2, eps, weight_decay, max_grad_norm, grad_scale,
                                         self.step_count, _ptr(self._gnorm), _stream()))
        return self._gnorm[0]

    # ------------------------------------------------------------------ data-parallel exchange through the C ABI (RCCL)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().gget_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        L.check(self.lib.gget_comm_init(self.h, int(rank), int(world), C.create_string_buffer(unique_id, 128)))
        self.comm_world = int(world)

    def comm_init_loopback(self, world: int):
        """A peer-less communicator (gget_comm_init_loopback): every all-reduce multiplies the range by `world` - the exchange schedule
        of a `world`-rank job on one GPU."""
        L.check(self.lib.gget_comm_init_loopback(self.h, int(world)))
        self.comm_world = int(world)

    def comm_destroy(self):
        L.check(self.lib.gget_comm_destroy(self.h))
        self.comm_world = 0

    def comm_adopt(self, other: "Engine"):
        """Take over `other`'s communicator (gget_comm_move): used when the model re-creates its engine with larger capacities,
        so that the ranks whose batch grew keep the communicator the other ranks still use (no collective re-init)."""
        if other.comm_world > 0:
            L.check(self.lib.gget_comm_move(self.h, other.h))
            self.comm_world, other.comm_world = other.comm_world, 0

    def allreduce_grads_async(self, bucket: int = -1, fp32_accumulate: bool = False, stream: Optional[torch.cuda.Stream] = None):
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        L.check(self.lib.gget_allreduce_grads_async(self.h, int(bucket), int(bool(fp32_accumulate)), st))

    def allreduce_range_async(self, offset: int, count: int, fp32_accumulate: bool = False, stream: Optional[torch.cuda.Stream] = None):
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        L.check(self.lib.gget_allreduce_range_async(self.h, int(offset), int(count), int(bool(fp32_accumulate)), st))

    # ------------------------------------------------------------------ head outputs
    def head_counts(self):
        c = (C.c_int32 * 2)()
        L.check(self.lib.gget_head_counts(self.h, C.byref(c), _stream()))
        return int(c[0]), int(c[1])

    def head_logits(self) -> torch.Tensor:
        """bf16 [Lm, V] logits of the last pre-train forward (a copy)."""
        _, lm = self.head_counts()
        p, ld = C.c_void_p(), C.c_int32()
        L.check(self.lib.gget_head_logits(self.h, C.byref(p), C.byref(ld)))
        off = p.value - self.workspace.data_ptr()
        raw = self.workspace[off: off + lm * ld.value * 2].view(torch.bfloat16).view(lm, ld.value)
        return raw[:, : self.spec.vocab_size].clone()

    def token_confidence(self, rows: int, mode: int = 0):
        """Arg-max token and unmasking confidence of the first `rows` rows of the head logits of the last forward, computed
        in place on the workspace (no copy of the [rows, V] logits): (conf f32 [rows], token i64 [rows])."""
        p, ld = C.c_void_p(), C.c_int32()
        L.check(self.lib.gget_head_logits(self.h, C.byref(p), C.byref(ld)))
        conf = torch.empty(rows, dtype=torch.float32, device=self.device)
        tok = torch.empty(rows, dtype=torch.int64, device=self.device)
        L.check(self.lib.gget_op_token_confidence(p, ld.value, int(rows), self.spec.vocab_size, int(mode), _ptr(conf), _ptr(tok),
                                                  _stream()))
        return conf, tok

    def hidden_states(self, B: int, S: int) -> torch.Tensor:
        """bf16 [B,S,d] final-normed hidden states of the last forward (a copy; either token layout - gget_hidden_states_grid)."""
        return self._hidden_grid(-1, B, S)

    def layer_hidden_states(self, layer: int, B: int, S: int) -> torch.Tensor:
        """bf16 [B,S,d] residual stream entering decoder layer `layer` (num_layers = leaving the last one) of the last forward (a copy)."""
        return self._hidden_grid(int(layer), B, S)

    def _hidden_grid(self, layer: int, B: int, S: int) -> torch.Tensor:
        out = torch.empty(B, S, self.spec.hidden_size, dtype=torch.bfloat16, device=self.device)
        L.check(self.lib.gget_hidden_states_grid(self.h, layer, _ptr(out), _stream()))
        return out
