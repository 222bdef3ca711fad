"""CPU oracle for the GraphGPT Graph-Eulerian-Transformer hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module; the product path (graph-gpt_amd/) never does and fails loudly without its HIP library.

What it is: a plain PyTorch-CPU restatement (explicit tensor arithmetic, no nn.Module, no
transformers import) of the reference's algorithm for SURVEY.md §8a rows A1-A12.  Each function
cites the reference file:line (paths relative to the reference repo root; "hf:" = the un-vendored
third-party `transformers.models.llama.modeling_llama`, pinned ==4.53.3 by the reference's
requirements.txt:20, line numbers from the 5.15.0 copy the survey read).

Parity status: PINNED against golden vectors captured by importing the real reference in the build
container (`tools/make_golden.py` -> `tests/golden/*.npz`, checked by tests/test_oracle_golden.py).
The reference itself ships no tests for this path, so those fixtures are the anchor.

dtype: pass torch.float32 for the exact restatement; torch.bfloat16 reproduces the reference's
bf16 module path (bf16 parameters/activations with fp32 islands: RMSNorm statistics, RoPE tables,
softmax, cross-entropy) which is what the HIP kernels are compared against.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as Fnn

LABEL_PAD = -100
_EPSILON = 1e-7  # reference modeling_helpers.py `_EPSILON`


# --------------------------------------------------------------------------- K1  (row A1)
def stacked_embed(emb_w: torch.Tensor, input_ids: torch.Tensor, gate_w: Optional[torch.Tensor], embed_keep=None,
                  stack_long: bool = False):
    """`_get_stacked_inputs_embeds` (modeling_helpers.py:89-114) + `StackedFeatAggregation.forward`
    (modeling_common.py:127-135).  ids [B,S,F] (or [B,S]) -> [B,S,d]; returns (embeds, in_).
    `embed_keep`: multipliers (0 or 1/(1-p)) of `embed_dropout` on the gathered rows [B,S,F,d] (:96-98); None = eval mode.
    `stack_long`: config.stack_method == "long" - the per-token 1 / (non-zero ids) ratio of :106-110."""
    e = Fnn.embedding(input_ids, emb_w, padding_idx=0)   # nn.Embedding(padding_idx=pad_token_id): the pad row takes no gradient
    if embed_keep is not None:
        e = e * torch.as_tensor(embed_keep).reshape(e.shape).to(e.dtype)
    if input_ids.dim() == 3:
        if gate_w is not None:
            e = torch.einsum("nsfd,fd->nsd", e, gate_w)
        else:
            e = torch.sum(e, dim=-2)
        in_ = input_ids[:, :, 0]
        if stack_long:
            nonzero_feat = (input_ids != 0).sum(dim=-1, keepdim=True) + 1e-7     # int64 + float -> fp32
            ratio = torch.clamp(1 / nonzero_feat.to(e.dtype), max=1)
            e = e * ratio
    else:
        in_ = input_ids
    return e, in_


# --------------------------------------------------------------------------- K4  (row A4a)
def raw_embeds_branch(spec, p, raw, dtype, labels=None, first_label_only=False, raw_keep=None):
    """Raw-embedding inputs (config.embed_dim > 0).  Pre-train (modeling_pretrain.py:131-149, `labels` given): rows whose labels are all
    set (smtp_inside: whose first label is set) are replaced by emb_mask_token; then embed_layernorm, raw_embed_dropout (`raw_keep`
    multipliers; None = eval) and embed_proj.  Fine-tune (modeling_helpers.py:127-139): the same without the mask token."""
    x = raw.to(dtype)
    if labels is not None:
        if labels.dim() == 2:
            labels = labels[:, :, None]
        embed_mask = (labels[:, :, 0:1] == LABEL_PAD) if first_label_only else (labels == LABEL_PAD).sum(dim=-1, keepdim=True).to(torch.bool)
        x = embed_mask.to(dtype) * x + (~embed_mask).to(dtype) * p["emb_mask_token"].reshape(1, 1, -1)
    x = rmsnorm(x, p["embed_layernorm.weight"], spec.rms_eps)
    if raw_keep is not None:
        x = x * torch.as_tensor(raw_keep).reshape(x.shape).to(x.dtype)
    return Fnn.linear(x, p["embed_proj.weight"])


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float):
    """hf LlamaRMSNorm.forward :62-67 - statistics in fp32, cast back, then scale by weight."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


# --------------------------------------------------------------------------- K3  (row A4)
def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, dtype):
    """hf LlamaRotaryEmbedding.forward :111-127 (default rope): fp32 tables cast to the act dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids.to(torch.float32)[:, :, None] * inv_freq[None, None, :]   # [B,S,dh/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x):
    """hf rotate_half :130-135."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """hf apply_rotary_pos_emb :138-160 (q,k are [B,H,S,dh]; cos/sin [B,S,dh])."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (_rotate_half(q) * sin), (k * cos) + (_rotate_half(k) * sin)


# --------------------------------------------------------------------------- K2  (row A3)
def additive_mask(attention_mask: torch.Tensor, S: int, dtype, causal: bool):
    """Key-padding mask of `_update_causal_mask` -> `_prepare_4d_attention_mask`
    (modeling_helpers.py:38-48): 0 where the key is real, finfo.min where it is padding; no causal
    term when causal_attention=False.  With causal_attention=True the reference hands the 2-D mask to
    hf LlamaModel which builds causal AND padding (hf :394-400).  Returns [B,1,S,S]."""
    B = attention_mask.shape[0]
    neg = torch.finfo(dtype).min
    if attention_mask.dim() == 3:
        # packed sequences: block-diagonal [B,S,S] mask, `_expand_mask_from_3d_mask` (modeling_helpers.py:51-64)
        inv = 1.0 - attention_mask[:, None, :, :].to(dtype)
        return inv.masked_fill(inv.to(torch.bool), neg)
    m = torch.zeros(B, 1, S, S, dtype=dtype)
    m = m.masked_fill(attention_mask[:, None, None, :] == 0, neg)
    if causal:
        tri = torch.ones(S, S, dtype=torch.bool).tril()
        m = m.masked_fill(~tri[None, None], neg)
    return m


# --------------------------------------------------------------------------- K5-K8 (row A4b)
def attention(x, p, pre, mask4d, cos, sin, H, dh, keep=None):
    """hf LlamaAttention.forward :243-281 with eager_attention_forward :191-214.  `keep` [B,H,S,S]: the dropout multiplier
    (0 or 1/(1-p)) applied to the softmax output, hf :210 `nn.functional.dropout(attn_weights, p, training)` with the mask
    made explicit (None = eval mode / p = 0)."""
    B, S, d = x.shape
    q = Fnn.linear(x, p[pre + "q_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    k = Fnn.linear(x, p[pre + "k_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    v = Fnn.linear(x, p[pre + "v_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    w = torch.matmul(q, k.transpose(2, 3)) * (dh ** -0.5)
    w = w + mask4d
    w = Fnn.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    if keep is not None:
        w = w * keep.to(w.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().view(B, S, d)
    return Fnn.linear(o, p[pre + "o_proj.weight"])


# --------------------------------------------------------------------------- K9  (row A4c)
def mlp(x, p, pre, keep=None):
    """hf LlamaMLP.forward :174-176 with hidden_act="gelu" = exact erf GELU
    (reference configs/model/base.yaml:21).  `keep` = (act_keep [B,S,ff], out_keep [B,S,d]): the multipliers of the
    reference's own MLP subclass used when mlp_pdrop > 0 (utils_graphgpt.py:69-80: mlp_act_dropout on act(gate)*up,
    mlp_dropout on down_proj's output); None = eval mode."""
    g = Fnn.gelu(Fnn.linear(x, p[pre + "gate_proj.weight"]))
    u = Fnn.linear(x, p[pre + "up_proj.weight"])
    hgu = g * u
    if keep is not None:
        hgu = hgu * torch.as_tensor(keep[0]).reshape(hgu.shape).to(hgu.dtype)
    out = Fnn.linear(hgu, p[pre + "down_proj.weight"])
    if keep is not None:
        out = out * torch.as_tensor(keep[1]).reshape(out.shape).to(out.dtype)
    return out


def backbone(spec, p: Dict[str, torch.Tensor], x, attention_mask, position_ids, collect=None, path_mult=None, attn_keep=None,
             mlp_keep=None):
    """hf LlamaModel.forward :367-418 / LlamaDecoderLayer.forward :295-325; LayerScale variant
    utils_graphgpt.LlamaDecoderLayer.forward (utils_graphgpt.py:107-173).  `path_mult(layer, which)` -> [B] tensor of
    DropPath multipliers (0 or 1/keep_prob per sample, utils_graphgpt.py:64-66 / BeitDropPath); None = eval mode.
    `attn_keep(layer)` -> [B,H,S,S] attention-dropout multipliers of that layer (see attention); `mlp_keep(layer)` -> the
    pair of MLP dropout multipliers (see mlp)."""
    B, S, d = x.shape
    rope_range = float(getattr(spec, "rope_range", 0) or 0)
    if rope_range > 0 and position_ids is not None:
        # utils_graphgpt.reset_pos_ids (:574-581) via resolve_forward_defaults (modeling_common.py:185-203): per-row rescaling to
        # [0, rope_range), float positions from here on
        position_ids = position_ids.float() * rope_range / (position_ids.max(dim=-1, keepdim=True).values + 1).float()
    if position_ids is None:
        position_ids = torch.arange(S)[None, :].expand(B, S)      # hf :389-392
    cos, sin = rope_cos_sin(position_ids, spec.head_dim, spec.rope_theta, x.dtype)
    mask4d = additive_mask(attention_mask, S, x.dtype, spec.causal)
    for i in range(spec.num_layers):
        pre = f"model.layers.{i}."
        h = rmsnorm(x, p[pre + "input_layernorm.weight"], spec.rms_eps)
        a = attention(h, p, pre + "self_attn.", mask4d, cos, sin, spec.num_heads, spec.head_dim,
                      keep=attn_keep(i) if attn_keep is not None else None)
        if spec.layer_scale_init > 0:
            a = p[pre + "lambda_1"] * a
        if path_mult is not None:
            a = a * path_mult(i, 0)[:, None, None].to(a.dtype)
        x = x + a
        h = rmsnorm(x, p[pre + "post_attention_layernorm.weight"], spec.rms_eps)
        m = mlp(h, p, pre + "mlp.", keep=mlp_keep(i) if mlp_keep is not None else None)
        if spec.layer_scale_init > 0:
            m = p[pre + "lambda_2"] * m
        if path_mult is not None:
            m = m * path_mult(i, 1)[:, None, None].to(m.dtype)
        x = x + m
        if collect is not None:
            collect.append(x)
    return rmsnorm(x, p["model.norm.weight"], spec.rms_eps)


# --------------------------------------------------------------------------- K11-K14 (rows A5-A7)
def smtp_head(spec, p, hidden, labels, sample_wgt=None, focal_gamma=0.0, stack_long=False):
    """`prepare_for_stacked_feat_labels` (modeling_helpers.py:362-393), "short" stacking:
    no wgt -> `_prepare_for_stacked_feat_labels_per_mix_lvl` (:263-301);
    wgt    -> `_prepare_for_stacked_feat_labels_wgt_per_feat_lvl` (:345-359);
    then lm_head (modeling_pretrain.py:218) and `_get_ce_loss` (:145-177) or `_get_dlm_ce_loss`
    (:180-198) / (B*S*F) (modeling_pretrain.py:230-236).  labels None => logits for every cell.
    "long" stacking (`stack_long`): `_prepare_for_stacked_feat_labels_per_feat_lvl` (:327-342) - the wgt path with every
    labelled cell of a sample weighted by 1 / (labelled cells of the sample + 1e-7), whatever sample_wgt was passed."""
    B, S, d = hidden.shape
    n = spec.next_n_token
    proj = p.get("n_token_proj.weight")

    def _proj(h):
        return Fnn.linear(h, proj) if proj is not None else h

    wgt = None
    if labels is not None and labels.dim() == 2:
        labels = labels[:, :, None]
    if stack_long:
        hs = _proj(hidden).reshape(B, S, n, d)
        mask_m = labels != LABEL_PAD
        hs = hs[mask_m]
        wgt = mask_m.float()
        wgt = (wgt / (wgt.sum(dim=-1).sum(dim=-1)[:, None, None] + 1e-7))[mask_m]
        labels = labels[mask_m]
    elif sample_wgt is None:
        if labels is not None:
            mask = labels != LABEL_PAD
            mask_m = mask.any(dim=-1)
            mask = mask[mask_m]
        else:
            mask_m = torch.ones(B, S, dtype=torch.bool)
        hs = _proj(hidden[mask_m]).reshape(-1, d)
        if labels is not None:
            labels = labels[mask_m][mask]
            hs = hs[mask.reshape(-1)]
    else:
        hs = _proj(hidden).reshape(B, S, n, d)
        mask_m = labels != LABEL_PAD
        hs = hs[mask_m]
        wgt = sample_wgt[:, None, None].repeat(1, S, n)[mask_m]
        labels = labels[mask_m]
    logits = Fnn.linear(hs, p["lm_head.weight"])
    loss = None
    if labels is not None:
        if wgt is None and focal_gamma > 0:
            # utils_graphgpt.FocalLoss.forward (:356-376): -(1 - pt)^gamma * log pt, pt DETACHED, mean over the rows
            logpt = Fnn.log_softmax(logits.float(), dim=-1).gather(1, labels.view(-1, 1)).view(-1)
            loss = (-1 * (1 - logpt.detach().exp()) ** focal_gamma * logpt).mean()
        elif wgt is None:
            loss = Fnn.cross_entropy(logits.float(), labels)
        else:
            l_ = Fnn.cross_entropy(logits.float(), labels, reduction="none")
            loss = (l_ * wgt.view(-1)).float().sum() / (B * S * n)
    return loss, logits


def pretrain_forward(spec, p, input_ids, attention_mask, labels=None, sample_wgt=None,
                     position_ids=None, collect=None, embed_keep=None, mlp_keep=None, focal_gamma=0.0, stack_long=False,
                     inputs_raw_embeds=None, raw_keep=None, smtp_inside=False):
    """`GraphGPTPretrainBase.forward` (modeling_pretrain.py:152-266), generative head only."""
    x, _ = stacked_embed(p["model.embed_tokens.weight"], input_ids, p.get("stacked_feat_agg.weight"), embed_keep=embed_keep,
                         stack_long=stack_long)
    if inputs_raw_embeds is not None:
        x = x + raw_embeds_branch(spec, p, inputs_raw_embeds, x.dtype, labels, smtp_inside, raw_keep)
    hidden = backbone(spec, p, x, attention_mask, position_ids, collect, mlp_keep=mlp_keep)
    loss, logits = smtp_head(spec, p, hidden, labels, sample_wgt, focal_gamma=focal_gamma, stack_long=stack_long)
    return dict(head1_loss=loss, head1_logits=logits, hidden=hidden)


# --------------------------------------------------------------------------- K15 (row A10)
def auc_loss(y_pred, y_true, num_neg, idx):
    """`auc_loss` + `_auc_loss` (src/utils/loss_utils.py:25-53): y_pred 1-D logits, y_true 1-D {0,1}; every positive is
    paired with num_neg negatives `y_pred_neg[idx]`, idx = torch.randperm(P * num_neg) % n_neg in the reference - here an
    INPUT (recorded draws: the fixture's, or graph-gpt_amd.modeling.auc_pairs for the engine's counter hash)."""
    pos = y_pred[y_true.bool()]
    neg = y_pred[(1 - y_true).bool()][torch.as_tensor(idx, dtype=torch.int64)]
    return torch.square(1 - (pos.reshape(-1, 1) - neg.reshape(-1, num_neg))).mean()


def task_forward(spec, p, input_ids, attention_mask, position_ids=None, task_labels=None,
                 sample_wgt=None, problem_type="single_label_classification", loss_type=None, path_mult=None, attn_keep=None,
                 num_neg=1, auc_idx=None, embed_keep=None, mlp_keep=None, head_keep=None, stack_long=False,
                 inputs_raw_embeds=None, raw_keep=None):
    """`GraphGPTTaskModel.forward` (modeling_finetune.py:236-326) + `calculate_task_loss`
    (:167-234) + `_get_sequence_len` (modeling_helpers.py:78-86); Linear score head, "last" pooling."""
    if input_ids.dim() == 3:
        input_ids = input_ids[:, :, : spec.stacked_feat]
    x, in_ = stacked_embed(p["model.embed_tokens.weight"], input_ids, p.get("stacked_feat_agg.weight"), embed_keep=embed_keep,
                           stack_long=stack_long)
    if inputs_raw_embeds is not None:
        x = x + raw_embeds_branch(spec, p, inputs_raw_embeds, x.dtype, None, False, raw_keep)
    hidden = backbone(spec, p, x, attention_mask, position_ids, path_mult=path_mult, attn_keep=attn_keep, mlp_keep=mlp_keep)
    B = hidden.shape[0]
    seq_len = (in_ != spec.pad_token_id).sum(-1) - 1
    pooled_h = hidden[torch.arange(B), seq_len]
    if len(getattr(spec, "head_mlp", ())) > 0:
        # `MLP` head (src/utils/modules_utils.py:8-34): x = Linear_i(dropout(act(x))) for every Linear, activation first; the
        # reference applies it to every row and indexes the pooled one (modeling_finetune.py:281-296) - row-wise, so the same
        pooled = pooled_h
        for i in range(len(spec.head_mlp) + 1):
            pooled = Fnn.gelu(pooled)
            if head_keep is not None:
                pooled = pooled * torch.as_tensor(head_keep(i)).reshape(pooled.shape).to(pooled.dtype)
            pooled = Fnn.linear(pooled, p[f"score.mlp_modules.{i}.weight"], p.get(f"score.mlp_modules.{i}.bias"))
    else:
        logits = Fnn.linear(hidden, p["score.weight"], p.get("score.bias"))
        pooled = logits[torch.arange(B), seq_len]
        if loss_type == "token_ce":
            pooled = logits     # get_logits_for_token_lvl_task (modeling_finetune.py:162-164): the all-row logits are what is returned
    loss = None
    if task_labels is not None:
        if problem_type == "single_label_classification" and loss_type == "token_ce":
            # modeling_finetune.py:198-202: CrossEntropyLoss over every row, labels [B,S] with -100 (ignore_index) on the unlabelled ones
            loss = Fnn.cross_entropy(logits.view(-1, spec.num_labels).float(), task_labels.view(-1))
        elif problem_type == "regression":
            y = task_labels.to(pooled.dtype)
            if loss_type == "l1":
                loss = Fnn.l1_loss(pooled.squeeze(), y.squeeze())
            else:
                loss = Fnn.mse_loss(pooled.squeeze(), y.squeeze())
        elif problem_type == "single_label_classification" and loss_type == "auc":
            lg = pooled.view(-1, spec.num_labels)
            loss = auc_loss(lg[:, 1].float() - lg[:, 0].float(), task_labels.view(-1), num_neg, auc_idx)
        elif problem_type == "single_label_classification":
            if sample_wgt is None:
                loss = Fnn.cross_entropy(pooled.view(-1, spec.num_labels).float(), task_labels.view(-1))
            else:
                l_ = Fnn.cross_entropy(pooled.view(-1, spec.num_labels).float(), task_labels.view(-1),
                                       reduction="none")
                loss = (l_.float().view(-1) * sample_wgt.float().view(-1)).sum() / sample_wgt.float().sum()
        elif problem_type == "multi_label_classification":
            is_l = task_labels == task_labels
            loss = Fnn.binary_cross_entropy_with_logits(pooled[is_l], task_labels[is_l])
        else:
            raise ValueError(problem_type)
    return dict(task_loss=loss, task_logits=pooled.float(), task_hidden_states=pooled_h, hidden=hidden)


# --------------------------------------------------------------------------- params / grads
def to_params(state: Dict[str, "object"], dtype=torch.float32, requires_grad=True):
    out = {}
    for k, v in state.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def loss_and_grads(fn, p: Dict[str, torch.Tensor], loss_key: str, **kw):
    for t in p.values():
        t.grad = None
    out = fn(p, **kw)
    out[loss_key].backward()
    grads = {k: (t.grad.detach().clone() if t.grad is not None else torch.zeros_like(t)) for k, t in p.items()}
    return out, grads


# --------------------------------------------------------------------------- K16 (rows A11/A12)
def clip_coef(grads: Dict[str, torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ as called at training_utils.py:72 (L2, eps 1e-6, clamp 1)."""
    tot = torch.sqrt(sum((g.float() ** 2).sum() for g in grads.values()))
    return float(torch.clamp(max_norm / (tot + 1e-6), max=1.0)), float(tot)


def adamw_step(master: Dict[str, torch.Tensor], grads, m, v, step: int, lr, beta1, beta2, eps, wd,
               max_grad_norm: float = 0.0):
    """One clip-then-AdamW update on fp32 master weights: torch.optim.AdamW semantics
    (opt_utils.py:18-24; decoupled decay `p *= 1-lr*wd`, bias-corrected m/v,
    denom = sqrt(v)/sqrt(bc2) + eps) = DeepSpeed FusedAdam adam_w_mode (ds_config2_pt.json:11-19).
    `step` is 1-based.  Returns the gradient global norm (pre-clip)."""
    coef, gnorm = (1.0, 0.0)
    if max_grad_norm and max_grad_norm > 0:
        coef, gnorm = clip_coef(grads, max_grad_norm)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, w in master.items():
        g = grads[k].float() * coef
        m[k].mul_(beta1).add_(g, alpha=1 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        w.mul_(1 - lr * wd)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m[k], denom, value=-lr / bc1)
    return gnorm


def one_cycle_lr(step: int, max_lr: float, total_steps: int, pct_start: float, min_lr: float = 0.0):
    """torch OneCycleLR(anneal="cos", three_phase=False, div_factor 25) as configured by
    `_py_one_cycle` (loss_utils.py:322-367): value at 0-based scheduler step `step`."""
    div = 25.0
    initial = max_lr / div
    final = min_lr if min_lr > 0 else initial / 1e4
    up_end = float(pct_start * total_steps) - 1
    down_end = total_steps - 1

    def cos_anneal(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    if step <= up_end:
        return cos_anneal(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos_anneal(max_lr, final, (step - up_end) / (down_end - up_end))


def warmup_decay_lr(step: int, max_lr: float, min_lr: float, warmup: int, total: int):
    """DeepSpeed WarmupDecayLR (log warmup then linear decay), the scheduler named by
    examples/ds_config2_pt.json:20-28.  DeepSpeed is not installed in the build container, so this
    restates its published formula (deepspeed/runtime/lr_schedules.py); parity for it is UNPINNED."""
    if step < warmup:
        gamma = math.log(step + 1) / math.log(max(2, warmup))
    else:
        gamma = max(0.0, (total - step) / max(1.0, total - warmup))
    return min_lr + (max_lr - min_lr) * gamma


# ----------------------------------------------------------------------------- in-model SMTP masking (row A9 / N1)
def smtp_2d_inputs_labels(input_ids, node_idx, u_sample, u_rate, u_cell, token_shift, u_replace, *, smtp_2d_rate: float,
                          power: float, replace_rate: float, vocab: int, global_2d_mask: bool = False,
                          mask_token_id: int = 1, label_pad_token_id: int = -100):
    """reference prepare_for_2d_smtp_inputs_labels (src/models/graphgpt/modeling_helpers.py:399-449) and
    _get_gaussian_rnd_tokens (:460-468) with the random draws passed in, in the order the reference makes them:
    u_sample [B] ~U (sample mask, :418-420), u_rate [B] ~U (mask rate of the sample, :427), u_cell [B,S,F] ~U (per
    (node, feature) draw, :429-432), token_shift [B,S,F] = randn*10 before rounding (:463-464), u_replace [B,S,F] ~U (:444-446).
    The 3-D `pos` branch (:422-425) is outside the hot path (no 3-D models)."""
    bz = input_ids.shape[0]
    bz_idx = torch.arange(bz).view(-1, 1)
    sample_mask = (u_sample.view(bz, 1, 1) < smtp_2d_rate)
    mask_per_node = u_cell > (u_rate.view(bz, 1, 1).to(torch.float32) ** power)
    if not global_2d_mask:
        mask_per_node = mask_per_node & sample_mask
    mask_per_token = mask_per_node[bz_idx, node_idx]
    mask_per_token = mask_per_token & (input_ids > 0)
    labels = input_ids.clone().masked_fill_(~mask_per_token, label_pad_token_id)
    ids = input_ids.clone().masked_fill_(mask_per_token, mask_token_id)
    shift = token_shift.round().long().masked_fill(input_ids <= 0, 0)
    rnd_tokens = (input_ids + shift) % vocab
    replace_mask = mask_per_token & (u_replace < replace_rate)
    ids = ids * (~replace_mask).long() + rnd_tokens * replace_mask.long()
    return ids, labels


# ----------------------------------------------------------------------------- generation loop (next item N3)
def top_p_logits(logits, top_p):
    """reference top_p_logits (src/utils/generation_utils.py:22-34), with the sort made stable (ties keep index order)."""
    sorted_logits, sorted_indices = torch.sort(logits, descending=True, stable=True)
    cumulative = torch.cumsum(Fnn.softmax(sorted_logits, dim=-1), dim=-1)
    remove = cumulative > top_p
    remove[..., 1:] = remove[..., :-1].clone()
    remove[..., 0] = 0
    mask = torch.zeros_like(logits, dtype=torch.bool).scatter_(-1, sorted_indices, remove)
    return logits.masked_fill(mask, torch.finfo(logits.dtype).min)


def top_k_logits(logits, top_k):
    """reference top_k_logits (:37-42)."""
    top_k = min(top_k, logits.size(-1))
    remove = logits < torch.topk(logits, top_k)[0][..., -1, None]
    return logits.masked_fill(remove, torch.finfo(logits.dtype).min)


def sample_tokens(logits, temperature=0.0, top_p=None, top_k=None, margin_confidence=False, neg_entropy=False,
                  u=None, x0_override=None):
    """reference sample_tokens (:45-82).  The categorical draw (`dists.Categorical(probs).sample()` in the reference) is made
    explicit: `x0_override` replays recorded draws (the reference fixture), otherwise `u` in [0,1) per row selects by inverse
    CDF in index order (first c with cumsum(probs) > u) - the convention of the HIP sampling kernel."""
    if temperature > 0:
        logits = logits / temperature
    if top_p is not None and top_p < 1:
        logits = top_p_logits(logits, top_p)
    if top_k is not None:
        logits = top_k_logits(logits, top_k)
    probs = torch.softmax(logits, dim=-1)
    if temperature > 0:
        if x0_override is not None:
            x0 = x0_override
        else:
            cdf = torch.cumsum(probs, dim=-1)
            x0 = (cdf > u[..., None]).float().argmax(dim=-1)
            none = ~(cdf > u[..., None]).any(dim=-1)
            if none.any():   # u beyond the accumulated mass (rounding): the last token with probability
                last = probs.shape[-1] - 1 - (probs.flip(-1) > 0).float().argmax(dim=-1)
                x0 = torch.where(none, last, x0)
        confidence = torch.gather(probs, -1, x0.unsqueeze(-1)).squeeze(-1)
    else:
        confidence, x0 = probs.max(dim=-1)
    if margin_confidence:
        sp, _ = torch.sort(probs, dim=-1, descending=True)
        confidence = sp[..., 0] - sp[..., 1]
    if neg_entropy:
        confidence = torch.sum(probs * torch.log(probs + 1e-10), dim=-1)
    return confidence, x0


def sample_tokens_t0(logits: torch.Tensor, margin_confidence: bool = False, neg_entropy: bool = False):
    """reference sample_tokens at temperature 0 without top-p / top-k (src/utils/generation_utils.py:45-82)."""
    return sample_tokens(logits, margin_confidence=margin_confidence, neg_entropy=neg_entropy)


def sample_per_batch(logits_fn, input_ids: torch.Tensor, *, alg: str, steps: int, eps: float, mask_token_id: int,
                     conf_fn=None, temperature: float = 0.0, top_p=None, top_k=None, alg_temp=None, draw_fn=None,
                     surplus: str = "reference"):
    """reference sample_per_batch + _batch_unmask_without_for_loop (generation_utils.py:84-237), every algorithm.
    `logits_fn(ids [B,S,F])` returns logits [B*S*F, V].  `conf_fn(iteration, logits [B,N,V]) -> (confidence, candidates)`
    replaces sample_tokens + the Gumbel perturbation (tests feed the HIP kernel's outputs through it); otherwise
    `draw_fn(iteration) -> dict` supplies the random draws the reference takes from torch's RNG: "u_cat" [B,N] (inverse-CDF
    uniforms) or "x0" [B,N] (recorded categorical draws), "u_gumbel" [B,N], "u_transfer" [B,N].
    surplus = "reference": the fixed-k scatter of the reference (surplus ranks rewrite <mask>, also over -inf positions picked by
    torch.topk); "skip": only ranks < n_reveal[b] are written, ranking by stable descending sort (the engine's rule).
    Returns (tokens [B, S*F], history list of [B,S,F])."""
    bz, seq, next_n = input_ids.shape
    x = input_ids.clone().view(bz, seq * next_n)
    m = x == mask_token_id
    n_steps = min(int(torch.max(m.sum(dim=-1).float()).item()), steps)
    timesteps = torch.linspace(1, eps, n_steps + 1)
    hist = []
    i = it = 0
    while i < n_steps:
        logits = logits_fn(x.view(bz, seq, next_n)).view(bz, seq * next_n, -1)
        mask_index = x == mask_token_id
        d = draw_fn(it) if draw_fn is not None else {}
        if alg == "origin":
            t, s = timesteps[i], timesteps[i + 1]
            p_transfer = 1 - s / t if i < n_steps - 1 else 1.0
            if conf_fn is not None:
                _, cand = conf_fn(it, logits)
            else:
                _, cand = sample_tokens(logits, temperature=temperature, top_p=top_p, top_k=top_k, u=d.get("u_cat"),
                                        x0_override=d.get("x0"))
            transfer = d["u_transfer"] < p_transfer
            x = torch.where(mask_index & transfer, cand, x)
            i += 1
        else:
            k = 0
            num_masked = mask_index.sum(dim=1)
            num_all = num_masked.sum().item()
            num_transfer = torch.zeros_like(num_masked).int()
            while (k == 0) and (num_all > 0) and (i < n_steps):
                t, s = timesteps[i], timesteps[i + 1]
                p_transfer = 1 - s / t if i < n_steps - 1 else 1.0
                num_transfer = torch.floor(num_masked * p_transfer).int()
                k = num_transfer.max().item()
                i += 1
            if conf_fn is not None:
                confidence, cand = conf_fn(it, logits)
                confidence = confidence.clone()
                confidence[~mask_index] = -torch.inf
            else:
                confidence, cand = sample_tokens(logits, temperature=temperature, top_p=top_p, top_k=top_k,
                                                 margin_confidence=(alg == "topk_margin"), neg_entropy=(alg == "entropy"),
                                                 u=d.get("u_cat"), x0_override=d.get("x0"))
                confidence = confidence.clone()
                confidence[~mask_index] = -torch.inf
                if alg_temp is not None and alg_temp > 0:
                    confidence = confidence / alg_temp
                    confidence = confidence + (-torch.log(-torch.log(d["u_gumbel"] + 1e-9) + 1e-9))
            if surplus == "reference":
                _, idx = torch.topk(confidence, k=k, dim=1)
                updates = torch.gather(cand, 1, idx)
                mask_out = torch.arange(k)[None, :] >= num_transfer[:, None]
                final = torch.where(mask_out, mask_token_id, updates)
                x.scatter_(1, idx, final)
            else:
                idx = torch.sort(confidence, dim=1, descending=True, stable=True).indices[:, :k]
                updates = torch.gather(cand, 1, idx)
                keep = torch.arange(k)[None, :] < num_transfer[:, None]
                x.scatter_(1, idx, torch.where(keep, updates, torch.gather(x, 1, idx)))
        it += 1
        hist.append(x.view(bz, seq, next_n).clone())
    return x, hist


def smtp_mask_ratio(r: float, umr_min: float, umr_max: float, power: float):
    """polynomial schedule of prepare_inputs_for_pretrain_mlm (src/utils/tokenizer_utils.py:259-271), python floats:
    t = umr_min + (umr_max - umr_min) r ; mask ratio alpha = 1 - t^power ; dLM weight = power / t."""
    t = umr_min + (umr_max - umr_min) * r
    return 1 - t ** power, power / t


def mask_stacked_input_ids_v2(input_ids, idx_masked, mask_token_id: int = 1, pad_token_id: int = 0):
    """reference _mask_stacked_input_ids_v2 (src/utils/tokenizer_utils.py:112-148) with mask_token_precent (1, 0, 0) and
    the sampled cell list passed in (`random.sample(range(seq*dim), k=ceil(seq*dim*mask_ratio))` in the reference):
    input_ids [seq, dim] (numpy int64).  Returns (masked ids, labels) as numpy arrays."""
    import numpy as np
    ids = np.array(input_ids)
    seq, dim = ids.shape
    labels = np.full((seq, dim), -100)
    for idx in idx_masked:
        i, j = divmod(int(idx), dim)          # np.ndindex((seq, dim)) order
        labels[i, j] = ids[i, j]
        if ids[i, j] != pad_token_id:
            ids[i, j] = mask_token_id
    return ids, labels
