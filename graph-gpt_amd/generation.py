"""Iterative unmasking generation (MaskGIT / dLLM style) on top of the pre-train forward - SURVEY.md next item N3.

Restates the reference's `sample_per_batch` + `_batch_unmask_without_for_loop` (src/utils/generation_utils.py:84-237)
for its deterministic settings: temperature 0 (arg-max candidates), no top-p / top-k filtering, `alg` in
{"maskgit_plus", "topk_margin", "entropy"} with `alg_temp` 0 / None.  Each iteration is one forward with `labels=None`
(logits for all B*S*F feature tokens stay in the engine's workspace), one pass of the HIP confidence kernel
(`gget_op_token_confidence`) over those logits, and a per-sample top-k on [B, S*F] confidences.  The stochastic variants
(`alg="origin"`, temperature > 0, Gumbel `alg_temp`) draw from torch's RNG in the reference and are not offered here.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Tuple

import torch

from . import _lib as L

_MODES = {"maskgit_plus": 0, "topk_margin": 1, "entropy": 2}   # GGET_CONF_* of include/gget.h


@dataclasses.dataclass
class GenerationConfig:
    """Fields of the reference's `GenerationConfig` (src/conf/generation/generation_configs.py) that the loop reads."""
    alg: str = "maskgit_plus"
    alg_temp: Optional[float] = None
    steps: int = 512
    eps: float = 1e-3
    temperature: float = 0.0
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    output_history: bool = False
    mask_token_id: int = 1

    def validate(self):
        if self.alg not in _MODES:
            raise NotImplementedError(f"alg={self.alg!r}: only the deterministic confidence-ranked algorithms "
                                      f"{sorted(_MODES)} are provided ('origin' samples with torch's RNG)")
        if self.temperature and self.temperature > 0:
            raise NotImplementedError("temperature > 0 (categorical sampling of candidates) is not provided")
        if (self.top_p is not None and self.top_p < 1) or self.top_k is not None:
            raise NotImplementedError("top-p / top-k filtering only changes sampled candidates; arg-max decoding ignores it")
        if self.alg_temp is not None and self.alg_temp > 0:
            raise NotImplementedError("alg_temp > 0 (Gumbel-perturbed ranking) is not provided")


def unmask_step(x: torch.Tensor, conf: torch.Tensor, cand: torch.Tensor, timesteps: torch.Tensor, i: int,
                cfg: GenerationConfig) -> Tuple[torch.Tensor, int]:
    """One confidence-ranked update (reference :163-236): x [B, N] tokens, conf / cand [B, N] from the head.  Advances the
    step counter until some sample may reveal at least one token; every sample reveals its floor(n_masked * p) most confident
    masked positions."""
    steps = len(timesteps) - 1
    masked = x == cfg.mask_token_id
    n_masked = masked.sum(dim=1)
    k = 0
    n_reveal = torch.zeros_like(n_masked, dtype=torch.int32)
    if int(n_masked.sum().item()) > 0:
        while k == 0 and i < steps:
            t, s = timesteps[i], timesteps[i + 1]
            p = 1 - s / t if i < steps - 1 else 1.0
            n_reveal = torch.floor(n_masked * p).int()
            k = int(n_reveal.max().item())
            i += 1
    if k == 0:
        return x, i
    conf = conf.masked_fill(~masked, float("-inf"))
    order = torch.topk(conf, k=k, dim=1).indices                       # [B, k] most confident first
    new = torch.gather(cand, 1, order)
    surplus = torch.arange(k, device=x.device)[None, :] >= n_reveal[:, None]
    new = torch.where(surplus, torch.full_like(new, cfg.mask_token_id), new)
    x = x.scatter(1, order, new)
    return x, i


@torch.no_grad()
def sample_per_batch(model, cfg: GenerationConfig, *, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                     inputs_raw_embeds=None) -> Tuple[torch.Tensor, Optional[List[torch.Tensor]]]:
    """Fill the `<mask>` cells of input_ids [B, S, F].  Returns (tokens [B, S*F], history or None)."""
    cfg.validate()
    assert input_ids.dim() == 3, "expect [bz, seq, next_n]"
    assert inputs_raw_embeds is None
    model.eval()
    B, S, F = input_ids.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    x = input_ids.to(dev).reshape(B, S * F).clone()
    att = attention_mask.to(dev)
    n_steps = min(int((x == cfg.mask_token_id).sum(dim=-1).max().item()), cfg.steps)
    timesteps = torch.linspace(1, cfg.eps, n_steps + 1, device=dev)
    history = [] if cfg.output_history else None
    mode = _MODES[cfg.alg]
    i = 0
    while i < n_steps:
        model(input_ids=x.view(B, S, F), attention_mask=att, labels=None)      # logits for all B*S*F rows, in the workspace
        conf, cand = model._engine.token_confidence(B * S * F, mode)
        x, i = unmask_step(x, conf.view(B, S * F), cand.view(B, S * F), timesteps, i, cfg)
        if history is not None:
            history.append(x.view(B, S, F).clone())
    return x, history
