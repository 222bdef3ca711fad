"""Iterative unmasking generation (MaskGIT / dLLM style) on top of the pre-train forward - SURVEY.md next item N3.

Restates the reference's `sample_per_batch` + `_batch_unmask_without_for_loop` + `sample_tokens`
(src/utils/generation_utils.py:22-237) with every setting: `alg` in {"origin", "maskgit_plus", "topk_margin", "entropy"},
candidate temperature / top-p / top-k, Gumbel-perturbed ranking (`alg_temp`).  Each iteration is one forward with
`labels=None` (logits for all B*S*F feature tokens stay in the engine's workspace), one pass of the HIP sampling kernel
(`gget_op_token_sample`: filters, softmax, candidate, confidence, Gumbel noise) over those logits, and either the "origin"
update kernel (`gget_op_unmask_origin`) or a per-sample ranking on the [B, S*F] confidences.

Randomness: where the reference draws from torch's global RNG (Categorical.sample, torch.rand, torch.rand_like) the kernels use
a counter hash of (seed, iteration, stream, row / cell) - the same scheme as the SMTP masking kernels - so a run is a pure
function of `GenerationConfig.seed` and `draws()` below regenerates the uniforms bit-exactly for the parity tests.

One deliberate difference: the reference scatters a fixed k = max_b(n_reveal[b]) positions per sample and writes <mask> over
the surplus; when k exceeds a sample's masked count the surplus targets are already-revealed positions picked by torch.topk's
tie order among -inf confidences (device dependent), i.e. it re-masks arbitrary known tokens.  Here only the first
n_reveal[b] ranked positions of a sample are written (ranking = stable descending sort: ties go to the lower index), which
is deterministic on every device and never destroys revealed tokens.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from .smtp import _rng24

_MODES = {"origin": 0, "maskgit_plus": 0, "topk_margin": 1, "entropy": 2}   # GGET_CONF_* of include/gget.h


@dataclasses.dataclass
class GenerationConfig:
    """Fields of the reference's `GenerationConfig` (src/conf/generation/generation_configs.py) that the loop reads, plus the
    seed of the counter-hash draws."""
    alg: str = "maskgit_plus"
    alg_temp: Optional[float] = None
    steps: int = 512
    eps: float = 1e-3
    temperature: float = 0.0
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    output_history: bool = False
    mask_token_id: int = 1
    seed: int = 0

    def validate(self):
        if self.alg not in _MODES:
            raise ValueError(f"alg={self.alg!r}: expected one of {sorted(_MODES)}")
        if self.temperature is not None and self.temperature < 0:
            raise ValueError("temperature must be >= 0")


def iteration_seed(seed: int, iteration: int) -> int:
    return (int(seed) * 0x9E3779B1 + (int(iteration) + 1) * 0x85EBCA6B) & 0xFFFFFFFF


def draws(seed: int, iteration: int, B: int, N: int):
    """Python twin of the kernels' draws of one iteration: (u_categorical [B*N], u_gumbel [B*N], u_transfer [B, N]) fp32."""
    s = iteration_seed(seed, iteration)
    inv = np.float32(1.0 / 16777216.0)
    rows = np.arange(B * N)
    u_cat = _rng24(s, 32, rows, 0).astype(np.float32) * inv
    u_gum = _rng24(s, 33, rows, 0).astype(np.float32) * inv
    u_tr = (_rng24(s, 34, np.arange(B)[:, None], np.arange(N)[None, :]).astype(np.float32) * inv)
    T = torch.from_numpy
    return T(u_cat), T(u_gum), T(u_tr)


def token_sample(engine, rows: int, cfg: GenerationConfig, seed: int, gumbel: bool):
    """Candidates and confidences of the first `rows` rows of the head logits of the last forward (in place on the
    workspace): (conf f32 [rows], token i64 [rows])."""
    p, ld = C.c_void_p(), C.c_int32()
    L.check(engine.lib.gget_head_logits(engine.h, C.byref(p), C.byref(ld)))
    conf = torch.empty(rows, dtype=torch.float32, device=engine.device)
    tok = torch.empty(rows, dtype=torch.int64, device=engine.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    top_p = float(cfg.top_p) if (cfg.top_p is not None and cfg.top_p < 1) else 0.0
    top_k = int(cfg.top_k) if cfg.top_k is not None else 0
    alg_temp = float(cfg.alg_temp) if (gumbel and cfg.alg_temp is not None and cfg.alg_temp > 0) else 0.0
    L.check(engine.lib.gget_op_token_sample(p, ld.value, int(rows), engine.spec.vocab_size, _MODES[cfg.alg],
                                            float(cfg.temperature or 0.0), top_p, top_k, alg_temp, int(seed) & 0xFFFFFFFF,
                                            C.c_void_p(conf.data_ptr()), C.c_void_p(tok.data_ptr()), st))
    return conf, tok


def reveal_counts(x: torch.Tensor, timesteps: torch.Tensor, i: int, mask_token_id: int):
    """Step bookkeeping of the confidence-ranked algorithms (reference :166-178): advance i until some sample may reveal at
    least one token; returns (n_reveal int32 [B], k, i)."""
    steps = len(timesteps) - 1
    n_masked = (x == mask_token_id).sum(dim=1)
    k = 0
    n_reveal = torch.zeros_like(n_masked, dtype=torch.int32)
    if int(n_masked.sum().item()) > 0:
        while k == 0 and i < steps:
            t, s = timesteps[i], timesteps[i + 1]
            p = 1 - s / t if i < steps - 1 else 1.0
            n_reveal = torch.floor(n_masked * p).int()
            k = int(n_reveal.max().item())
            i += 1
    return n_reveal, k, i


def unmask_step(x: torch.Tensor, conf: torch.Tensor, cand: torch.Tensor, timesteps: torch.Tensor, i: int,
                cfg: GenerationConfig) -> Tuple[torch.Tensor, int]:
    """One confidence-ranked update (reference :163-236): x [B, N] tokens, conf / cand [B, N] from the sampling kernel.
    Every sample reveals its floor(n_masked * p) most confident masked positions (see the module docstring for the one
    difference from the reference's fixed-k scatter)."""
    n_reveal, k, i = reveal_counts(x, timesteps, i, cfg.mask_token_id)
    if k == 0:
        return x, i
    masked = x == cfg.mask_token_id
    conf = conf.masked_fill(~masked, float("-inf"))
    order = torch.sort(conf, dim=1, descending=True, stable=True).indices[:, :k]     # [B, k] most confident first
    new = torch.gather(cand, 1, order)
    keep = torch.arange(k, device=x.device)[None, :] < n_reveal[:, None]
    old = torch.gather(x, 1, order)
    x = x.scatter(1, order, torch.where(keep, new, old))
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
    i = it = 0
    while i < n_steps:
        model(input_ids=x.view(B, S, F), attention_mask=att, labels=None)      # logits for all B*S*F rows, in the workspace
        seed = iteration_seed(cfg.seed, it)
        conf, cand = token_sample(model._engine, B * S * F, cfg, seed, gumbel=cfg.alg != "origin")
        if cfg.alg == "origin":
            t, s = timesteps[i], timesteps[i + 1]
            p_transfer = float(1 - s / t) if i < n_steps - 1 else 1.0
            L.check(model._engine.lib.gget_op_unmask_origin(C.c_void_p(x.data_ptr()), C.c_void_p(cand.data_ptr()), B, S * F,
                                                            p_transfer, seed, int(cfg.mask_token_id),
                                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            i += 1
        else:
            x, i = unmask_step(x, conf.view(B, S * F), cand.view(B, S * F), timesteps, i, cfg)
        it += 1
        if history is not None:
            history.append(x.view(B, S, F).clone())
    return x, history
