"""Checkpoint interchange with the reference.

The parameter names of this package ARE the reference state-dict keys (spec.py), so a reference checkpoint loads by
name.  This module restates the reference's loading rules:
  * `get_latest_ckp`            - reference src/utils/misc_utils.py:33-49 (`epoch_<n>` sub-directories, newest or oldest)
  * `load_from_ckp_with_try`    - reference src/utils/loader_utils.py:176-220: `<ckp>/model.pt` written by
    `torch.save(model.state_dict())` under DDP (keys prefixed `module.`, misc_utils.py:105-121); the prefix is stripped,
    keys containing "score" are dropped when `skip_keys` (a pre-trained trunk under a fresh fine-tune head), the rest is
    loaded with `strict=False`, and the missing / unexpected keys are reported.
  * `load_from_ckp`             - reference loader_utils.py:165-174 (only when `pretrain_cpt` differs from `output_dir`).
Also read: `model.safetensors` / `pytorch_model.bin` (HF `save_pretrained` layout of the published checkpoints).
DeepSpeed ZeRO shards (the reference's fallback `get_fp32_state_dict_from_zero_checkpoint`) need DeepSpeed's own
consolidation; a directory that only holds ZeRO shards raises with that message instead of guessing.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

MODEL_NAME = "model.pt"          # reference misc_utils.py:25
_CANDIDATES = (MODEL_NAME, "model.safetensors", "pytorch_model.bin")


def _scan_ckps(pretrain_cpt: str):
    names = [f.name for f in os.scandir(pretrain_cpt) if f.is_dir()]
    tails = [n.split("_")[-1] for n in names]
    return sorted(int(t) for t in tails if t.isnumeric())


def get_latest_ckp(pretrain_cpt: str, eval_only: int = 0) -> Tuple[str, Optional[int]]:
    eps = _scan_ckps(pretrain_cpt)
    if eps:
        ep = min(eps) if eval_only else max(eps)
        return os.path.join(pretrain_cpt, f"epoch_{ep}"), ep
    return pretrain_cpt, None


def read_state_dict(ckp: str, use_ema: bool = False) -> Dict[str, torch.Tensor]:
    """State dict of a checkpoint directory (or file), DDP prefix stripped, tensors on the CPU."""
    if os.path.isfile(ckp):
        path = ckp
    elif use_ema:
        path = os.path.join(ckp, "../model_ema_best.pt")
    else:
        path = next((os.path.join(ckp, n) for n in _CANDIDATES if os.path.isfile(os.path.join(ckp, n))), None)
        if path is None:
            zero = [n for n in os.listdir(ckp)] if os.path.isdir(ckp) else []
            hint = " (DeepSpeed ZeRO shards: consolidate them with DeepSpeed's zero_to_fp32 first)" \
                if any(n.startswith("global_step") or n == "latest" for n in zero) else ""
            raise FileNotFoundError(f"no {', '.join(_CANDIDATES)} under {ckp}{hint}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and all(not torch.is_tensor(v) for v in sd.values()):
            sd = sd["state_dict"]
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_from_ckp_with_try(model, ckp: str, skip_keys: bool = True, strict: bool = False, use_ema: bool = False):
    print(f"[gget] reading checkpoint {ckp}")
    sd = read_state_dict(ckp, use_ema=use_ema)
    dropped = [key for key in sd if skip_keys and "score" in key]   # a pre-trained trunk under a fresh fine-tune head
    for key in dropped:
        del sd[key]
    if dropped:
        print(f"[gget] head tensors left at their fresh initialisation: {dropped}")
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    print(f"[gget] {len(sd)} tensors loaded by name; not in the checkpoint: {list(missing)}; not in the model: {list(unexpected)}")
    model.last_load_result = (list(missing), list(unexpected))
    return model


def load_from_ckp(pretrain_cpt: str, output_dir: str, model, config=None, skip_keys: bool = True, strict: bool = False):
    if len(pretrain_cpt) > 0 and pretrain_cpt != output_dir:
        ckp, _ = get_latest_ckp(pretrain_cpt)
        model = load_from_ckp_with_try(model, ckp, skip_keys, strict)
    return model


def save_model(model, output_dir: str, ddp_prefix: bool = False):
    """`model.pt` in the reference's DDP layout (misc_utils.py:105-121) + config.json; readable by the reference."""
    os.makedirs(output_dir, exist_ok=True)
    sd = {("module." + k if ddp_prefix else k): v.detach().float().cpu() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(output_dir, MODEL_NAME))
    if hasattr(model, "config") and hasattr(model.config, "save_pretrained"):
        model.config.save_pretrained(output_dir)
    return os.path.join(output_dir, MODEL_NAME)
