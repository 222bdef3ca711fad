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
  * `zero_to_fp32_state_dict`   - the reference's fallback for a directory that holds DeepSpeed ZeRO shards
    (loader_utils.py:199-207 calls `deepspeed.utils.zero_to_fp32.get_fp32_state_dict_from_zero_checkpoint`; the reference's
    DeepSpeed pre-training - ds_config2_pt.json: bf16 + ZeRO stage 2 - writes exactly such directories).  DeepSpeed is a
    third-party dependency (requirements.txt:7 `deepspeed==0.15.4`) that is not installed here: the consolidation is RESTATED from
    that module's published algorithm for ZeRO stages 1 / 2 and is UNPINNED - no shard written by DeepSpeed itself was available to
    check it against; tests/test_checkpoint.py round-trips shards written in the documented layout.  Stage 3 raises.
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
            names = os.listdir(ckp) if os.path.isdir(ckp) else []
            if "latest" in names or any(n.startswith("global_step") for n in names) or any(n.endswith("_optim_states.pt") for n in names):
                # the reference's `except` branch (loader_utils.py:199-207): DeepSpeed ZeRO shards -> consolidated fp32 state dict
                sd = zero_to_fp32_state_dict(ckp)
                return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
            raise FileNotFoundError(f"no {', '.join(_CANDIDATES)} under {ckp}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and all(not torch.is_tensor(v) for v in sd.values()):
            sd = sd["state_dict"]
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def _natural_key(name: str):
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def zero_to_fp32_state_dict(checkpoint_dir: str, tag: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """fp32 state dict out of a DeepSpeed ZeRO stage-1 / 2 checkpoint directory - what
    `deepspeed.utils.zero_to_fp32.get_fp32_state_dict_from_zero_checkpoint(checkpoint_dir, tag)` returns (deepspeed 0.15.4, restated;
    the reference calls it at loader_utils.py:199-207).  Layout read:

        <dir>/latest                                        text file holding the tag ("global_step<N>") when `tag` is None;
                                                            a directory that IS the tag directory is accepted as well
        <dir>/<tag>/*_model_states.pt                       rank 0's file: "module" (the low-precision weights, for buffers),
                                                            "buffer_names", "param_shapes" (one ordered name -> shape map per optimizer
                                                            group), "shared_params", optional "frozen_param_shapes" / "frozen_param_fragments"
        <dir>/<tag>/*zero_pp_rank_<r>_mp_rank_00_optim_states.pt   one per data-parallel rank: "optimizer_state_dict" with "zero_stage",
                                                            "partition_count" and "single_partition_of_fp32_groups" (that rank's slice of
                                                            every group's flat fp32 master vector)

    Per optimizer group the ranks' slices are concatenated in rank order and cut into the group's parameters in `param_shapes` order; the
    tail of a group is padding up to a multiple of 2 x world size (checked like the original: consumed and available element counts must
    agree after rounding both up to that multiple).  Buffers come from "module" (cast to fp32), frozen parameters from their fragments,
    tied parameters are aliased last."""
    import glob
    import math
    if tag is None:
        latest = os.path.join(checkpoint_dir, "latest")
        if os.path.isfile(latest):
            with open(latest) as fh:
                tag = fh.read().strip()
        elif glob.glob(os.path.join(checkpoint_dir, "*_optim_states.pt")):
            tag = ""
        else:
            raise ValueError(f"Unable to find 'latest' file at {latest}")
    ds_dir = os.path.join(checkpoint_dir, tag) if tag else checkpoint_dir
    if not os.path.isdir(ds_dir):
        raise FileNotFoundError(f"Directory '{ds_dir}' doesn't exist")
    optim_files = sorted(glob.glob(os.path.join(ds_dir, "*_optim_states.pt")), key=lambda f: _natural_key(os.path.basename(f)))
    model_files = sorted(glob.glob(os.path.join(ds_dir, "*_model_states.pt")), key=lambda f: _natural_key(os.path.basename(f)))
    if not optim_files or not model_files:
        raise FileNotFoundError(f"can't find *_optim_states.pt / *_model_states.pt files in directory '{ds_dir}'")
    osd = [torch.load(f, map_location="cpu", weights_only=False)["optimizer_state_dict"] for f in optim_files]
    if "zero_stage" not in osd[0]:
        raise ValueError(f"{optim_files[0]} is not a zero checkpoint")
    stage = int(osd[0]["zero_stage"])
    world = osd[0]["partition_count"]
    world = max(world) if isinstance(world, (list, tuple)) else int(world)
    if world != len(optim_files):
        raise ValueError(f"Expected {world} of '*_optim_states.pt' under '{ds_dir}' but found {len(optim_files)} files. "
                         "Possibly due to an overwrite of an old checkpoint, or a checkpoint didn't get saved by one or more processes.")
    if stage > 2:
        raise NotImplementedError(f"ZeRO stage {stage} shards: the reference's configurations use stage 2 (examples/ds_config2*.json)")
    flat = [d["single_partition_of_fp32_groups"] for d in osd]              # [rank][group] -> flat fp32 slice
    ms = torch.load(model_files[0], map_location="cpu", weights_only=False)
    if "buffer_names" not in ms:
        raise ValueError(f"{model_files[0]} is not a model state checkpoint")
    param_shapes = ms["param_shapes"]
    if isinstance(param_shapes, dict):     # (older DeepSpeed: one map, not a list of per-group maps)
        param_shapes = [param_shapes]
    out: Dict[str, torch.Tensor] = {}
    for k in ms.get("buffer_names", []):
        if k in ms["module"]:
            out[k] = ms["module"][k].float()
    for name, frag in (ms.get("frozen_param_fragments") or {}).items():
        out[name] = frag.float() if torch.is_tensor(frag) else frag
    if len(param_shapes) != len(flat[0]):
        raise ValueError(f"{len(param_shapes)} parameter groups in the model states, {len(flat[0])} in the optimizer states")
    align = 2 * world
    up = lambda x: align * math.ceil(x / align)
    for gi, shapes in enumerate(param_shapes):
        full = torch.cat([flat[r][gi].reshape(-1).float() for r in range(world)], 0)
        off = 0
        for name, shape in shapes.items():
            n = int(math.prod(tuple(shape))) if len(tuple(shape)) else 1
            out[name] = full.narrow(0, off, n).view(tuple(shape)).clone()
            off += n
        if up(off) != up(full.numel()):
            raise ValueError(f"consumed {up(off)} numels out of {up(full.numel())} - something is wrong")
    for pair in ms.get("shared_params", []) or []:
        if pair[1] in out:
            out[pair[0]] = out[pair[1]]
    return out


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
