"""The slice of the reference's structured-config tree that the hot path reads (reference src/conf/base_configs.py:28-203).

The reference drives `TrainingPipeline(cfg, mode)` with a Hydra/OmegaConf `Config` whose sub-trees are `tokenization`, `model`,
`training`, `generation`.  Hydra, the tokenization tree and the generation tree stay on the host; what the
training path consumes is

    cfg.model                         -> convert_to_legacy_config            (configuration_graphgpt.py:210-342)
    cfg.training.batch_size / .deepspeed_conf_file / .pretrain_cpt / .output_dir / .task_type
    cfg.training.schedule             -> total_num_steps / warmup_num_steps  (base_configs.py:52-66, 166-176)
    cfg.training.optimizer            -> AdamW hyper-parameters + clip       (conf_utils.py:49-103, opt_utils.py:7-36)
    cfg.training.distributed          <- world_size / rank                   (pipeline.py:135-139)

The dataclasses below carry exactly those fields with the reference's names and defaults, so that a maintainer's config object -
the reference's own dataclasses, an OmegaConf node, a SimpleNamespace tree or these - is read the same way (attribute access only).
The functions restate the reference's schedule arithmetic; each cites the lines it follows.
"""
from __future__ import annotations

import dataclasses
import json
import math
import os
from typing import Any, List, Optional


@dataclasses.dataclass
class DistConfig:                       # base_configs.py:28-31
    world_size: int = 1
    rank: int = 0


@dataclasses.dataclass
class ScheduleConfig:                   # base_configs.py:34-49
    epochs: Optional[int] = None
    warmup_epochs: Optional[float] = None
    total_tokens: float = 1e9
    warmup_tokens: float = 1e8
    total_num_steps: Optional[int] = None       # MISSING in the reference: filled by update_num_steps / update_ft_num_steps
    warmup_num_steps: Optional[int] = None
    logging_steps: int = 100
    samples_per_saving: Optional[int] = None
    steps_per_saving: Optional[int] = None
    samples_per_eval: Optional[int] = None


@dataclasses.dataclass
class OptimizerConfig:                  # base_configs.py:75-87
    lr: float = 0.001
    min_lr: float = 0.0
    betas: List[float] = dataclasses.field(default_factory=lambda: [0.9, 0.95])
    weight_decay: float = 0.1
    eps: float = 1e-6
    max_grad_norm: float = 1.0
    gradient_accumulation_steps: int = 1
    use_ema: bool = False
    ema_decay: float = 0.9999


@dataclasses.dataclass
class FinetuneTrainConfig:              # base_configs.py:106-115
    freeze: int = -1
    seed: int = -1
    use_aux: bool = False
    aux_ratio: float = 0.0
    task_ratio: float = 1.0


@dataclasses.dataclass
class TrainingConfig:                   # base_configs.py:131-163 (the fields the training path reads)
    deepspeed_conf_file: str = ""
    use_deepspeed: bool = False
    pretrain_cpt: str = ""
    task_type: str = "pretrain"
    output_dir: str = "../exp/models/graph_llama_test"
    batch_size: int = 128
    batch_size_eval: Optional[int] = None
    pack_tokens: float = 0
    focal_gamma: float = 0
    distributed: DistConfig = dataclasses.field(default_factory=DistConfig)
    schedule: ScheduleConfig = dataclasses.field(default_factory=ScheduleConfig)
    optimizer: OptimizerConfig = dataclasses.field(default_factory=OptimizerConfig)
    finetune: FinetuneTrainConfig = dataclasses.field(default_factory=FinetuneTrainConfig)


@dataclasses.dataclass
class Config:
    """`src.conf.Config`.  Reference-shaped use: `tokenization`, `model` (the nested GraphGPTModelConfig), `training`, `generation`
    (base_configs.py:187-203).  The lean form of earlier rounds - `model` = GraphGPTConfig or its keyword dict, `optim`, `batches`,
    `max_steps`, `log_every`, `output_dir`, `resume_from` - keeps working: `TrainingPipeline` tells the two apart by `training`."""
    tokenization: Any = None
    model: Any = None
    training: Any = None
    generation: Any = None
    # lean form
    optim: Any = None
    batches: Any = None
    max_steps: int = 0
    log_every: int = 0
    output_dir: Optional[str] = None
    resume_from: Optional[str] = None


def is_reference_config(cfg) -> bool:
    tr = cfg.get("training") if isinstance(cfg, dict) else getattr(cfg, "training", None)
    return tr is not None and _get(tr, "optimizer", None) is not None and _get(tr, "schedule", None) is not None


def _get(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _set(obj, name, value):
    if isinstance(obj, dict):
        obj[name] = value
    else:
        setattr(obj, name, value)


def update_num_steps(sched_cfg, tokens_per_sample, batch_size, world_size) -> None:
    """base_configs.py:52-58 - optimizer steps of a token-budgeted run, rounded UP; the global batch is world * batch_size."""
    per_step = tokens_per_sample * batch_size * world_size
    _set(sched_cfg, "total_num_steps", int(math.ceil(_get(sched_cfg, "total_tokens") / per_step)))
    _set(sched_cfg, "warmup_num_steps", int(math.ceil(_get(sched_cfg, "warmup_tokens") / per_step)))


def update_epochs(sched_cfg, tokens_per_sample, samples_per_gpu, world_size) -> None:
    """base_configs.py:61-64."""
    _set(sched_cfg, "epochs", int(math.ceil(_get(sched_cfg, "total_tokens") / (tokens_per_sample * samples_per_gpu * world_size))))


def update_ft_num_steps(train_cfg, samples_per_gpu) -> None:
    """base_configs.py:166-176 - fine-tune schedules are epoch-budgeted; steps per epoch round DOWN."""
    sched = _get(train_cfg, "schedule")
    per_epoch = samples_per_gpu // _get(train_cfg, "batch_size")
    _set(sched, "total_num_steps", _get(sched, "epochs") * per_epoch)
    _set(sched, "warmup_num_steps", int(_get(sched, "warmup_epochs") * per_epoch))


def set_finetune_cfg(ft_cfg) -> None:
    """base_configs.py:179-184."""
    _set(ft_cfg, "aux_ratio", 1 - _get(ft_cfg, "task_ratio"))
    _set(ft_cfg, "use_aux", _get(ft_cfg, "aux_ratio") > 0)


# DeepSpeed scheduler types the reference patches in place (loss_utils.py:19, :170-215); any other `scheduler.type` in the JSON is a
# torch scheduler built by loss_utils.set_py_scheduler (the fine-tune JSONs name OneCycleLR, examples/ds_config2.json:20-23)
_DS_SCHEDULERS = ("WarmupLR", "WarmupDecayLR", "OneCycle", "LRRangeTest")


def ds_scheduler_type(train_cfg, default: str) -> str:
    """`scheduler.type` of the DeepSpeed JSON `training.deepspeed_conf_file` points at (conf_utils.py:57-58, :74-77).  The reference
    cannot start without the file; here a missing file (the JSONs live under the reference's examples/) falls back to the type the
    reference's own JSON of that stage names: WarmupDecayLR for pre-training (ds_config2_pt.json:20-28), OneCycleLR for fine-tuning."""
    path = _get(train_cfg, "deepspeed_conf_file", "") or ""
    if path and os.path.isfile(path):
        with open(path) as fh:
            ds = json.load(fh)
        return ds.get("scheduler", {}).get("type", default)
    return default


def optim_from_training(train_cfg, use_deepspeed: bool, finetune: bool):
    """`training.optimizer` + `training.schedule` -> OptimConfig (what `deepspeed.initialize(config=ds_config)` /
    `initialize_optimizer` set up in the reference).

    * AdamW: lr, betas, eps, weight_decay and the clip threshold straight from `training.optimizer`
      (conf_utils.py:65-72 patches them into the JSON's Adam block; opt_utils.py:18-24 hands them to torch.optim.AdamW).
    * DeepSpeed + WarmupDecayLR (pre-training): the reference passes `min_lr = optim_cfg.lr` as `warmup_min_lr` (conf_utils.py:53,
      :82-83) - with DeepSpeed's lr = min + (max - min) * gamma the schedule degenerates to the CONSTANT lr (SURVEY.md row A12's quirk,
      reproduced on purpose: min_lr = lr).  `training.optimizer.min_lr` (0.1 * lr, pretrain_mode.py:108) is not read on this path.
    * DeepSpeed + a torch scheduler (fine-tuning, OneCycleLR): total_steps = total_num_steps, pct_start = warmup / total,
      min_lr = optimizer.min_lr (conf_utils.py:106-131).
    * DDP (no DeepSpeed JSON): OneCycleLR over total_num_steps + 1 steps, min_lr = optimizer.min_lr (opt_utils.py:25-33)."""
    from .training import OptimConfig
    oc, sc = _get(train_cfg, "optimizer"), _get(train_cfg, "schedule")
    total, warm = _get(sc, "total_num_steps"), _get(sc, "warmup_num_steps")
    if total is None or warm is None or isinstance(total, str) or isinstance(warm, str):
        raise ValueError("training.schedule.total_num_steps / warmup_num_steps are not set: the mode's prepare_data fills them from the "
                         "token budget (update_num_steps) or the epoch budget (update_ft_num_steps)")
    lr = float(_get(oc, "lr"))
    kw = dict(lr=lr, betas=tuple(_get(oc, "betas")), eps=float(_get(oc, "eps")), weight_decay=float(_get(oc, "weight_decay")),
              max_grad_norm=float(_get(oc, "max_grad_norm")), warmup_num_steps=int(warm), total_num_steps=int(total))
    if use_deepspeed:
        # conf_utils.py:59-66: the DS engine gets `gradient_accumulation_steps` (its step() fires at the boundary only)
        kw["gradient_accumulation_steps"] = int(_get(oc, "gradient_accumulation_steps", 1) or 1)
        kind = ds_scheduler_type(train_cfg, "OneCycleLR" if finetune else "WarmupDecayLR")
        if kind == "WarmupDecayLR":
            return OptimConfig(schedule="warmup_decay", min_lr=lr, **kw)
        if kind == "OneCycleLR":
            return OptimConfig(schedule="onecycle", min_lr=float(_get(oc, "min_lr") or 0.0), onecycle_extra_step=0, **kw)
        raise NotImplementedError(f"DeepSpeed JSON scheduler.type = {kind!r}: the reference's launch scripts use WarmupDecayLR "
                                  "(pre-training) and OneCycleLR (fine-tuning) only")
    return OptimConfig(schedule="onecycle", min_lr=float(_get(oc, "min_lr") or 0.0), onecycle_extra_step=1, **kw)
