"""`src.conf` of the reference, reduced to the names the training path reads (examples/train_pretrain.py:7, train_supervised.py:7,
src/conf/base_configs.py:28-203).  Hydra and the tokenization / generation trees are host-side plumbing outside the hot path
(SURVEY.md section 2); `Config` carries the reference's four sub-trees - `TrainingPipeline(cfg, mode)` reads `cfg.model` (nested
GraphGPTModelConfig) and `cfg.training` exactly as the reference's does - and still accepts the lean form of earlier rounds
(`model`, `optim`, `batches`, ...).  `TrainingStats` = the fields the reference's step functions read from theirs."""
import dataclasses
import importlib as _il
from typing import Any

_c = _il.import_module("graph-gpt_amd.conf")
Config, TrainingConfig, ScheduleConfig, OptimizerConfig = _c.Config, _c.TrainingConfig, _c.ScheduleConfig, _c.OptimizerConfig
DistConfig, FinetuneTrainConfig = _c.DistConfig, _c.FinetuneTrainConfig


@dataclasses.dataclass
class TrainingStats:
    """The fields of the reference's TrainingStats (src/conf/stats_configs.py) that batch_training / ft_batch_training touch."""
    device: Any = None
    has_embeds_input: bool = False
    use_deepspeed: bool = True
    i: int = 0
    loss: Any = None
    main_loss: Any = None
    aux_loss: Any = None
    inputs_shape: Any = None
    sliced_raw_embeds: Any = None


__all__ = ["Config", "TrainingConfig", "ScheduleConfig", "OptimizerConfig", "DistConfig", "FinetuneTrainConfig", "TrainingStats"]
