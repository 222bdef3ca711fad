"""`src.conf` of the reference, reduced to the names the entry scripts import (examples/train_pretrain.py:7, train_supervised.py:7).
The Hydra structured-config tree itself (src/conf/base_configs.py and sub-packages) is host-side configuration plumbing outside
the hot path: `Config` here is the plain container the lean `TrainingPipeline` reads - `model`
(GraphGPTConfig or its keyword dict), `optim`, `batches` (an iterable of collated batches), `max_steps`, `log_every`, `output_dir`,
`resume_from` - and `TrainingStats` the three fields the reference's step functions read from theirs."""
import dataclasses
from typing import Any, Iterable, Optional


@dataclasses.dataclass
class Config:
    model: Any = None
    optim: Any = None
    batches: Optional[Iterable] = None
    max_steps: int = 0
    log_every: int = 0
    output_dir: Optional[str] = None
    resume_from: Optional[str] = None


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


__all__ = ["Config", "TrainingStats"]
