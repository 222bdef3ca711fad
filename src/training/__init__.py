"""`src.training` of the reference (src/training/__init__.py) for the hot path."""
from .pipeline import TrainingPipeline, launch
from .mode import TrainingMode
from .pretrain_mode import PretrainMode
from .finetune_mode import FinetuneMode

__all__ = ["TrainingPipeline", "TrainingMode", "launch", "PretrainMode", "FinetuneMode"]
