"""`src.training` of the reference (src/training/__init__.py) for the hot path."""
import importlib as _il

_t = _il.import_module("graph-gpt_amd.training")
TrainingPipeline = _t.TrainingPipeline
TrainingMode = _t.TrainingMode
PretrainMode = _t.PretrainMode
FinetuneMode = _t.FinetuneMode
launch = _t.launch

__all__ = ["TrainingPipeline", "TrainingMode", "launch", "PretrainMode", "FinetuneMode"]
