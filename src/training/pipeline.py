"""`src.training.pipeline` of the reference (src/training/pipeline.py:15-257): `TrainingPipeline(cfg, mode).run()` and `launch`."""
import importlib as _il

_t = _il.import_module("graph-gpt_amd.training")
TrainingPipeline = _t.TrainingPipeline
launch = _t.launch

__all__ = ["TrainingPipeline", "launch"]
