"""`src.training.mode` of the reference (src/training/mode.py:5-89): the strategy interface."""
import importlib as _il

TrainingMode = _il.import_module("graph-gpt_amd.training").TrainingMode

__all__ = ["TrainingMode"]
