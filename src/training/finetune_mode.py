"""`src.training.finetune_mode` of the reference (src/training/finetune_mode.py:43): imported by examples/train_supervised.py:6."""
import importlib as _il

FinetuneMode = _il.import_module("graph-gpt_amd.training").FinetuneMode

__all__ = ["FinetuneMode"]
