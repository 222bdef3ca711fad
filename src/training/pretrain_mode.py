"""`src.training.pretrain_mode` of the reference (src/training/pretrain_mode.py:48): imported by examples/train_pretrain.py:6."""
import importlib as _il

PretrainMode = _il.import_module("graph-gpt_amd.training").PretrainMode

__all__ = ["PretrainMode"]
