"""`src.models.graphgpt.modeling_common` of the reference: the output container (modeling_common.py:55-99)."""
import importlib as _il

DoubleHeadsModelOutput = _il.import_module("graph-gpt_amd.modeling").DoubleHeadsModelOutput

__all__ = ["DoubleHeadsModelOutput"]
