"""`src.models.graphgpt.modeling_finetune` of the reference: GraphGPTTaskModel (modeling_finetune.py:236-326)."""
import importlib as _il

GraphGPTTaskModel = _il.import_module("graph-gpt_amd.modeling").GraphGPTTaskModel

__all__ = ["GraphGPTTaskModel"]
