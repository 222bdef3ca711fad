"""`src.models.graphgpt.modeling_pretrain` of the reference: GraphGPTPretrainBase (modeling_pretrain.py:152-266)."""
import importlib as _il

GraphGPTPretrainBase = _il.import_module("graph-gpt_amd.modeling").GraphGPTPretrainBase

__all__ = ["GraphGPTPretrainBase"]
