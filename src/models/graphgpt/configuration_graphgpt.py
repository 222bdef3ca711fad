"""`src.models.graphgpt.configuration_graphgpt` of the reference (configuration_graphgpt.py:6-342)."""
import importlib as _il

_m = _il.import_module("graph-gpt_amd.modeling")
GraphGPTConfig = _m.GraphGPTConfig
convert_to_legacy_config = _m.convert_to_legacy_config

__all__ = ["GraphGPTConfig", "convert_to_legacy_config"]
