"""`src.models` of the reference (src/models/__init__.py:1-20) for the hot-path classes."""
import importlib as _il

_m = _il.import_module("graph-gpt_amd.modeling")
GraphGPTPretrainBase = _m.GraphGPTPretrainBase
GraphGPTTaskModel = _m.GraphGPTTaskModel
GraphGPTConfig = _m.GraphGPTConfig
DoubleHeadsModelOutput = _m.DoubleHeadsModelOutput
convert_to_legacy_config = _m.convert_to_legacy_config

__all__ = ["convert_to_legacy_config", "GraphGPTTaskModel", "GraphGPTPretrainBase", "GraphGPTConfig",
           "DoubleHeadsModelOutput"]
