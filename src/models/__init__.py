"""`src.models` of the reference (src/models/__init__.py:1-20) for the hot-path classes."""
from .graphgpt.modeling_graphgpt import GraphGPTPretrainBase, GraphGPTTaskModel, DoubleHeadsModelOutput
from .graphgpt.configuration_graphgpt import GraphGPTConfig, convert_to_legacy_config

__all__ = ["convert_to_legacy_config", "GraphGPTTaskModel", "GraphGPTPretrainBase", "GraphGPTConfig",
           "DoubleHeadsModelOutput"]
