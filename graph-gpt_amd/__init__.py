"""gget-mi355x: MI355X-native engine for the GraphGPT Graph-Eulerian-Transformer hot path.

Import with ``importlib.import_module("graph-gpt_amd")`` (the directory name is fixed by the build
contract and is not a Python identifier) or through the drop-in surface ``src.models`` /
``src.training`` at the repo root.
"""
from .spec import ModelSpec, spec_from_size, KIND_PRETRAIN, KIND_TASK, MODEL_SIZES  # noqa: F401
