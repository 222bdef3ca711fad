"""`src.models.graphgpt.modeling_graphgpt` of the reference (modeling_graphgpt.py:26-29 re-exports the four modeling_* modules); the
model classes outside the hot path (GraphGPTPosPred, the double-heads / denoising fine-tune models - SURVEY.md section 2 out of
scope) are not provided."""
from .modeling_common import DoubleHeadsModelOutput
from .modeling_pretrain import GraphGPTPretrainBase
from .modeling_finetune import GraphGPTTaskModel

__all__ = ["DoubleHeadsModelOutput", "GraphGPTPretrainBase", "GraphGPTTaskModel"]
