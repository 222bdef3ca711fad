"""`src.utils.log_eval_dump_utils` of the reference: the evaluation passes (log_eval_dump_utils.py:77-163, :242-304)."""
import importlib as _il

_t = _il.import_module("graph-gpt_amd.training")
evaluate = _t.evaluate
ft_evaluate = _t.ft_evaluate

__all__ = ["evaluate", "ft_evaluate"]
