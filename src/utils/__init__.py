"""`src.utils` of the reference, restricted to what touches the hot path: checkpoint interchange
(reference src/utils/loader_utils.py:165-220, misc_utils.py:33-49,105-121) the metrics of SURVEY.md row A14 and the pre-train evaluation pass (log_eval_dump_utils.py:242-304)."""
import importlib as _il
import types as _types

_c = _il.import_module("graph-gpt_amd.checkpoint")
_m = _il.import_module("graph-gpt_amd.metrics")

loader_utils = _types.SimpleNamespace(load_from_ckp=_c.load_from_ckp, load_from_ckp_with_try=_c.load_from_ckp_with_try)
misc_utils = _types.SimpleNamespace(get_latest_ckp=_c.get_latest_ckp, MODEL_NAME=_c.MODEL_NAME, save_model=_c.save_model)
metrics_utils = _m
_t = _il.import_module("graph-gpt_amd.training")
log_eval_dump_utils = _types.SimpleNamespace(evaluate=_t.evaluate, ft_evaluate=_t.ft_evaluate)

__all__ = ["loader_utils", "misc_utils", "metrics_utils", "log_eval_dump_utils"]
