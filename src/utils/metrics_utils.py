"""`src.utils.metrics_utils` of the reference (metrics_utils.py:38-189): the fine-tune metric objects."""
import importlib as _il

_m = _il.import_module("graph-gpt_amd.metrics")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("_")})
