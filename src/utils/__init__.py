"""`src.utils` of the reference, restricted to what touches the hot path: the per-step functions (training_utils.py), checkpoint
interchange (loader_utils.py:165-220, misc_utils.py:33-49,105-121), the metrics of SURVEY.md row A14 and the evaluation passes
(log_eval_dump_utils.py:77-163, :242-304)."""
from . import loader_utils, misc_utils, metrics_utils, log_eval_dump_utils, training_utils

__all__ = ["loader_utils", "misc_utils", "metrics_utils", "log_eval_dump_utils", "training_utils"]
