"""`src.utils.misc_utils` of the reference: checkpoint naming / saving (misc_utils.py:33-49, :105-121), the variable-length
all-gather (:472-504) and the distributed environment (:507-539)."""
import importlib as _il

_c = _il.import_module("graph-gpt_amd.checkpoint")
_t = _il.import_module("graph-gpt_amd.training")
get_latest_ckp = _c.get_latest_ckp
MODEL_NAME = _c.MODEL_NAME
save_model = _c.save_model
all_gather = _t.all_gather_varlen
set_dist_env = _t.set_dist_env

__all__ = ["get_latest_ckp", "MODEL_NAME", "save_model", "all_gather", "set_dist_env"]
