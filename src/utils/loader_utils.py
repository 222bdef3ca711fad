"""`src.utils.loader_utils` of the reference: checkpoint loading (loader_utils.py:165-220) and the rank samplers (:70-90, :328-333)."""
import importlib as _il

_c = _il.import_module("graph-gpt_amd.checkpoint")
_t = _il.import_module("graph-gpt_amd.training")
load_from_ckp = _c.load_from_ckp
load_from_ckp_with_try = _c.load_from_ckp_with_try
distribute_sampler = _t.eval_rank_sampler
distribute_sampler_with_rnd_seed = _t.finetune_rank_sampler

__all__ = ["load_from_ckp", "load_from_ckp_with_try", "distribute_sampler", "distribute_sampler_with_rnd_seed"]
