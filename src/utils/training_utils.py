"""`src.utils.training_utils` of the reference (training_utils.py:7-205): the per-step functions, in the reference's positional
form `batch_training(data, model, train_cfg, train_stats, opt_stats)` / `ft_batch_training(data, model, fthead_cfg, train_cfg,
train_stats, opt_stats)` as well as the short `(data, engine)` form."""
import importlib as _il

_t = _il.import_module("graph-gpt_amd.training")
batch_training = _t.batch_training
ft_batch_training = _t.ft_batch_training

__all__ = ["batch_training", "ft_batch_training"]
