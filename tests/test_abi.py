"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/gget.h declares; host-only entry points behave; the product path refuses to run without a GPU."""
import ctypes as C
import importlib
import os
import re

import pytest
import torch

from _util import ROOT, spec_mod

L = importlib.import_module("graph-gpt_amd._lib")


@pytest.fixture(scope="module")
def lib():
    b = importlib.import_module("graph-gpt_amd.build")
    b.build()
    return L.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gget.h")).read()
    declared = set(re.findall(r"\b(gget_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gget_config_t", "gget_sizes_t"}
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in gget.h but not exported: {missing}"
    assert declared == set(L.SIGNATURES), f"ctypes table out of sync: {declared ^ set(L.SIGNATURES)}"


def test_query_sizes_and_param_layout(lib):
    spec = spec_mod.spec_from_size("base", vocab_size=756, stacked_feat=13, next_n_token=13)
    cfg = L.GgetConfig()
    (cfg.kind, cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads, cfg.stacked_feat,
     cfg.next_n_token, cfg.gated_agg, cfg.causal, cfg.max_position, cfg.num_labels, cfg.score_bias,
     cfg.pad_token_id) = spec.as_c_ints()
    cfg.rms_eps, cfg.rope_theta, cfg.layer_scale_init, cfg.max_tokens, cfg.max_batch = 1e-6, 1e4, 0.0, 8192, 256
    sz = L.GgetSizes()
    L.check(lib.gget_query_sizes(C.byref(cfg), C.byref(sz)))
    # the reference's "base" pre-train model has 122 094 336 parameters
    assert spec.num_params() == 122_094_336
    assert sz.n_params >= spec.num_params() and sz.n_params - spec.num_params() < 64 * 1024
    assert sz.workspace_bytes < 16 * 2 ** 30


def test_bad_config_is_an_error_not_a_crash(lib):
    cfg = L.GgetConfig()
    sz = L.GgetSizes()
    assert lib.gget_query_sizes(C.byref(cfg), C.byref(sz)) != 0
    assert b"hidden_size" in lib.gget_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a GPU-less box")
def test_no_cpu_fallback():
    eng = importlib.import_module("graph-gpt_amd.engine")
    spec = spec_mod.spec_from_size("tiny", vocab_size=300, stacked_feat=1, next_n_token=1)
    with pytest.raises(L.GgetError):
        eng.Engine(spec, 64, 4)
    from src.models import GraphGPTConfig, GraphGPTPretrainBase
    m = GraphGPTPretrainBase(GraphGPTConfig(hidden_act="gelu", vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                            num_attention_heads=2, causal_attention=False, stacked_feat=1, next_n_token=1))
    with pytest.raises(L.GgetError):
        m(input_ids=torch.zeros(2, 8, 1, dtype=torch.long))


def test_state_dict_keys_match_reference_names():
    from src.models import GraphGPTConfig, GraphGPTTaskModel
    m = GraphGPTTaskModel(GraphGPTConfig(hidden_act="gelu", vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                         num_attention_heads=2, causal_attention=False, stacked_feat=4, num_labels=2,
                                         layer_scale_init_value=1.0))
    keys = list(m.state_dict().keys())
    assert "model.layers.1.lambda_2" in keys and "score.weight" in keys and "model.norm.weight" in keys
    assert m.model.layers[0].self_attn.q_proj.weight.shape == (128, 128)
    assert m.model.embed_tokens.weight[0].abs().sum() == 0  # padding row


def test_lr_schedule_host_logic():
    import numpy as np
    from _util import GOLDEN
    tr = importlib.import_module("graph-gpt_amd.training")
    z = np.load(os.path.join(GOLDEN, "lr_schedules.npz"))
    for tag in ("a", "b"):
        max_lr, min_lr, total, warm = z["onecycle_" + tag + "_params"]
        oc = tr.OptimConfig(lr=max_lr, min_lr=min_lr, warmup_num_steps=int(warm), total_num_steps=int(total),
                            schedule="onecycle")
        got = [oc.lr_at(s) for s in range(int(total))]
        np.testing.assert_allclose(got, z["onecycle_" + tag], rtol=1e-9, atol=1e-15)
