"""Where does the bf16 engine's deviation from the fp32 reference come from (VERDICT r3 parity item 4a)?

CPU half (oracle only): the fp32 oracle forward is re-run with bf16 ROUNDING INSERTED at one class of the engine's cast points at a time
(every tensor the engine stores as bf16: residual stream, RMSNorm outputs, rotated q / k and v, softmax probabilities P feeding P.V,
attention output, gate / up / gelu / h of the MLP) - each run tells how much of the final-logit and loss deviation that class alone
produces, and the run with all of them on is the engine's cast-point model.  `emulate_reference_bf16` additionally rounds the o_proj /
down_proj outputs BEFORE the residual add - the one place where the engine rounds less often than the reference's bf16 modules (its GEMM
epilogues add the residual on the fp32 accumulator).
GPU half: the engine's own residual stream after every layer (gget_layer_hidden_states) against the fp32 oracle's, layer by layer, next to
the cast-point model's - if the two curves agree the engine has no error source beyond its documented cast points.

The full-size case (ft_base_s2048: base d768 / L12, S = 2048, B = 2 - the case at the edge of its tolerance in round 3) takes minutes of
CPU time per class: it runs when GGET_ERROR_BUDGET=1 and writes gpurun_out/error_budget_<case>.json (committed copy:
profiles/r04_error_budget_s2048.json); the always-on CPU test runs the same machinery on a small fixture."""
import importlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from _util import ROOT, ft_problem, load_case, record_error, rel_l2, tb
from oracle import gget_oracle as O

CLASSES = ("resid", "norm", "qkv", "P", "attn_out", "mlp")


def _rb(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def emulated_task_forward(spec, p, b, rounds=(), proj=False, collect=None):
    """fp32 task forward of the oracle (same formulas: O.rmsnorm, O.apply_rope, O.additive_mask ...; reference: hf LlamaDecoderLayer
    :295-325, LlamaAttention :243-281, LlamaMLP :174-176, modeling_finetune.py:236-326) with bf16 rounding at the selected cast points."""
    R = set(rounds)
    ids, am, pos = b["input_ids"], b["attention_mask"], b["position_ids"]
    B, S = ids.shape[:2]
    H, dh, d = spec.num_heads, spec.head_dim, spec.hidden_size
    ids = ids[:, :, : spec.stacked_feat] if ids.dim() == 3 else ids
    x, in_ = O.stacked_embed(p["model.embed_tokens.weight"], ids, p.get("stacked_feat_agg.weight"))
    x = _rb(x, "resid" in R)
    cos, sin = O.rope_cos_sin(pos, dh, spec.rope_theta, torch.float32)
    mask4d = O.additive_mask(am, S, torch.float32, spec.causal)
    for i in range(spec.num_layers):
        pre = f"model.layers.{i}."
        if collect is not None:
            collect.append(x)
        h = _rb(O.rmsnorm(x, p[pre + "input_layernorm.weight"], spec.rms_eps), "norm" in R)
        a = pre + "self_attn."
        q = Fnn.linear(h, p[a + "q_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
        k = Fnn.linear(h, p[a + "k_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
        v = _rb(Fnn.linear(h, p[a + "v_proj.weight"]), "qkv" in R).view(B, S, H, dh).transpose(1, 2)
        q, k = O.apply_rope(q, k, cos, sin)
        q, k = _rb(q, "qkv" in R), _rb(k, "qkv" in R)
        w = torch.matmul(q, k.transpose(2, 3)) * (dh ** -0.5) + mask4d
        w = _rb(Fnn.softmax(w, dim=-1, dtype=torch.float32), "P" in R)
        o = _rb(torch.matmul(w, v).transpose(1, 2).contiguous().view(B, S, d), "attn_out" in R)
        ao = _rb(Fnn.linear(o, p[a + "o_proj.weight"]), proj)
        if spec.layer_scale_init > 0:
            ao = p[pre + "lambda_1"] * ao
        x = _rb(x + ao, "resid" in R)
        h = _rb(O.rmsnorm(x, p[pre + "post_attention_layernorm.weight"], spec.rms_eps), "norm" in R)
        m = pre + "mlp."
        g = _rb(Fnn.linear(h, p[m + "gate_proj.weight"]), "mlp" in R)
        u = _rb(Fnn.linear(h, p[m + "up_proj.weight"]), "mlp" in R)
        hg = _rb(_rb(Fnn.gelu(g), "mlp" in R) * u, "mlp" in R)
        mo = _rb(Fnn.linear(hg, p[m + "down_proj.weight"]), proj)
        if spec.layer_scale_init > 0:
            mo = p[pre + "lambda_2"] * mo
        x = _rb(x + mo, "resid" in R)
    if collect is not None:
        collect.append(x)
    hid = _rb(O.rmsnorm(x, p["model.norm.weight"], spec.rms_eps), "norm" in R)
    last = (in_ != spec.pad_token_id).sum(-1) - 1
    pooled = hid[torch.arange(B), last]
    logits = Fnn.linear(pooled, p["score.weight"], p.get("score.bias"))
    return logits, Fnn.cross_entropy(logits, b["task_labels"])


def budget(name, classes=CLASSES):
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    out = {"case": name, "reference_fp32_loss": float(z["loss"]), "reference_bf16_loss": float(z["loss_bf16"]),
           "reference_bf16_logits_max_abs_dev": float(np.abs(z["logits_bf16"] - z["logits"]).max()), "classes": {}}
    with torch.no_grad():
        p32 = O.to_params(state, torch.float32, requires_grad=False)
        pbf = O.to_params(st_bf, torch.float32, requires_grad=False)
        col0 = []
        lg0, loss0 = emulated_task_forward(spec, p32, b, collect=col0)
        out["oracle_fp32_loss_vs_reference_fp32_rel"] = abs(float(loss0) - float(z["loss"])) / abs(float(z["loss"]))

        def run(params, rounds, proj=False):
            col = []
            lg, loss = emulated_task_forward(spec, params, b, rounds, proj, collect=col)
            real = b["attention_mask"].bool()
            return {"logits_max_abs_dev": float((lg - lg0).abs().max()), "loss_rel_dev": abs(float(loss) - float(loss0)) / abs(float(loss0)),
                    "loss_signed_rel_dev": (float(loss) - float(loss0)) / abs(float(loss0)),
                    "per_layer_residual_rel_l2": [rel_l2(c[real].numpy(), c0[real].numpy()) for c, c0 in zip(col, col0)]}

        out["classes"]["weights_only"] = run(pbf, ())
        for c in classes:
            out["classes"][c] = run(pbf, (c,))
        out["classes"]["engine_cast_points (all of the above)"] = run(pbf, classes)
        out["classes"]["reference_bf16_cast_points (+ projection outputs before the residual add)"] = run(pbf, classes, proj=True)
    return out, (spec, state, batch, col0, lg0)


def test_error_budget_machinery_small_case():
    """Always on (seconds): on the tiny fine-tune fixture the fp32 emulation with no rounding reproduces the reference's fp32 loss, and
    the all-cast-points model lands in the bf16 class of the reference's own bf16 path (same order of magnitude of logit deviation)."""
    out, _ = budget("ft_tiny_f4_b32")
    assert out["oracle_fp32_loss_vs_reference_fp32_rel"] < 2e-5
    eng = out["classes"]["engine_cast_points (all of the above)"]
    ref = out["reference_bf16_logits_max_abs_dev"]
    assert 0.1 * ref <= eng["logits_max_abs_dev"] <= 10 * ref, (eng["logits_max_abs_dev"], ref)
    assert all(out["classes"][c]["logits_max_abs_dev"] <= 10 * ref for c in CLASSES)


@pytest.mark.skipif(not os.environ.get("GGET_ERROR_BUDGET"), reason="minutes of CPU time: set GGET_ERROR_BUDGET=1 (output committed under profiles/)")
@pytest.mark.parametrize("name", ["ft_base_s2048"])
def test_error_budget_full_size_cpu(name):
    out, _ = budget(name)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"error_budget_{name}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    assert out["oracle_fp32_loss_vs_reference_fp32_rel"] < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ft_base_ls_s256", "ft_base_s2048"])
def test_engine_per_layer_error_matches_cast_point_model(name):
    """The engine's residual stream after every decoder layer against the fp32 oracle's (real tokens), next to the cast-point model's:
    the engine must not be worse than 1.5 x the model at any layer (it has no error source beyond its documented bf16 cast points), and
    its final logits must sit inside 2 x the reference's own bf16-vs-fp32 logit gap."""
    eng_mod = importlib.import_module("graph-gpt_amd.engine")
    L = importlib.import_module("graph-gpt_amd._lib")
    out, (spec, state, batch, col0, lg0) = budget(name, classes=())       # fp32 reference curve + weights-only
    b = tb(batch)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    with torch.no_grad():
        col_m = []
        lg_m, _ = emulated_task_forward(spec, O.to_params(st_bf, torch.float32, requires_grad=False), b, CLASSES, collect=col_m)
    B, S = batch["input_ids"].shape[:2]
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL)
    torch.cuda.synchronize()
    real = b["attention_mask"].bool()
    rows = []
    for i in range(spec.num_layers + 1):
        x = e.layer_hidden_states(i, B, S).float().cpu()
        eng = rel_l2(x[real].numpy(), col0[i][real].numpy())
        mod = rel_l2(col_m[i][real].numpy(), col0[i][real].numpy())
        rows.append({"layer_input": i, "engine_rel_l2": eng, "cast_point_model_rel_l2": mod})
        record_error(name, f"residual_stream_rel_l2_vs_fp32 entering layer {i} (cast-point model {mod:.2e})", eng, 1.5 * mod + 1e-4)
        assert eng <= 1.5 * mod + 1e-4, rows
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    dev_eng = float((logits.cpu() - lg0).abs().max())
    dev_mod = float((lg_m - lg0).abs().max())
    dev_ref = float(np.abs(z["logits_bf16"] - z["logits"]).max())
    rec = {"case": name, "per_layer": rows, "logits_max_abs_dev": {"engine": dev_eng, "cast_point_model": dev_mod, "reference_bf16": dev_ref},
           "loss": {"engine": float(loss), "fp32": float(z["loss"]), "reference_bf16": float(z["loss_bf16"])}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"error_budget_gpu_{name}.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    assert dev_eng <= 2 * max(dev_ref, dev_mod) + 1e-3, rec["logits_max_abs_dev"]
