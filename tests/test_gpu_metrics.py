"""SURVEY.md row A14 on the device: the fine-tune metrics of the reference (SingleLabelClassificationMetrics.update / compute -
src/utils/metrics_utils.py:38-56,143-189; `_eval_ogbl_ppa` Hits@K - src/utils/ogb_utils.py:83-90; PCQM4Mv2 MAE - :199-204) computed
by `ft_evaluate` (log_eval_dump_utils.py:77-163) from the HIP engine's `task_logits`, against the same metrics computed from the
REFERENCE's logits held by the golden fixtures (tools/make_golden.py ran the real reference).  Tolerances are derived from the logit
deviation, not picked: a rank metric can only move by the pairs / samples whose reference margin is smaller than twice the largest
score deviation; MAE by at most the mean absolute logit deviation."""
import importlib

import numpy as np
import pytest
import torch

from _util import ft_problem, load_case, record_error

pytestmark = pytest.mark.gpu

M = importlib.import_module("graph-gpt_amd.modeling")
tr = importlib.import_module("graph-gpt_amd.training")
met = importlib.import_module("graph-gpt_amd.metrics")


def _model_for(spec, state, problem_type, loss_type):
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=spec.vocab_size, hidden_size=spec.hidden_size,
                           intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
                           num_attention_heads=spec.num_heads, max_position_embeddings=spec.max_position,
                           causal_attention=spec.causal, stacked_feat=spec.stacked_feat, num_labels=spec.num_labels,
                           problem_type=problem_type, loss_type=loss_type, layer_scale_init_value=spec.layer_scale_init,
                           rms_norm_eps=spec.rms_eps, pad_token_id=spec.pad_token_id,
                           stacked_feat_agg_method="gated" if spec.gated_agg else "sum")
    model = M.GraphGPTTaskModel(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return model.cuda()


def _loader(batch, n_parts, device_tensors):
    """the fixture's batch cut into `n_parts` collated batches with running sample indices (what the eval DataLoader yields)"""
    B = batch["input_ids"].shape[0]
    cuts = np.linspace(0, B, n_parts + 1).astype(int)
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        d = {k: torch.from_numpy(np.ascontiguousarray(v[a:b])) for k, v in batch.items() if k != "lengths"}
        d["idx"] = torch.arange(a, b)
        if device_tensors:
            d = {k: v.cuda() for k, v in d.items()}
        out.append(d)
    return out


@pytest.mark.parametrize("device_tensors", [False, True])
@pytest.mark.parametrize("name", ["ft_tiny_f4_b32", "ft_base_ls_s256", "ft_tiny_f4", "ft_tiny_wce"])
def test_classification_metrics_from_hip_logits(name, device_tensors):
    z, spec, state, batch = load_case(name)
    b = {k: torch.from_numpy(v) for k, v in batch.items()}
    problem, loss_type = ft_problem(spec, b, name)
    assert problem == "single_label_classification"
    if "wgt" in batch:      # (the weighted-CE fixture: the weights only enter the loss)
        batch = dict(batch)
    model = _model_for(spec, state, problem, loss_type)
    y = batch["task_labels"].astype(np.int64)
    parts = 2 if y.shape[0] >= 8 else 1
    loss, m, res, d = tr.ft_evaluate(model, _loader(batch, parts, device_tensors), problem_type=problem, num_labels=spec.num_labels,
                                     dataset_name="ogbl-ppa")
    assert model.training                                   # ft_evaluate puts the model back in train mode
    lg_ref = z["logits"].astype(np.float64)
    s_ref = lg_ref[:, 1] - lg_ref[:, 0]
    s_hip = d["y_pred"].numpy().astype(np.float64)
    assert list(d["idx"].numpy()) == list(range(y.shape[0])) and np.array_equal(d["y_true"].numpy(), y)
    delta = float(np.abs(s_hip - s_ref).max())
    scale = float(np.abs(s_ref).max())
    record_error(name, f"metrics_score_max_abs_dev_rel ({'device' if device_tensors else 'host'} batches)", delta / scale, 4e-2)
    assert delta <= 4e-2 * scale, (delta, scale)            # the bf16-class logit tolerance the forward tests hold the engine to
    pos, neg = s_ref[y == 1], s_ref[y == 0]
    # accuracy: only samples whose reference score sits within delta of the decision boundary can flip
    acc_ref = met.accuracy(lg_ref, y)
    acc_tol = float((np.abs(s_ref) <= delta).mean())
    record_error(name, "acc_abs_dev", abs(m.acc - acc_ref), acc_tol + 1e-12)
    assert abs(m.acc - acc_ref) <= acc_tol + 1e-12, (m.acc, acc_ref, acc_tol)
    if len(pos) and len(neg):
        # AUROC / Hits@K: only (positive, negative) pairs whose reference margin is below 2 delta can change order
        close = float((np.abs(pos[:, None] - neg[None, :]) <= 2 * delta).mean())
        auroc_ref = met.auroc(s_ref, y)
        record_error(name, "auroc_abs_dev", abs(m.auroc - auroc_ref), close + 1e-12)
        assert abs(m.auroc - auroc_ref) <= close + 1e-12, (m.auroc, auroc_ref, close)
        for k in (1, 3, 100):
            thr = np.sort(neg)[::-1][min(k, len(neg)) - 1] if len(neg) >= k else -np.inf
            hits_tol = float((np.abs(pos - thr) <= 2 * delta).mean()) if np.isfinite(thr) else 0.0
            got = met.hits_at_k(s_hip[y == 1], s_hip[y == 0], k)
            want = met.hits_at_k(pos, neg, k)
            record_error(name, f"hits@{k}_abs_dev", abs(got - want), hits_tol + 1e-12)
            assert abs(got - want) <= hits_tol + 1e-12, (k, got, want, hits_tol)
        assert res == {"hits@100": met.hits_at_k(s_hip[y == 1], s_hip[y == 0], 100)}     # `_eval_ogbl_ppa` on the gathered dict
    # the evaluation loss is the reference's task loss of the batch (mean over equally sized parts)
    if parts == 1 or name == "ft_tiny_f4_b32":
        ce = lambda lg, yy: float(torch.nn.functional.cross_entropy(torch.from_numpy(lg).float(), torch.from_numpy(yy)))
        cuts = np.linspace(0, y.shape[0], parts + 1).astype(int)
        want_loss = float(np.mean([ce(z["logits"][a:b_], y[a:b_]) for a, b_ in zip(cuts[:-1], cuts[1:])])) if "wgt" not in batch else None
        if want_loss is not None:
            assert abs(float(loss) - want_loss) <= 2e-2 * max(abs(want_loss), 0.1), (float(loss), want_loss)


@pytest.mark.parametrize("name", ["ft_tiny_reg", "ft_tiny_mse"])
def test_regression_mae_from_hip_logits(name):
    z, spec, state, batch = load_case(name)
    b = {k: torch.from_numpy(v) for k, v in batch.items()}
    problem, loss_type = ft_problem(spec, b, name)
    assert problem == "regression"
    model = _model_for(spec, state, problem, loss_type)
    loss, m, res, d = tr.ft_evaluate(model, _loader(batch, 1, True), problem_type="regression", num_labels=1, dataset_name="PCQM4Mv2")
    y = batch["task_labels"].astype(np.float64)
    ref = z["logits"].astype(np.float64).reshape(-1)
    hip = d["y_pred"].numpy().astype(np.float64).reshape(-1)
    dev = float(np.abs(hip - ref).mean())
    mae_ref = met.mae(ref, y)
    record_error(name, "mae_abs_dev (bound: mean |logit deviation|)", abs(res["mae"] - mae_ref), dev + 1e-9)
    assert abs(res["mae"] - mae_ref) <= dev + 1e-9                      # triangle inequality
    assert dev <= 4e-2 * max(float(np.abs(ref).max()), 1e-3)
    assert res["mae"] == m.mae
    if loss_type == "l1":                                               # the reference's L1 task loss IS the MAE (modeling_finetune.py:183-197)
        assert abs(float(loss) - res["mae"]) <= 2e-6 * max(res["mae"], 1e-3) + 1e-6


@pytest.mark.parametrize("name", ["ft_tiny_f4", "ft_tiny_ls"])
def test_output_hidden_states_field_matches_oracle(name):
    """`DoubleHeadsModelOutput.hidden_states` (reference modeling_finetune.py:316-326 returns `outputs.hidden_states` of the backbone:
    hf LlamaModel.forward :401-414, L + 1 tensors - the stream entering every layer, then the final-normed output): filled when
    `output_hidden_states=True`, None otherwise; every tensor against the fp32 oracle on the real tokens (bf16-class tolerance), and
    the forward that produced them gives the same loss / logits as the plain call."""
    from oracle import gget_oracle as O
    from _util import rel_l2
    z, spec, state, batch = load_case(name)
    b = {k: torch.from_numpy(v) for k, v in batch.items()}
    problem, loss_type = ft_problem(spec, b, name)
    model = _model_for(spec, state, problem, loss_type).eval()
    kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], task_labels=b["task_labels"])
    with torch.no_grad():
        plain = model(**kw)
        out = model(output_hidden_states=True, **kw)
    assert plain.hidden_states is None and plain.attentions is None
    hs = out.hidden_states
    assert isinstance(hs, tuple) and len(hs) == spec.num_layers + 1
    assert float(out.task_loss) == pytest.approx(float(plain.task_loss), rel=1e-6) and torch.equal(out.task_logits, plain.task_logits)
    p = O.to_params(state, torch.float32, requires_grad=False)
    col = []
    with torch.no_grad():
        x, _ = O.stacked_embed(p["model.embed_tokens.weight"], b["input_ids"][:, :, : spec.stacked_feat], p.get("stacked_feat_agg.weight"))
        final = O.backbone(spec, p, x, b["attention_mask"], b["position_ids"], collect=col)
    want = [x] + col[:-1] + [final]          # entering layer 0 .. L-1, then norm(leaving the last layer)
    real = b["attention_mask"].bool()
    B, S = b["input_ids"].shape[:2]
    for i, (g, w) in enumerate(zip(hs, want)):
        assert tuple(g.shape) == (B, S, spec.hidden_size) and g.dtype == torch.bfloat16
        err = rel_l2(g.float().cpu()[real].numpy(), w[real].numpy())
        assert err < 2e-2, (i, err)
    last = (b["attention_mask"].sum(-1) - 1).long()
    assert torch.equal(hs[-1][torch.arange(B), last.cuda()], out.task_hidden_states)     # the pooled row IS a row of the last tensor
