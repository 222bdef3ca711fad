"""Var-len (padding-free) token layout of the engine (include/gget.h: gget_set_token_count) against the padded layout and the
reference fixtures.  The reference runs every token-wise module over the padded [B,S] grid (modeling_helpers.py:38-64) - pad
rows never influence real rows - so the two layouts must agree on every reference-visible output: loss, head logits, task
logits, every gradient.  Not bit-wise: a different row count selects other GEMM tile shapes / K splits (fp32 summation order),
whose differences surface as isolated bf16 rounding flips; the tolerances below are a tenth of the bf16-class tolerances the
padded path is held to against the fp32 reference."""
import importlib

import numpy as np
import pytest
import torch

from _util import FT_CASES, PT_CASES, ft_problem, load_case, loss_tolerance, record_error, rel_l2, tb

pytestmark = pytest.mark.gpu

eng_mod = importlib.import_module("graph-gpt_amd.engine")
L = importlib.import_module("graph-gpt_amd._lib")
spec_mod = importlib.import_module("graph-gpt_amd.spec")
weights_mod = importlib.import_module("graph-gpt_amd.weights")
synth = importlib.import_module("graph-gpt_amd.synth")


def _forward(e, spec, b, kind, name, n_tok):
    if kind == "pt":
        return e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"), num_tokens=n_tok), None
    pt_, lt_ = ft_problem(spec, b, name)
    problem = {"regression": L.PROBLEM_REGRESSION_L1 if lt_ == "l1" else L.PROBLEM_REGRESSION_MSE,
               "multi_label_classification": L.PROBLEM_MULTI_LABEL, "single_label_classification": L.PROBLEM_SINGLE_LABEL}[pt_]
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], b.get("wgt"), problem,
                                     num_tokens=n_tok)
    return loss, logits


def _run(spec, state, batch, kind, name, varlen, dropout=None, seed=77):
    b = tb(batch)
    B, S = batch["input_ids"].shape[:2]
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)
    if dropout:
        e.set_dropout(dropout[0], dropout[1], seed)
    n_tok = int(batch["attention_mask"].sum()) if varlen else None
    loss, tlog = _forward(e, spec, b, kind, name, n_tok)
    e.backward()
    torch.cuda.synchronize()
    ran, rows, mismatch = e.varlen_status()
    out = {"loss": float(loss.item()), "grads": {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()},
           "varlen": ran, "rows": rows, "mismatch": mismatch}
    if kind == "pt":
        out["logits"] = e.head_logits().float().cpu().numpy()
    else:
        out["logits"] = tlog.cpu().numpy().copy()
    return out


def _compare(name, a, b, loss_tol=2e-5, logit_tol=3e-3, grad_tol=6e-3):
    """a: padded, b: var-len"""
    assert abs(a["loss"] - b["loss"]) <= loss_tol * abs(a["loss"]) + 1e-7, f"{name}: loss {a['loss']} (padded) vs {b['loss']} (var-len)"
    record_error(name, "varlen_vs_padded_loss_rel", abs(a["loss"] - b["loss"]) / (abs(a["loss"]) + 1e-30), loss_tol)
    assert a["logits"].shape == b["logits"].shape
    err = rel_l2(b["logits"], a["logits"])
    record_error(name, "varlen_vs_padded_logits_rel_l2", err, logit_tol)
    assert err < logit_tol, f"{name}: logits rel-L2 {err}"
    gmax = max(float(np.linalg.norm(v)) for v in a["grads"].values())
    worst = (0.0, "")
    for k, ga in a["grads"].items():
        gb = b["grads"][k]
        scale = max(float(np.linalg.norm(ga)), 1e-2 * gmax)
        err = float(np.linalg.norm(ga.astype(np.float64) - gb)) / scale
        worst = max(worst, (err, k))
    record_error(name, "varlen_vs_padded_worst_grad_rel_l2 (" + worst[1] + ")", worst[0], grad_tol)
    assert worst[0] < grad_tol, f"{name}: gradient {worst[1]} differs by {worst[0]} between the layouts"


@pytest.mark.parametrize("name", [c for c in PT_CASES + FT_CASES if c != "pt_tiny_packed"])
def test_varlen_matches_padded_and_reference_on_fixtures(name):
    """Every reference fixture with right padding: the var-len run reproduces the padded run AND sits within the padded path's
    own tolerance of the reference's fp32 loss."""
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    pad = _run(spec, state, batch, kind, name, varlen=False)
    vl = _run(spec, state, batch, kind, name, varlen=True)
    B, S = batch["input_ids"].shape[:2]
    n_tok = int(batch["attention_mask"].sum())
    assert not pad["varlen"] and pad["rows"] == B * S
    if (n_tok + 63) // 64 * 64 < B * S:
        assert vl["varlen"] and vl["rows"] == (n_tok + 63) // 64 * 64 and not vl["mismatch"]
    else:
        assert not vl["varlen"]          # nothing to gain: the engine keeps the padded rows
    _compare(name, pad, vl)
    want = float(z["loss"])
    from test_gpu_model import FT_FACTOR, LOSS_FLOOR
    tol = max(loss_tolerance(z, factor=1.5 if kind == "pt" else FT_FACTOR), LOSS_FLOOR.get(name, 0.0))
    assert abs(vl["loss"] - want) <= tol * abs(want) + 1e-6, f"{name}: var-len loss {vl['loss']} vs reference fp32 {want}"


def _tiny_spec(kind, S, causal=False, layer_scale=0.0, path_pdrop=0.0, F=4, V=500, layers=2):
    return spec_mod.ModelSpec(kind=kind, vocab_size=V, hidden_size=128, intermediate_size=512, num_layers=layers, num_heads=2,
                              head_dim=64, stacked_feat=F, next_n_token=F if kind == spec_mod.KIND_PRETRAIN else 1, gated_agg=False,
                              causal=causal, max_position=max(1024, S), num_labels=2, score_bias=False, pad_token_id=0,
                              layer_scale_init=layer_scale, path_pdrop=path_pdrop)


@pytest.mark.parametrize("S,B,causal", [(24, 12, False), (32, 64, False), (72, 6, False), (160, 5, True), (256, 6, False), (320, 4, True),
                                         (640, 3, False), (1088, 2, False),
                                         # 32 < S <= 64: every sample by its own row count (one-tile kernels + the 33 .. 64-row launches)
                                         (40, 40, False), (48, 9, True), (56, 70, False), (64, 16, False)])
@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_varlen_every_attention_kernel_class(kind, S, B, causal):
    """S <= 32 (one-wave kernels), 32 < S <= 64 (var-len: every sample by its own row count - attn_fwd_long_kernel / attn_bwd_long_kernel for
    the samples of 33 .. 64 rows; padded: the two-tile backward for all), 64 < S < 256 (multi-wave register-prefetch kernels), S >= 256 (64-row LDS-DMA stages; the dense
    pipelined forward from S = 512 when not causal; 128-row dK/dV stages from S = 512), with attention dropout 0.1 (the masks are
    hashes of the LOGICAL (b, h, q, k) coordinates, so both layouts draw the same one), ragged lengths incl. very short samples."""
    pt = kind == "pt"
    spec = _tiny_spec(spec_mod.KIND_PRETRAIN if pt else spec_mod.KIND_TASK, S, causal=causal)
    state = weights_mod.make_state_dict(spec, seed=3, std=0.06, head_std=0.15)
    if pt:
        batch = synth.make_pretrain_batch(B=B, S=S, F=4, V=500, seed=11 + S, lengths="uniform", min_len=max(2, S // 8))
    else:
        batch = synth.make_task_batch(B=B, S=S, F=4, V=500, seed=13 + S, lengths="uniform", min_len=max(2, S // 8))
    batch = {k: v for k, v in batch.items() if k != "lengths"}
    name = f"varlen_{kind}_S{S}{'_causal' if causal else ''}"
    pad = _run(spec, state, batch, kind, "", varlen=False, dropout=(0.1, 0.0))
    vl = _run(spec, state, batch, kind, "", varlen=True, dropout=(0.1, 0.0))
    assert vl["varlen"] and not vl["mismatch"] and vl["rows"] < B * S
    _compare(name, pad, vl)


def test_varlen_layerscale_droppath_training_mode():
    """The ogbl-ppa fine-tune form (LayerScale 1.0, stochastic depth 0.2, attention dropout 0.1; examples/edge_lvl/ppa_supervised.sh):
    DropPath draws one decision per SAMPLE - the compact rows carry their sample index - so both layouts drop the same branches."""
    S, B = 96, 24
    spec = _tiny_spec(spec_mod.KIND_TASK, S, layer_scale=1.0, path_pdrop=0.2, layers=3)
    state = weights_mod.make_state_dict(spec, seed=5, std=0.06, head_std=0.15)
    batch = {k: v for k, v in synth.make_task_batch(B=B, S=S, F=4, V=500, seed=21, lengths="uniform", min_len=8).items() if k != "lengths"}
    pad = _run(spec, state, batch, "ft", "", varlen=False, dropout=(0.1, 0.2), seed=1234)
    vl = _run(spec, state, batch, "ft", "", varlen=True, dropout=(0.1, 0.2), seed=1234)
    assert vl["varlen"]
    _compare("varlen_ft_layerscale_droppath", pad, vl)
    # the mask is really applied: another seed moves the loss
    other = _run(spec, state, batch, "ft", "", varlen=True, dropout=(0.1, 0.2), seed=99)
    assert abs(other["loss"] - vl["loss"]) > 1e-5


def test_varlen_wrong_token_count_is_flagged_and_fallbacks():
    """A caller's count that disagrees with the mask raises the device-side flag (gget_varlen_status); full-logit inference and full
    rows keep the padded layout by themselves; element dropouts no longer do (round 5: their hashes are keyed by the logical row)."""
    spec = _tiny_spec(spec_mod.KIND_PRETRAIN, 32)
    state = weights_mod.make_state_dict(spec, seed=3)
    batch = synth.make_pretrain_batch(B=16, S=32, F=4, V=500, seed=5)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    e = eng_mod.Engine(spec, max_tokens=16 * 32, max_batch=16)
    e.load_state_dict(state)
    n = int(batch["attention_mask"].sum())
    e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n - 3)
    assert e.varlen_status() == (True, (n - 3 + 63) // 64 * 64, True)
    e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n)
    assert e.varlen_status() == (True, (n + 63) // 64 * 64, False)
    # hidden-state accessors after a var-len forward (round 6: gget_hidden_states_grid spreads the compact rows back over [B,S,d];
    # positions behind a sample's tokens read as zero) against the padded forward's
    hv, lv = e.hidden_states(16, 32).float().cpu(), [e.layer_hidden_states(i, 16, 32).float().cpu() for i in range(spec.num_layers + 1)]
    e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"])           # no count -> padded
    assert e.varlen_status()[0] is False
    real = torch.from_numpy(batch["attention_mask"]).bool()
    hp, lp = e.hidden_states(16, 32).float().cpu(), [e.layer_hidden_states(i, 16, 32).float().cpu() for i in range(spec.num_layers + 1)]
    for name_, a_, p_ in [("final", hv, hp)] + [(f"layer {i}", x, y) for i, (x, y) in enumerate(zip(lv, lp))]:
        assert float(a_[~real].abs().max()) == 0.0, name_
        assert rel_l2(a_[real].numpy(), p_[real].numpy()) < 6e-3, name_
    assert torch.equal(lv[0][real], lp[0][real])                                    # (the embedding sum: the same arithmetic per row)
    e.forward_pretrain(b["input_ids"], b["attention_mask"], None, num_tokens=n)    # full-logit inference: var-len since round 5, no flag for
    assert e.varlen_status() == (True, (n + 63) // 64 * 64, False)                 # the cells of padded positions
    full = torch.ones_like(b["attention_mask"])
    e.forward_pretrain(b["input_ids"], full, b["labels"], num_tokens=16 * 32)       # no padding -> nothing to compact
    assert e.varlen_status()[0] is False
    e.set_dropout_ex(0.1, 0.0, 0.0)                                                 # element dropouts hash the LOGICAL row: no fallback
    e.set_dropout(0.0, 0.0, 5)
    e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n)
    assert e.varlen_status() == (True, (n + 63) // 64 * 64, False)


@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_varlen_element_dropouts_draw_the_padded_grids_masks(kind):
    """embed_dropout + the two MLP dropouts (utils_graphgpt.py:69-80, modeling_helpers.py:96-101) on the var-len layout: the counter
    hashes are keyed by the logical [B,S] row (ElemDropArg::rows), so the compact step must reproduce the padded step's loss and
    gradients with the SAME seed - exactly the masks the Python twins pin against the oracle in
    tests/test_gpu_model.py::test_embed_and_mlp_dropouts_exact_mask.  ft: LayerScale + DropPath + attention dropout on top (the C3
    training configuration with mlp_pdrop > 0)."""
    S, F, V, B = 40, 4, 500, 24
    import dataclasses
    if kind == "pt":
        spec = dataclasses.replace(_tiny_spec(spec_mod.KIND_PRETRAIN, S), mlp_pdrop=0.2, embed_pdrop=0.15)
        batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=15)
    else:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                       layer_scale_init=1.0, path_pdrop=0.2, gated_agg=True, mlp_pdrop=0.2, embed_pdrop=0.15)
        batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=15)
    state = weights_mod.make_state_dict(spec, seed=3, std=0.05, head_std=0.1)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    n = int(batch["attention_mask"].sum())
    out = {}
    for lay, cnt in (("padded", None), ("varlen", n)):
        e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
        e.load_state_dict(state)
        e.set_dropout(0.1, 0.2 if kind == "ft" else 0.0, 77)
        e.set_dropout_ex(0.15, 0.2, 0.0)
        if kind == "pt":
            loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=cnt)
        else:
            loss, _, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                        num_tokens=cnt)
        assert e.varlen_status()[0] == (lay == "varlen")
        e.backward()
        torch.cuda.synchronize()
        out[lay] = (float(loss), {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()})
        # ... and the masks really are on: the same step without the element dropouts gives another loss
        if lay == "padded":
            e.set_dropout_ex(0.0, 0.0, 0.0)
            l0 = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"]) if kind == "pt" else \
                e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL)[0]
            assert abs(float(l0) - float(loss)) > 1e-4 * abs(float(loss))      # (measured 7e-4: the head's logit variance dominates this loss)
    (lp, gp), (lv, gv) = out["padded"], out["varlen"]
    assert abs(lv - lp) <= 2e-5 * abs(lp), (lv, lp)
    gmax = max(float(np.linalg.norm(g)) for g in gp.values())
    for k in gp:
        err = float(np.linalg.norm(gv[k] - gp[k])) / max(float(np.linalg.norm(gp[k])), 1e-2 * gmax)
        assert err < 1e-2, (k, err)        # same masks, bf16 gradients, another summation order over the rows


def test_varlen_rope_range_matches_padded():
    """rope_range > 0 (utils_graphgpt.reset_pos_ids :574-581: positions rescaled per row, angles evaluated per token): on the var-len
    layout a compact row reads the angle-table row of its logical token - loss, logits and gradients of the padded run."""
    S, F, V, B = 48, 4, 500, 12
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2, rope_range=6.0)
    state = weights_mod.make_state_dict(spec, seed=4, std=0.05, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=16)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    n = int(batch["attention_mask"].sum())
    out = {}
    for lay, cnt in (("padded", None), ("varlen", n)):
        e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
        e.load_state_dict(state)
        e.set_rope_range(6.0)
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                         num_tokens=cnt)
        assert e.varlen_status()[0] == (lay == "varlen")
        e.backward()
        torch.cuda.synchronize()
        out[lay] = (float(loss), logits.float().cpu().numpy().copy(), {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()})
    (lp, zp, gp), (lv, zv, gv) = out["padded"], out["varlen"]
    assert abs(lv - lp) <= 2e-5 * abs(lp) and np.abs(zv - zp).max() <= 1e-3 * max(1.0, np.abs(zp).max())
    gmax = max(float(np.linalg.norm(g)) for g in gp.values())
    for k in gp:
        assert float(np.linalg.norm(gv[k] - gp[k])) / max(float(np.linalg.norm(gp[k])), 1e-2 * gmax) < 1e-2, k


def test_varlen_model_classes_and_training_step(monkeypatch):
    """Through the drop-in classes: a HOST-side attention mask (what the reference's loops receive from the DataLoader) or
    data["num_tokens"] selects the var-len layout; three clip + AdamW steps track the padded run."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
    batch = synth.make_pretrain_batch(B=32, S=32, F=13, V=756, seed=8)
    n = synth.real_tokens(batch)
    host = {k: torch.from_numpy(v) for k, v in batch.items() if k != "lengths"}
    dev = {k: v.cuda() for k, v in host.items()}

    def run(data):
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda().eval()
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
        losses = [float(tr.batch_training(data, eng)) for _ in range(3)]
        torch.cuda.synchronize()
        return losses, model._engine.varlen_status()[0], model._engine.master.detach().cpu().numpy().copy()

    monkeypatch.setenv("GGET_VARLEN", "nosync")
    lp, vp, mp = run(dev)                                   # device mask, counting on the device switched off: padded
    monkeypatch.delenv("GGET_VARLEN")
    ld, vd, md = run(dev)                                   # device mask, no count: the engine counts it (GGET_TOKENS_AUTO)
    lh, vh, mh = run(host)                                  # host mask: counted for free
    ln, vn, mn = run(dict(dev, num_tokens=n))               # explicit count
    assert (vp, vd, vh, vn) == (False, True, True, True)
    np.testing.assert_allclose(lh, lp, rtol=5e-4)
    np.testing.assert_allclose(ln, lp, rtol=5e-4)
    np.testing.assert_array_equal(mh, mn)                   # the three var-len runs are the same computation
    np.testing.assert_array_equal(md, mn)
    np.testing.assert_allclose(ld, ln, rtol=2e-6)           # (the loss sum is an fp32-atomic reduction: equal up to its order)
    upd = np.linalg.norm(mp - M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda()._engine.master.detach().cpu().numpy())
    assert np.linalg.norm(mh - mp) < 0.05 * upd
    monkeypatch.setenv("GGET_VARLEN", "0")
    assert run(host)[1] is False
    assert run(dev)[1] is False


def test_varlen_c1_full_size_matches_padded():
    """BASELINE's headline shape (base d768 / L12, B = 256, S = 32, F = 13, V = 756, attention dropout 0.1): loss and every gradient
    of the var-len step against the padded step."""
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13, causal=False,
                                   max_position=1024)
    state = weights_mod.make_state_dict(spec, seed=0)
    batch = {k: v for k, v in synth.make_pretrain_batch(B=256, S=32, F=13, V=756, seed=1234).items() if k != "lengths"}
    lib = L.load()
    try:
        # (i) both layouts on the SAME summation orders (no in-block K split, no stream-K, 256-row tiles only: gget_debug_set key 1):
        # every real row then goes through identical arithmetic, so the forward must agree bit for bit - any difference would be an
        # indexing error of the compact layout, not rounding
        L.check(lib.gget_debug_set(1, 1 | 2 | 4 | 8))
        pad = _run(spec, state, batch, "pt", "", varlen=False, dropout=(0.1, 0.0))
        vl = _run(spec, state, batch, "pt", "", varlen=True, dropout=(0.1, 0.0))
        assert vl["varlen"] and vl["rows"] < 0.8 * 256 * 32
        assert np.array_equal(pad["logits"], vl["logits"]), "head logits of the two layouts differ with identical kernels"
        _compare("varlen_c1_full_size_same_kernels", pad, vl, loss_tol=2e-6, logit_tol=1e-9, grad_tol=2e-3)
    finally:
        L.check(lib.gget_debug_set(1, 0))
    # (ii) the shipped kernel selection (192-row tiles, stream-K, K-split kernels differ between the layouts): bf16-class agreement - at
    # the standard init the logits carry ~1e-2 of bf16 noise against fp32 in EITHER layout (tests/test_gpu_model.py holds the padded
    # path to max(2.5 x reference bf16 error, 1.5e-2)), and two summation orders draw two noise patterns
    pad = _run(spec, state, batch, "pt", "", varlen=False, dropout=(0.1, 0.0))
    vl = _run(spec, state, batch, "pt", "", varlen=True, dropout=(0.1, 0.0))
    _compare("varlen_c1_full_size", pad, vl, loss_tol=5e-5, logit_tol=1.5e-2, grad_tol=3e-2)


def test_varlen_changing_batches_do_not_leak_stale_rows():
    """Six optimiser steps on six DIFFERENT batches (another real-token count every step, shrinking and growing: rows a previous, larger
    step left behind in the token-major buffers lie beyond the current row count, the <= 63 tail rows inside it): the var-len run must
    track the padded run step by step - a stale row read anywhere would show up as a jump."""
    spec = _tiny_spec(spec_mod.KIND_PRETRAIN, 48, layers=2)
    state = weights_mod.make_state_dict(spec, seed=4, std=0.05, head_std=0.1)
    B, S = 24, 48
    mins = [40, 4, 30, 2, 44, 12]      # shortest sample of each batch -> very different totals
    batches = [{k: v for k, v in synth.make_pretrain_batch(B=B, S=S, F=4, V=500, seed=100 + i, lengths="uniform", min_len=m).items()
                if k != "lengths"} for i, m in enumerate(mins)]

    def run(varlen):
        e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
        e.load_state_dict(state)
        out, rows = [], []
        for bt in batches:
            b = tb(bt)
            n = int(bt["attention_mask"].sum()) if varlen else None
            e.set_dropout(0.1, 0.0, 11)
            loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n)
            e.backward()
            e.adamw_step(1e-3, max_grad_norm=1.0)
            out.append(float(loss.item()))
            rows.append(e.varlen_status()[1])
        return out, rows, e.master.detach().cpu().numpy().copy()

    lp, rp, mp = run(False)
    lv, rv, mv = run(True)
    # (the fifth batch is so full that round_up(real, 64) reaches B * S: the engine keeps the padded rows for that step - the run mixes
    #  the layouts step by step)
    assert len(set(rv)) >= 4 and sum(r < B * S for r in rv) >= 5 and all(r == B * S for r in rp)
    np.testing.assert_allclose(lv, lp, rtol=3e-4)
    base = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    base.load_state_dict(state)
    upd = np.linalg.norm(mp - base.master.detach().cpu().numpy())
    assert np.linalg.norm(mv - mp) < 0.03 * upd, (np.linalg.norm(mv - mp), upd)


def test_reference_shaped_step_runs_varlen_from_device_tensors():
    """a step shaped EXACTLY like the reference's (training_utils.py:7-45: positional `batch_training(data, model,
    train_cfg, train_stats, opt_stats)`, every tensor moved with `.to(device)` before the model sees it, no `num_tokens` keyword
    anywhere) runs on the compact layout - the engine counts the device-side mask itself - and is the SAME computation as the step that
    was handed the count: loss and every master weight after three clip + AdamW steps are bit-equal."""
    import types
    M = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    from src.utils.training_utils import batch_training as ref_batch_training          # the drop-in import path
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
    batch = synth.make_pretrain_batch(B=32, S=32, F=13, V=756, seed=8)
    n = synth.real_tokens(batch)
    host = {k: torch.from_numpy(v) for k, v in batch.items() if k != "lengths"}
    host["position_ids"] = torch.arange(32)[None, :].repeat(32, 1)

    def run(reference_form):
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda().eval()
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
        stats = types.SimpleNamespace(device=torch.device("cuda"), has_embeds_input=False, use_deepspeed=True)
        losses = []
        for _ in range(3):
            if reference_form:
                ref_batch_training(host, eng, types.SimpleNamespace(optimizer=None), stats, None)
                losses.append(float(stats.loss))
                assert stats.main_loss is stats.loss and stats.aux_loss is None and tuple(stats.inputs_shape) == (32, 32, 13)
            else:
                losses.append(float(tr.batch_training(dict({k: v.cuda() for k, v in host.items()}, num_tokens=n), eng)))
        torch.cuda.synchronize()
        st = model._engine.varlen_status()
        model.check_deferred()
        return losses, st, model._engine.master.detach().cpu().numpy().copy()

    lr_, sr, mr = run(True)
    lc, sc, mc = run(False)
    assert sr == (True, (n + 63) // 64 * 64, False) and sc == sr
    np.testing.assert_allclose(lr_, lc, rtol=2e-6)          # (the loss sum is an fp32-atomic reduction: equal up to its order)
    np.testing.assert_array_equal(mr, mc)


def test_varlen_wrong_count_poisons_loss_and_raises_deferred():
    """a caller's token count that disagrees with the mask cannot pass silently - the step's loss is NaN, the sticky
    device flag survives later (correct) steps until `check_deferred()` raises ValueError, no kernel leaves the rows of the step (the
    samples are cut at the count), and a FAILED forward does not leave its count to the next call.  Labels at padded positions raise
    the same flag."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    cfg = dict(hidden_act="gelu", vocab_size=500, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=64, causal_attention=False, stacked_feat=4, next_n_token=4)
    batch = synth.make_pretrain_batch(B=16, S=32, F=4, V=500, seed=5)
    n = synth.real_tokens(batch)
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=2).cuda().eval()
    good = float(model(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"]).head1_loss)
    assert np.isfinite(good) and model._engine.varlen_status() == (True, (n + 63) // 64 * 64, False)
    model.check_deferred()
    for wrong in (n - 70, n + 5):
        bad = float(model(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"], num_tokens=wrong).head1_loss)
        assert np.isnan(bad)
        again = float(model(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"]).head1_loss)
        assert abs(again - good) <= 2e-6 * abs(good)          # the next, correct step is untouched ...
        with pytest.raises(ValueError):
            model.check_deferred()                             # ... and the flag is still up
        model.check_deferred()                                 # cleared by the read
    # a forward that fails its argument checks consumes the count it was given
    import ctypes as C
    e = model._engine
    ids, att, lab0 = (dev[k].to(torch.int64).contiguous() for k in ("input_ids", "attention_mask", "labels"))
    loss = torch.zeros(1, device="cuda")
    args = lambda B: (e.h, C.c_void_p(ids.data_ptr()), C.c_void_p(att.data_ptr()), C.c_void_p(lab0.data_ptr()), None, None, B, 32,
                      C.c_void_p(loss.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    e.set_token_count(n - 9)
    assert e.lib.gget_forward_pretrain(*args(10 ** 6)) != 0   # exceeds the handle's capacity: fails its argument checks
    assert e.lib.gget_forward_pretrain(*args(16)) == 0
    assert e.varlen_status()[0] is False                      # no stale count: padded layout
    model.check_deferred()
    # labels at padded positions: flagged on the compact layout
    lab = dev["labels"].clone()
    pad = (dev["attention_mask"] == 0).nonzero()[0]
    lab[pad[0], pad[1], 0] = 30
    out = float(model(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=lab).head1_loss)
    assert np.isfinite(out)
    with pytest.raises(ValueError):
        model.check_deferred()


def test_reference_ddp_branch_step_skips_a_non_finite_step():
    """`train_stats.use_deepspeed = False`: the reference's DDP branch (training_utils.py:46-86 - fp16 autocast, GradScaler, clip,
    scaler.step, lr_scheduler.step) on the bf16 engine.  With finite gradients it is the DeepSpeed-branch step (same losses, same weights:
    the loss scale of a bf16 path is 1); with an inf in the gradients GradScaler skips the optimizer step - weights, Adam moments and Adam's
    step count stay, the LR schedule advances - where the DeepSpeed-branch step would have destroyed the weights."""
    import types
    M = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    from src.utils.training_utils import batch_training as ref_batch_training
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
    batch = synth.make_pretrain_batch(B=16, S=32, F=13, V=756, seed=9)
    host = {k: torch.from_numpy(v) for k, v in batch.items() if k != "lengths"}
    host["position_ids"] = torch.arange(32)[None, :].repeat(16, 1)
    tcfg = types.SimpleNamespace(optimizer=types.SimpleNamespace(gradient_accumulation_steps=1, max_grad_norm=1.0))

    def run(use_ds, steps):
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda().eval()
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0, schedule="onecycle", total_num_steps=10, warmup_num_steps=2))
        stats = types.SimpleNamespace(device=torch.device("cuda"), has_embeds_input=False, use_deepspeed=use_ds)
        losses = []
        for _ in range(steps):
            ref_batch_training(host, eng, tcfg, stats, None)
            losses.append(float(stats.loss))
        torch.cuda.synchronize()
        return model, eng, stats, losses

    m_ds, _, _, l_ds = run(True, 3)
    m_dp, eng, stats, l_dp = run(False, 3)
    np.testing.assert_allclose(l_dp, l_ds, rtol=2e-6)
    np.testing.assert_array_equal(m_dp._engine.master.cpu().numpy(), m_ds._engine.master.cpu().numpy())
    assert getattr(eng, "skipped_steps", 0) == 0 and m_dp._engine.step_count == 3 and eng.global_steps == 3
    # a step whose gradients hold an inf: forward + backward by hand, poison one gradient element, then the branch's step rule
    e = m_dp._engine
    before = {k: e.master.clone() for k in ("w",)}
    m0, v0 = e.adam_m.clone(), e.adam_v.clone()
    out = eng(input_ids=host["input_ids"].cuda(), attention_mask=host["attention_mask"].cuda(), labels=host["labels"].cuda())
    real_backward = eng.backward

    def poisoned_backward(loss=None):
        real_backward(loss)
        e.grad_bf16[12345] = float("inf")
    eng.backward = poisoned_backward
    gn = tr._reference_optimizer_step(eng, tcfg, stats, out.head1_loss)
    eng.backward = real_backward
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(gn))
    assert torch.equal(e.master, before["w"]) and torch.equal(e.adam_m, m0) and torch.equal(e.adam_v, v0)
    assert eng.skipped_steps == 1 and e.step_count == 3 and eng.global_steps == 4          # Adam's count stays, the schedule moved on
    # ... and training continues from the untouched state
    ref_batch_training(host, eng, tcfg, stats, None)
    torch.cuda.synchronize()
    assert np.isfinite(float(stats.loss)) and e.step_count == 4 and not torch.equal(e.master, before["w"])
    # the DeepSpeed-branch rule has no such guard (DeepSpeed's bf16 optimizer): the same poisoned step ruins the weights
    e2 = m_ds._engine
    eng2 = tr.initialize(m_ds, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
    out2 = eng2(input_ids=host["input_ids"].cuda(), attention_mask=host["attention_mask"].cuda(), labels=host["labels"].cuda())
    eng2.backward(out2.head1_loss)
    e2.grad_bf16[12345] = float("inf")
    eng2.step()
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(e2.master).all())


def test_model_token_layout_attribute_overrides_the_environment(monkeypatch):
    """`model.token_layout` - the per-model switch beside the process-wide GGET_VARLEN: "padded" keeps the [B,S] grid for a batch that
    would run var-len (and the hidden-state accessors work), "nosync" runs a device mask padded and a host mask var-len, "auto" hands the
    decision back to the environment; unknown names raise."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    cfg = dict(hidden_act="gelu", vocab_size=500, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=4, next_n_token=4)
    batch = synth.make_pretrain_batch(B=8, S=32, F=4, V=500, seed=5)
    host = {k: torch.from_numpy(v) for k, v in batch.items() if k != "lengths"}
    dev = {k: v.cuda() for k, v in host.items()}
    model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda().eval()

    def ran_varlen(data):
        with torch.no_grad():
            loss = float(model(**data).head1_loss)
        return model._engine.varlen_status()[0], loss

    assert model.token_layout == "auto"
    v0, l0 = ran_varlen(dev)
    model.token_layout = "padded"
    v1, l1 = ran_varlen(dev)
    hs = model._engine.hidden_states(8, 32)
    assert tuple(hs.shape) == (8, 32, 128)
    model.token_layout = "nosync"
    v2, _ = ran_varlen(dev)
    v3, l3 = ran_varlen(host)
    monkeypatch.setenv("GGET_VARLEN", "0")
    model.token_layout = "varlen"                           # the attribute wins over the environment
    v4, _ = ran_varlen(dev)
    model.token_layout = "auto"                             # ... and "auto" gives the decision back to it
    v5, _ = ran_varlen(dev)
    assert (v0, v1, v2, v3, v4, v5) == (True, False, False, True, True, False)
    assert abs(l0 - l1) <= 5e-5 * abs(l1) and abs(l3 - l1) <= 5e-5 * abs(l1)
    with pytest.raises(ValueError):
        model.token_layout = "compact"


def test_varlen_full_logit_inference_matches_padded():
    """labels = None (generation, `sample_per_batch`: logits for every cell of the [B,S,F] grid): the layer stack runs on the compact rows,
    the head still writes [B S F, V] logits in cell order - at every REAL position they are the padded run's (the cells of padded positions
    are as meaningless as the reference's in both runs)."""
    spec = _tiny_spec(spec_mod.KIND_PRETRAIN, 32)
    state = weights_mod.make_state_dict(spec, seed=3, std=0.05, head_std=0.1)
    batch = synth.make_pretrain_batch(B=16, S=32, F=4, V=500, seed=21)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    n = int(batch["attention_mask"].sum())
    outs = []
    for cnt in (None, n):
        e = eng_mod.Engine(spec, max_tokens=16 * 32, max_batch=16)
        e.load_state_dict(state)
        e.forward_pretrain(b["input_ids"], b["attention_mask"], None, num_tokens=cnt)
        assert e.varlen_status()[0] == (cnt is not None)
        assert e.deferred_status() == (False, False)          # selecting the cells of padded positions raises no flag in inference
        M, Lm = e.head_counts()
        assert Lm == 16 * 32 * 4
        outs.append(e.head_logits().float().cpu().view(16, 32, 4, -1))
    real = b["attention_mask"].bool()
    a, c = outs[0][real], outs[1][real]
    assert float((a - c).abs().max()) <= 2e-2 * max(1.0, float(a.abs().max())) and rel_l2(c.numpy(), a.numpy()) < 3e-3
    assert bool(torch.isfinite(outs[1]).all())


def test_varlen_token_level_head_matches_padded():
    """loss_type = "token_ce" (node-level tasks: `score` and the cross-entropy on every row, task_logits [B,S,C]) on the var-len layout
    (round 5): the logits of the compact rows are scattered back to [B,S,C] order (padded positions: zeros), the labels are read through the
    compact -> logical row map; loss, logits at the real positions and every gradient equal the padded run's."""
    S, F, V, B, C_ = 40, 4, 500, 12, 5
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=C_)
    state = weights_mod.make_state_dict(spec, seed=6, std=0.05, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=19)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    g = torch.Generator().manual_seed(3)
    y = torch.randint(0, C_, (B, S), generator=g)
    y[torch.rand(B, S, generator=g) < 0.3] = -100
    y[b["attention_mask"] == 0] = -100            # the collator pads labels with -100
    n = int(batch["attention_mask"].sum())
    out = {}
    for lay, cnt in (("padded", None), ("varlen", n)):
        e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
        e.load_state_dict(state)
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], y, None, L.PROBLEM_TOKEN_CE, num_tokens=cnt)
        assert e.varlen_status()[0] == (lay == "varlen")
        e.backward()
        torch.cuda.synchronize()
        out[lay] = (float(loss), logits.float().cpu().clone(), {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()})
    (lp, zp, gp), (lv, zv, gv) = out["padded"], out["varlen"]
    real = b["attention_mask"].bool()
    assert tuple(zv.shape) == (B, S, C_) and abs(lv - lp) <= 2e-5 * abs(lp)
    assert float((zv[real] - zp[real]).abs().max()) <= 2e-3 * max(1.0, float(zp[real].abs().max()))
    assert bool((zv[~real] == 0).all())
    gmax = max(float(np.linalg.norm(x)) for x in gp.values())
    for k in gp:
        assert float(np.linalg.norm(gv[k] - gp[k])) / max(float(np.linalg.norm(gp[k])), 1e-2 * gmax) < 1e-2, k


@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_varlen_raw_embedding_inputs_match_padded(kind):
    """config.embed_dim > 0 (raw per-token embeddings [B,S,E] next to the ids; modeling_pretrain.py:131-149) on the var-len layout
    (round 5): the raw rows and the labels of the mask-token blend are read at the logical row of every compact row, the branch's dropout
    stream is keyed by it - loss and every gradient (incl. embed_proj, embed_layernorm, emb_mask_token) equal the padded run's, with
    embed_dropout on."""
    S, F, V, B, E = 40, 4, 500, 12, 64
    pt = kind == "pt"
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN if pt else spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F,
                                   next_n_token=F if pt else 1, num_labels=2, embed_dim=E, embed_pdrop=0.1)
    state = weights_mod.make_state_dict(spec, seed=8, std=0.05, head_std=0.1)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=23) if pt else synth.make_task_batch(B=B, S=S, F=F, V=V, seed=23)
    b = tb({k: v for k, v in batch.items() if k != "lengths"})
    raw = torch.randn(B, S, E, generator=torch.Generator().manual_seed(5))
    n = int(batch["attention_mask"].sum())
    out = {}
    for lay, cnt in (("padded", None), ("varlen", n)):
        e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
        e.load_state_dict(state)
        e.set_dropout(0.0, 0.0, 91)
        e.set_dropout_ex(0.1, 0.0, 0.0)
        e.set_raw_embeds(raw)
        if pt:
            loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=cnt)
        else:
            loss, _, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                        num_tokens=cnt)
        assert e.varlen_status()[0] == (lay == "varlen")
        e.backward()
        torch.cuda.synchronize()
        out[lay] = (float(loss), {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()})
    (lp, gp), (lv, gv) = out["padded"], out["varlen"]
    assert abs(lv - lp) <= 2e-5 * abs(lp), (lv, lp)
    assert "embed_proj.weight" in gp and float(np.linalg.norm(gp["embed_proj.weight"])) > 0
    gmax = max(float(np.linalg.norm(x)) for x in gp.values())
    for k in gp:
        assert float(np.linalg.norm(gv[k] - gp[k])) / max(float(np.linalg.norm(gp[k])), 1e-2 * gmax) < 1e-2, k


def test_prefetched_batches_give_the_bit_identical_training_run():
    """the batch hand-over inside the product.  DevicePrefetcher copies batch t + 1 through pinned staging buffers on a side
    stream while step t runs and hands the host-side mask sum over as `num_tokens`; the steps must be exactly the steps of the reference's
    synchronous hand-over (every tensor `.to(device)` at the top of the step, training_utils.py:17-26) - same losses, same weights, bit for
    bit - also when the staging buffers are reused (6 batches through 2 sets) and with batches of different widths in one run."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13, attention_dropout=0.1)
    host = [{k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=16, S=S_, F=13, V=756, seed=70 + i).items() if k != "lengths"}
            for i, S_ in enumerate((32, 32, 40, 24, 32, 56))]

    def run(prefetch):
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda()
        model._ensure_engine(16, 56)
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
        losses = []
        if prefetch:
            for data in tr.DevicePrefetcher(host, model.device):
                assert data["input_ids"].is_cuda and isinstance(data["num_tokens"], int)
                losses.append(tr.batch_training(data, eng))
        else:
            for b in host:
                data = {k: v.cuda() for k, v in b.items()}
                data["num_tokens"] = int(b["attention_mask"].sum())
                losses.append(tr.batch_training(data, eng))
        torch.cuda.synchronize()
        return [float(x) for x in losses], model._engine.master.clone(), model._engine.varlen_status()[0]
    l0, w0, v0 = run(False)
    l1, w1, v1 = run(True)
    assert v0 and v1
    assert l0 == l1, (l0, l1)
    assert torch.equal(w0, w1)


def test_counted_token_total_polled_word_equals_copy_and_event():
    """Round 6: a reference-shaped call (device mask, no count) gets the counted total through a store of `sum_lengths_kernel` into a
    pinned host word the host polls; `GGET_COUNT_COPY=1` is the 4-byte device-to-host copy + event of rounds 3 - 5.  The knob is read
    once per process, so the two forms run in two child processes: the same rows, loss bits and master weights after three steps over
    batches whose row count changes from step to step."""
    import json, os, subprocess, sys
    code = r'''
import importlib, json, sys, numpy as np, torch
M = importlib.import_module("graph-gpt_amd.modeling"); tr = importlib.import_module("graph-gpt_amd.training"); synth = importlib.import_module("graph-gpt_amd.synth")
cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
           max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=1).cuda().eval()
eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
out = {"loss": [], "rows": []}
for i in range(3):
    b = synth.make_pretrain_batch(B=32, S=32, F=13, V=756, seed=20 + i)
    dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}     # (device tensors, no num_tokens: the engine counts)
    out["loss"].append(float(tr.batch_training(dev, eng)).hex())
    out["rows"].append(list(model._engine.varlen_status()) + [(synth.real_tokens(b) + 63) // 64 * 64])
torch.cuda.synchronize(); model.check_deferred()
out["master"] = float(np.abs(model._engine.master.detach().cpu().numpy()).sum()).hex()
print("RESULT " + json.dumps(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for knob in ("0", "1"):
        env = dict(os.environ, GGET_COUNT_COPY=knob, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        res.append(json.loads(line[-1][7:]))
    polled, copied = res
    for st in polled["rows"]:
        assert st[0] is True and st[1] == st[3] and st[2] is False        # var-len ran on exactly the counted rows
    assert polled == copied
