"""End-to-end parity of the HIP engine (through the C ABI) with
  (a) the golden vectors captured from the real reference (tests/golden, fp32 and the reference's own bf16 path) and
  (b) the CPU oracle run on the same seeded inputs.
The engine computes in bf16 with fp32 islands, so tolerances are bf16-class and written next to each check;
the yard-stick is how far the reference's own bf16 module path sits from its fp32 path on the same batch."""
import importlib

import numpy as np
import pytest
import torch

from _util import ADAM, BASE_CASES, CLIP, FT_CASES, PT_CASES, ft_problem, load_case, loss_tolerance, record_error, rel_l2, tb
from oracle import gget_oracle as O

pytestmark = pytest.mark.gpu

eng_mod = importlib.import_module("graph-gpt_amd.engine")
L = importlib.import_module("graph-gpt_amd._lib")


def make_engine(spec, batch):
    B, S = batch["input_ids"].shape[:2]
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    return e


def run_forward(e, spec, b, kind, name=""):
    if kind == "pt":
        loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"))
        return loss, None
    pt_, lt_ = ft_problem(spec, b, name)
    problem = {"regression": L.PROBLEM_REGRESSION_L1 if lt_ == "l1" else L.PROBLEM_REGRESSION_MSE,
               "multi_label_classification": L.PROBLEM_MULTI_LABEL, "single_label_classification": L.PROBLEM_SINGLE_LABEL}[pt_]
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], b.get("wgt"), problem)
    return loss, logits


def oracle_fn(spec, b, kind, name=""):
    if kind == "pt":
        return (lambda p: O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"))), \
            "head1_loss", "head1_logits"
    problem, loss_type = ft_problem(spec, b, name)
    return (lambda p: O.task_forward(spec, p, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                     sample_wgt=b.get("wgt"), problem_type=problem, loss_type=loss_type)), "task_loss", "task_logits"


# Loss tolerance (tests/_util.py:loss_tolerance): north_star's 1e-4 relative, or 1.5 x the gap between the reference's OWN
# bf16 and fp32 losses on the case when that is larger.  Two cases get an explicit floor on top of that rule, each with its
# reason; every measured error is recorded next to its tolerance (gpurun_out/parity_errors.json -> profiles/).
LOSS_FLOOR = {
    # 4-sample CE / L1 losses whose reference bf16 path happens to land within 1e-5 of fp32 by cancellation: the rule's
    # tolerance collapses to 1e-4 although every pooled logit carries bf16 noise of ~4e-3 relative
    "ft_tiny_ls": 4e-3, "ft_tiny_reg": 4e-3, "ft_tiny_f4": 2e-3, "ft_tiny_ml": 2e-3,
    # MSE over 32 pooled rows: (pred - y)^2 doubles the relative error of pred (measured max-abs logit error 8e-3, like the
    # other heads), while the reference's bf16 loss lands 5.6e-4 from fp32 by cancellation across the rows
    "ft_tiny_mse": 4e-3,
}
# fine-tune losses are means over B pooled rows only (no averaging over thousands of masked cells): factor 2 instead of 1.5
FT_FACTOR = 2.0


@pytest.mark.parametrize("name", PT_CASES + FT_CASES)
def test_forward_matches_reference(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, tlogits = run_forward(e, spec, b, kind, name)
    torch.cuda.synchronize()
    got = float(loss.item())
    want = float(z["loss"])
    tol = max(loss_tolerance(z, factor=1.5 if kind == "pt" else FT_FACTOR), LOSS_FLOOR.get(name, 0.0))
    record_error(name, "loss_rel_vs_reference_fp32", abs(got - want) / abs(want), tol)
    record_error(name, "reference_bf16_vs_fp32_loss_gap", abs(float(z["loss_bf16"]) - want) / abs(want), float("nan"))
    assert abs(got - want) <= tol * abs(want) + 1e-6, f"{name}: loss {got} vs reference fp32 {want} (bf16 ref {float(z['loss_bf16'])})"
    if kind == "pt":
        M, Lm = e.head_counts()
        assert Lm == int(z["logits_shape"][0])
        lg = e.head_logits().float().cpu().numpy()
        assert lg.shape == tuple(int(x) for x in z["logits_shape"])
        err = rel_l2(lg[:64], z["logits"])
        ref_err = rel_l2(z["logits_bf16"], z["logits"])
        record_error(name, "logits_rel_l2_vs_reference_fp32", err, max(2.5 * ref_err, 1.5e-2))
        assert err < max(2.5 * ref_err, 1.5e-2), f"{name}: logits rel-L2 {err} (reference bf16-vs-fp32 {ref_err})"
    else:
        err = np.abs(tlogits.cpu().numpy()[:64] - z["logits"]).max()
        ref_err = np.abs(z["logits_bf16"] - z["logits"]).max()
        record_error(name, "task_logits_max_abs_vs_reference_fp32", err, max(3 * ref_err, 2e-2))
        assert err < max(3 * ref_err, 2e-2), f"{name}: task logits max-abs {err} (reference bf16-vs-fp32 {ref_err})"


@pytest.mark.parametrize("name", ["pt_tiny_f13_a", "pt_tiny_f1", "pt_tiny_causal", "pt_tiny_gated", "pt_tiny_wgt",
                                  "pt_tiny_bigw", "pt_tiny_s72", "pt_tiny_packed", "ft_tiny_f4", "ft_tiny_ls", "ft_tiny_reg",
                                  "ft_tiny_ml", "ft_tiny_f4_b32", "ft_tiny_mse", "ft_tiny_wce"])
def test_backward_matches_oracle(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    run_forward(e, spec, b, kind, name)
    e.backward()
    torch.cuda.synchronize()
    got = {k: v.float().cpu().numpy() for k, v in e.grads().items()}
    # oracle on the bf16-rounded weights (what the engine computes with), fp32 arithmetic
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, _ = oracle_fn(spec, b, kind, name)
    _, grads = O.loss_and_grads(fn, p, lk)
    worst = []
    gmax = max(float(np.linalg.norm(grads[k].numpy())) for k in state)
    for k in state:
        w = grads[k].numpy()
        if np.linalg.norm(w) < 1e-12:
            assert np.linalg.norm(got[k]) < 1e-6, f"{name}: {k} should have zero gradient"
            continue
        err = rel_l2(got[k], w)
        # tensors whose gradient is <1% of the largest one (q/k projections at N(0,0.02) init: near-uniform
        # softmax) sit at the bf16 noise floor of the signal that feeds them: judge them on the global scale
        if np.linalg.norm(w) < 1e-2 * gmax:
            err = float(np.linalg.norm(got[k] - w)) / (1e-2 * gmax)
        worst.append((err, k))
    worst.sort(reverse=True)
    # bf16 activations/gradients through L layers: a few 1e-2 relative per tensor (measured: <= 2.1e-2 on every fixture, profiles/r06_parity_errors.json; 6e-2 until round 5)
    record_error(name, "worst_gradient_rel_l2_vs_oracle (" + worst[0][1] + ")", worst[0][0], 3e-2)
    assert worst[0][0] < 3e-2, f"{name}: worst gradient rel-L2 {worst[:4]}"
    # q / k projections judged on their OWN norm wherever it is not negligible (big-weight cases: attention far from uniform)
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight",
              "model.layers.1.self_attn.q_proj.weight", "model.layers.1.self_attn.k_proj.weight"):
        w = grads[k].numpy()
        if np.linalg.norm(w) >= 2e-3 * gmax:
            err = rel_l2(got[k], w)
            record_error(name, "grad_rel_l2_own_norm " + k, err, 3e-2)
            assert err < 3e-2, f"{name}: {k} rel-L2 {err} on its own norm ({np.linalg.norm(w) / gmax:.1e} of the largest)"
    # the three tensors the fixtures keep from the REAL reference
    for key, arr in (("model.embed_tokens.weight", "grad_embed"), ("model.layers.0.self_attn.q_proj.weight", "grad_l0_q"),
                     ("model.layers.1.mlp.down_proj.weight", "grad_l1_down")):
        err = rel_l2(got[key], z[arr])
        record_error(name, "grad_rel_l2_vs_reference " + key, err, 3e-2)
        assert err < 3e-2, f"{name}: {key} vs reference gradient rel-L2 {err}"


@pytest.mark.parametrize("name", ["pt_tiny_f13_a", "pt_tiny_bigw", "pt_tiny_wgt", "ft_tiny_f4", "ft_tiny_mse", "ft_tiny_wce"])
def test_adamw_trajectory(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    losses, gns = [], []
    for _ in range(3):
        loss, _ = run_forward(e, spec, b, kind, name)
        losses.append(float(loss.item()))
        e.backward()
        gns.append(float(e.adamw_step(ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"], ADAM["wd"], CLIP).item()))
    loss, _ = run_forward(e, spec, b, kind, name)
    losses.append(float(loss.item()))
    # clip-then-AdamW (training_utils.py:68-80): same loss curve as the reference within bf16 noise
    record_error(name, "adamw_3step_losses_max_rel", float(np.max(np.abs(np.array(losses) - z["adamw_losses"]) / np.maximum(np.abs(z["adamw_losses"]), 2e-3 / (4e-3 if kind == "pt" else 6e-2)))), 4e-3 if kind == "pt" else 6e-2)
    np.testing.assert_allclose(losses, z["adamw_losses"], rtol=4e-3 if kind == "pt" else 6e-2, atol=2e-3)
    # (fine-tune cases overfit the one batch within a step: the later norms belong to losses of 1e-3 and move with bf16 noise)
    np.testing.assert_allclose(gns, z["adamw_gnorms"], rtol=3e-2 if kind == "pt" else 6e-2)
    fin = np.array([float(e.view(k, "master").float().norm()) for k in state])
    np.testing.assert_allclose(fin, z["adamw_final_norms"], rtol=2e-3)


def test_hidden_states_and_padding_invariance():
    """Outputs at real positions do not depend on pad-token content (SURVEY.md 3.3 probe), and the final
    hidden states match the oracle at real positions."""
    z, spec, state, batch = load_case("pt_tiny_f13_a")
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    B, S = b["input_ids"].shape[:2]
    e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"])
    h1 = e.hidden_states(B, S).float().cpu()
    ids2 = b["input_ids"].clone()
    pad = b["attention_mask"] == 0
    ids2[pad] = 37  # garbage under the mask
    e.forward_pretrain(ids2, b["attention_mask"], b["labels"])
    h2 = e.hidden_states(B, S).float().cpu()
    real = ~pad
    assert torch.equal(h1[real], h2[real])
    p = O.to_params(state, torch.float32, requires_grad=False)
    with torch.no_grad():
        out = O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"])
    assert rel_l2(h1[real].numpy(), out["hidden"][real].numpy()) < 1.5e-2


def test_generation_mode_logits_for_every_cell():
    """labels=None => logits for all B*S*F cells (generation_utils.py:117-126 path)."""
    z, spec, state, batch = load_case("pt_tiny_f13_a")
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    B, S, F = b["input_ids"].shape
    assert e.forward_pretrain(b["input_ids"], b["attention_mask"], None) is None
    M, Lm = e.head_counts()
    assert (M, Lm) == (B * S, B * S * F)
    lg = e.head_logits().float().cpu()
    p = O.to_params(state, torch.float32, requires_grad=False)
    with torch.no_grad():
        out = O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], None)
    real = (b["attention_mask"] == 1)[:, :, None].expand(B, S, F).reshape(-1)
    assert rel_l2(lg[real].numpy(), out["head1_logits"][real].numpy()) < 2e-2


@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_medium_batch_matches_oracle(kind):
    """T = B*S = 512 tokens: K of the wgrad GEMMs is a multiple of 64, so the persistent grouped GEMM path, the
    multi-tile-per-block schedule and the counting-sort embedding backward all run; compared with the oracle."""
    from _util import spec_mod, weights_mod, synth
    B, S = 16, 32
    if kind == "pt":
        spec = spec_mod.spec_from_size("tiny", vocab_size=756, stacked_feat=13, next_n_token=13)
        batch = synth.make_pretrain_batch(B=B, S=S, F=13, V=756, seed=21)
    else:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2)
        batch = synth.make_task_batch(B=B, S=S, F=4, V=1000, seed=22)
    state = weights_mod.make_state_dict(spec, seed=5, std=0.04, head_std=0.08)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, _ = run_forward(e, spec, b, kind)
    e.backward()
    torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, _ = oracle_fn(spec, b, kind)
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(loss.item() - out[lk].item()) <= (3e-4 if kind == "pt" else 2e-2) * abs(out[lk].item())
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in state:
        w = grads[k].numpy()
        gk = got[k].float().cpu().numpy()
        err = float(np.linalg.norm(gk - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        assert err < 3e-2, f"{k}: {err}"


def test_staged_backward_equals_monolithic():
    """The DP path drives backward in L+2 stages (one gradient bucket each, all-reduced on a side stream); at world
    size 1 the exchange is the identity, so the staged schedule must reproduce the one-shot backward."""
    tr = importlib.import_module("graph-gpt_amd.training")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    from _util import synth
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                  num_attention_heads=2, max_position_embeddings=1024, causal_attention=False,
                                  stacked_feat=13, next_n_token=13)
    batch = synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=3)
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    grads = []
    for staged in (False, True):
        model = modeling.GraphGPTPretrainBase(cfg, seed=1)
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3))
        eng.force_staged = staged
        out = eng(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"])
        eng.backward(out.head1_loss)
        torch.cuda.synchronize()
        grads.append(model._engine.grad_bf16.clone())
        eng.step()
        torch.cuda.synchronize()
    # not bitwise: the fp32-atomic reductions (norm weights, embedding rows) are order-dependent between any two runs
    a, b = grads[0].float(), grads[1].float()
    assert float((a - b).norm() / b.norm()) < 2e-3
    # buckets tile the flat gradient array exactly once
    e = model._engine
    cover = torch.zeros(e.n_params, dtype=torch.int32)
    for off, cnt in e.buckets:
        cover[off: off + cnt] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1


def _path_keep(seed, layer, which, B, rate):
    """Python twin of path_keep() in csrc/engine.hip."""
    if rate <= 0:
        return torch.ones(B)
    M32 = 0xFFFFFFFF
    s = (seed ^ ((0xD6E8FEB8 * (layer * 2 + which + 1)) & M32)) & M32
    out = []
    for b in range(B):
        x = (s + b * 0x85EBCA77) & M32
        x ^= x >> 16; x = (x * 0x7FEB352D) & M32
        x ^= x >> 15; x = (x * 0x846CA68B) & M32
        x ^= x >> 16
        u = np.float32(x >> 8) * np.float32(1.0 / 16777216.0)
        out.append(0.0 if u < np.float32(rate) else float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))))
    return torch.tensor(out)


@pytest.mark.parametrize("layer_scale", [0.0, 1.0])
def test_drop_path_matches_oracle_with_same_mask(layer_scale):
    """Stochastic depth (utils_graphgpt.py:64-66,184: per-sample DropPath, rate linspace(0, path_pdrop, L)) as used by
    the ogbl-ppa fine-tune config (path 0.2 + LayerScale 1): the engine's counter-based per-sample mask is rebuilt in
    Python and fed to the oracle, forward loss and every gradient must agree."""
    from _util import spec_mod, weights_mod, synth
    B, S, seed, rate = 16, 24, 777, 0.5
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=1000, stacked_feat=4, next_n_token=1,
                                   num_labels=2, layer_scale_init=layer_scale, path_pdrop=rate)
    state = weights_mod.make_state_dict(spec, seed=9, std=0.05, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=4, V=1000, seed=33)
    b = tb(batch)
    e = eng_mod.Engine(spec, B * S, B)
    e.load_state_dict(state)
    e.set_dropout(0.0, rate, seed)
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None,
                                     L.PROBLEM_SINGLE_LABEL)
    e.backward()
    torch.cuda.synchronize()
    L_ = spec.num_layers
    pm = lambda l, w: _path_keep(seed, l, w, B, rate * l / (L_ - 1))
    assert any(float(pm(L_ - 1, w).min()) == 0.0 for w in (0, 1)), "mask should drop something at rate 0.5"
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    out, grads = O.loss_and_grads(lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"],
                                                           b["task_labels"], path_mult=pm), p, "task_loss")
    assert abs(loss.item() - out["task_loss"].item()) <= 2e-2 * abs(out["task_loss"].item())
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in state:
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        assert err < 3e-2, f"{k}: {err}"


@pytest.mark.gpu
def test_smtp_inside_forward_matches_explicit_masking():
    """config.smtp_inside (reference modeling_pretrain.py:175-189): the model masks on the device; its loss equals the
    loss of a plain model fed with the oracle's masked ids / labels for the same draws."""
    import importlib
    from oracle import gget_oracle as O
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    smtp = importlib.import_module("graph-gpt_amd.smtp")
    F, V, B, S = 4, 211, 6, 24
    kw = dict(vocab_size=V, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
              max_position_embeddings=64, causal_attention=False, stacked_feat=F, next_n_token=F)
    m_in = modeling.GraphGPTPretrainBase(modeling.GraphGPTConfig(hidden_act="gelu", smtp_inside=True, smtp_power=1.0, **kw), seed=4).cuda().eval()
    m_ex = modeling.GraphGPTPretrainBase(modeling.GraphGPTConfig(hidden_act="gelu", **kw), seed=4).cuda().eval()
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(10, S + 1, (B,), generator=g)
    full = torch.zeros(B, S, F + 4, dtype=torch.int64)
    full[:, :, :F] = torch.randint(2, V, (B, S, F), generator=g)
    att = torch.zeros(B, S, dtype=torch.int64)
    for b in range(B):
        full[b, lens[b]:, :F] = 0
        att[b, : lens[b]] = 1
        full[b, : lens[b], F + 2] = torch.randint(0, int(lens[b]) // 2 + 1, (int(lens[b]),), generator=g)
    out_in = m_in(input_ids=full.cuda(), attention_mask=att.cuda())
    seed = (m_in.dropout_seed * 0x9E3779B1 + 1 * 0x85EBCA77) & 0xFFFFFFFF
    us, ur, uc, sh, urep = smtp.draws(seed, B, S, F)
    ids, lab = O.smtp_2d_inputs_labels(full[:, :, :F].contiguous(), full[:, :, F + 2].contiguous(), us, ur, uc, sh, urep,
                                       smtp_2d_rate=1.0, power=1.0, replace_rate=0.0, vocab=V)
    out_ex = m_ex(input_ids=ids.cuda(), attention_mask=att.cuda(), labels=lab.cuda())
    assert (lab != -100).any()
    assert float(out_in.head1_loss.detach()) == pytest.approx(float(out_ex.head1_loss.detach()), rel=1e-6)


GEN_CASES = {
    "maskgit_plus": dict(alg="maskgit_plus"),
    "topk_margin": dict(alg="topk_margin"),
    "entropy": dict(alg="entropy"),
    "origin_sampled": dict(alg="origin", temperature=0.8, top_p=0.9, top_k=20, seed=11),
    "gumbel_ranked": dict(alg="maskgit_plus", temperature=0.5, top_k=30, alg_temp=0.4, seed=12),
    "margin_sampled": dict(alg="topk_margin", temperature=1.0, top_p=0.95, seed=13),
}


@pytest.mark.parametrize("case", sorted(GEN_CASES))
def test_generation_loop(case):
    """sample_per_batch on the engine (HIP forward + sampling kernel + update) against the oracle's loop driven by the SAME
    per-iteration kernel outputs (confidence, candidates) and, for "origin", the Python twin of the transfer-mask draws: the
    token grid must be IDENTICAL after every iteration, for the deterministic algorithms and for the stochastic settings
    (sampled candidates, top-p / top-k, Gumbel-perturbed ranking).  The deterministic algorithms are also compared with the
    reference fixture (fp32 model) as a sanity bound."""
    import importlib
    import os
    gen = importlib.import_module("graph-gpt_amd.generation")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    weights = importlib.import_module("graph-gpt_amd.weights")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation.npz"))
    std, head_std, seed = g["meta_init"]
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                  num_attention_heads=2, max_position_embeddings=1024, causal_attention=False, stacked_feat=4,
                                  next_n_token=4)
    model = modeling.GraphGPTPretrainBase(cfg, seed=0).cuda()
    sd = weights.make_state_dict(model.spec, seed=int(seed), std=float(std), head_std=float(head_std))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ids, att = torch.from_numpy(g["in_input_ids"]), torch.from_numpy(g["in_attention_mask"])
    kw = GEN_CASES[case]
    gcfg = gen.GenerationConfig(steps=6, eps=1e-3, mask_token_id=1, output_history=True, **kw)
    x, hist = gen.sample_per_batch(model, gcfg, input_ids=ids, attention_mask=att)
    B, N = 3, ids.shape[1] * ids.shape[2]

    def logits_fn(t):
        model(input_ids=t.cuda(), attention_mask=att.cuda(), labels=None)       # logits stay in the engine's workspace
        return torch.zeros(B * N, 1)                                             # the oracle loop gets them through conf_fn

    def conf_fn(it, _logits):
        c, t = gen.token_sample(model._engine, B * N, gcfg, gen.iteration_seed(gcfg.seed, it), gumbel=gcfg.alg != "origin")
        return c.view(B, N).cpu(), t.view(B, N).cpu()

    def draw_fn(it):
        return {"u_transfer": gen.draws(gcfg.seed, it, B, N)[2]}

    xo, ho = O.sample_per_batch(logits_fn, ids, alg=gcfg.alg, steps=6, eps=1e-3, mask_token_id=1, conf_fn=conf_fn,
                                draw_fn=draw_fn, surplus="skip")
    assert len(hist) == len(ho) and len(hist) >= 3
    for it, (a, b) in enumerate(zip(hist, ho)):
        assert torch.equal(a.cpu(), b), f"{case}: token grid differs after iteration {it}"
    assert torch.equal(x.cpu(), xo)
    start = ids.view(3, -1) == 1
    assert (start & (x.cpu() != 1)).any()
    assert not ((~start) & (x.cpu() != ids.view(3, -1))).any(), "revealed / given tokens must never be rewritten"
    if case in ("maskgit_plus", "topk_margin", "entropy"):
        ref = torch.from_numpy(g[f"{case}_x"])
        revealed = start & (ref != 1) & (x.cpu() != 1)
        agree = (x.cpu()[revealed] == ref[revealed]).float().mean().item()
        # a random-init model decodes near-uniform distributions: trajectories that part once (bf16 vs fp32 near-ties) keep
        # parting, so this is only a sanity bound; parity proper = oracle == reference (CPU tests) + engine == oracle above
        assert agree > 0.3, f"only {agree:.2f} of the revealed tokens match the fp32 reference run"
    with pytest.raises(ValueError):
        gen.GenerationConfig(alg="no_such_alg").validate()


def test_reproducible_mode_gives_bit_identical_parameters():
    """gget_debug_set(4, 1): the RMSNorm weight gradients - the one sum of the pre-train gradient path that is added with fp32 atomics -
    are summed in block order.  Two engines started from the same state then hold BIT-IDENTICAL fp32 master parameters and Adam moments
    after several clip + AdamW steps on the var-len layout (attention dropout on), and the same loss to the last bits of its own
    (atomic) sum; the default mode must stay within tolerance of it."""
    from _util import spec_mod, weights_mod, synth
    LL = L
    lib = LL.load()
    B, S, F, V = 96, 32, 13, 756
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=V, hidden_size=768, intermediate_size=3072, num_layers=2,
                              num_heads=12, head_dim=64, stacked_feat=F, next_n_token=F)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=5)
    state = weights_mod.make_state_dict(spec, seed=3, std=0.03, head_std=0.05)
    b = tb(batch)
    n_real = int(batch["attention_mask"].sum())

    def run(det):
        LL.check(lib.gget_debug_set(4, det))
        e = make_engine(spec, batch)
        e.load_state_dict(state)
        e.set_dropout(0.1, 0.0, 1234)
        losses = []
        for _ in range(4):
            losses.append(e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n_real).item())
            e.backward()
            e.adamw_step(1e-3, max_grad_norm=1.0)
        torch.cuda.synchronize()
        out = (e.master.clone(), e.adam_m.clone(), e.adam_v.clone(), losses)
        del e
        return out
    try:
        a = run(1)
        c = run(1)
        d = run(0)
    finally:
        LL.check(lib.gget_debug_set(4, 0))
    for k in range(3):
        assert torch.equal(a[k], c[k]), ("master", "adam_m", "adam_v")[k] + " differs between two reproducible-mode runs"
    assert a[3] == pytest.approx(c[3], rel=1e-5)      # (the REPORTED loss is still an atomic sum over blocks: equal to its last bits only)
    assert a[3] == pytest.approx(d[3], rel=2e-4), "default mode drifted from the reproducible one"
    assert float((a[0] - d[0]).norm()) <= 3e-2 * float((a[0] - run_init_master(spec, state, batch)).norm() + 1e-12)


def run_init_master(spec, state, batch):
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    m = e.master.clone()
    del e
    return m


@pytest.mark.parametrize("alg", ["maskgit_plus", "topk_margin", "entropy"])
def test_generation_single_iterations_match_reference_fixture(alg):
    """Teacher-forced comparison with the REFERENCE's recorded run (tests/golden/generation.npz: sample_per_batch of
    src/utils/generation_utils.py:22-237 on the fp32 model, the token grid after every iteration): every iteration is restarted
    from the reference's own grid, so one bf16 near-tie cannot compound into a different trajectory as it does in the free-running
    loop (test_generation_loop's "> 0.3 of the revealed tokens" bound).  One engine iteration = HIP forward + sampling kernel +
    ranking update; it must reveal the same cells with the same tokens as the reference did in that iteration, up to near-ties of
    a random-init model's almost flat distributions (measured: 0.986 / 0.950 / 0.950 of the reference's reveals reproduced, cell and
    token, for maskgit_plus / topk_margin / entropy; recorded in profiles/r03_parity_errors.json)."""
    import importlib
    import os
    from _util import record_error
    gen = importlib.import_module("graph-gpt_amd.generation")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    weights = importlib.import_module("graph-gpt_amd.weights")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation.npz"))
    std, head_std, seed = g["meta_init"]
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                  num_attention_heads=2, max_position_embeddings=1024, causal_attention=False, stacked_feat=4,
                                  next_n_token=4)
    model = modeling.GraphGPTPretrainBase(cfg, seed=0).cuda()
    sd = weights.make_state_dict(model.spec, seed=int(seed), std=float(std), head_std=float(head_std))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    ids, att = torch.from_numpy(g["in_input_ids"]).cuda(), torch.from_numpy(g["in_attention_mask"]).cuda()
    hist = torch.from_numpy(g[f"{alg}_hist"]).cuda()              # [iterations, B, S, F]
    B, S, F = ids.shape
    gcfg = gen.GenerationConfig(alg=alg, steps=6, eps=1e-3, mask_token_id=1)
    n_steps = min(int((ids.view(B, -1) == 1).sum(dim=-1).max().item()), gcfg.steps)
    timesteps = torch.linspace(1, gcfg.eps, n_steps + 1, device="cuda")
    i, same_cells, same_tokens, total = 0, 0, 0, 0
    per_it = []
    for it in range(hist.shape[0]):
        x_in = (ids if it == 0 else hist[it - 1]).reshape(B, S * F).clone()
        ref_out = hist[it].reshape(B, S * F)
        with torch.no_grad():
            model(input_ids=x_in.view(B, S, F), attention_mask=att, labels=None)
        conf, cand = gen.token_sample(model._engine, B * S * F, gcfg, gen.iteration_seed(gcfg.seed, it), gumbel=True)
        x_out, i = gen.unmask_step(x_in, conf.view(B, S * F), cand.view(B, S * F), timesteps, i, gcfg)
        # cells the reference revealed in this iteration (a masked cell that now holds a token)
        ref_new = (x_in == 1) & (ref_out != 1)
        got_new = (x_in == 1) & (x_out != 1)
        n = int(ref_new.sum())
        if n == 0:
            continue
        cells = int((ref_new & got_new).sum())
        toks = int((ref_new & got_new & (x_out == ref_out)).sum())
        per_it.append((n, cells / n, toks / n))
        same_cells += cells; same_tokens += toks; total += n
        assert int(got_new.sum()) <= n, f"{alg} iteration {it}: revealed {int(got_new.sum())} cells, reference {n}"
    assert total > 60
    record_error(f"generation_teacher_forced/{alg}", "share of the reference's per-iteration reveals not reproduced (cell and token)",
                 1.0 - same_tokens / total, 0.10)
    assert same_cells / total >= 0.90, f"{alg}: only {same_cells / total:.3f} of the reference's reveals fall on the same cells ({per_it})"
    assert same_tokens / total >= 0.90, f"{alg}: only {same_tokens / total:.3f} of the reference's reveals carry the same token ({per_it})"


def test_base_width_two_layers_matches_oracle():
    """d = 768 (the headline model's width, 2 layers): the shapes that pick the 128x192 / 192x192 / 256x256 tiles, the
    interleaved RoPE epilogue, the grouped four-problem weight-gradient launch and whole-line C stores, end to end against
    the oracle (loss and every gradient tensor)."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 64, 32, 13, 756
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=V, hidden_size=768, intermediate_size=3072, num_layers=2,
                              num_heads=12, head_dim=64, stacked_feat=F, next_n_token=F)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=31)
    state = weights_mod.make_state_dict(spec, seed=6, std=0.03, head_std=0.05)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, _ = run_forward(e, spec, b, "pt")
    e.backward()
    torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, _ = oracle_fn(spec, b, "pt")
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(loss.item() - out[lk].item()) <= 3e-4 * abs(out[lk].item())
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in state:
        w = grads[k].numpy()
        gk = got[k].float().cpu().numpy()
        err = float(np.linalg.norm(gk - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        assert err < 3e-2, f"{k}: {err}"


def test_degenerate_batches():
    """Edge cases of the SMTP head's device-side compaction: no masked cell at all (M = Lm = 0: nothing to back-propagate,
    loss 0 where the reference's mean over an empty set is NaN), one masked cell in a B=1 batch, rows of length 1."""
    from _util import spec_mod, weights_mod, synth
    spec = spec_mod.spec_from_size("tiny", vocab_size=756, stacked_feat=13, next_n_token=13)
    state = weights_mod.make_state_dict(spec, seed=5)
    b = tb(synth.make_pretrain_batch(B=2, S=24, F=13, V=756, seed=3))
    e = eng_mod.Engine(spec, max_tokens=48, max_batch=2)
    e.load_state_dict(state)
    loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], torch.full_like(b["labels"], -100))
    e.backward()
    assert float(loss) == 0.0 and e.head_counts() == (0, 0)
    assert all(float(g.float().abs().max()) == 0.0 for g in e.grads().values())
    b1 = tb(synth.make_pretrain_batch(B=1, S=8, F=13, V=756, seed=4))
    ids, lab = b1["input_ids"].clone(), torch.full_like(b1["labels"], -100)
    lab[0, 2, 5], ids[0, 2, 5] = 77, 1
    e1 = eng_mod.Engine(spec, max_tokens=8, max_batch=1)
    e1.load_state_dict(state)
    loss = e1.forward_pretrain(ids, b1["attention_mask"], lab)
    e1.backward()
    e1.adamw_step(1e-3)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    want = O.pretrain_forward(spec, O.to_params(st_bf, torch.float32, requires_grad=False), ids, b1["attention_mask"], lab)["head1_loss"]
    assert e1.head_counts() == (1, 1) and abs(float(loss) - float(want)) < 2e-3 * abs(float(want))
    att = torch.zeros(2, 24, dtype=torch.int64)
    att[:, 0] = 1
    ids, lab = b["input_ids"].clone(), torch.full_like(b["labels"], -100)
    ids[:, 1:] = 0
    lab[:, 0, 0] = ids[:, 0, 0]
    loss = e.forward_pretrain(ids, att, lab)
    e.backward()
    assert np.isfinite(float(loss)) and all(torch.isfinite(g.float()).all().item() for g in e.grads().values())


def test_head_counts_on_device_persistent_tiles():
    """Sizes whose head GEMMs take the persistent kernel with device-side row / reduction counts (T % 64 == 0): an empty
    selection (M = K = 0 -> zero gradients), a single masked cell and a normal batch, each against the oracle; the normal
    batch twice in a row so that stale rows of the longer selection sit behind the shorter one."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 8, 32, 13, 756
    spec = spec_mod.spec_from_size("tiny", vocab_size=V, stacked_feat=F, next_n_token=F)
    state = weights_mod.make_state_dict(spec, seed=9, std=0.05)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    b = tb(synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=12))
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)

    def check(ids, lab, tol=6e-2):
        loss = e.forward_pretrain(ids, b["attention_mask"], lab)
        e.backward()
        p = O.to_params(st_bf, torch.float32)
        out, grads = O.loss_and_grads(lambda q: O.pretrain_forward(spec, q, ids, b["attention_mask"], lab), p, "head1_loss")
        assert abs(float(loss) - out["head1_loss"].item()) <= 2e-3 * abs(out["head1_loss"].item())
        got = e.grads()
        gmax = max(float(g.norm()) for g in grads.values())
        for k in ("lm_head.weight", "n_token_proj.weight", "model.layers.0.mlp.down_proj.weight", "model.embed_tokens.weight"):
            w = grads[k].numpy()
            err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
            assert err < tol, f"{k}: {err}"

    check(b["input_ids"], b["labels"])
    ids, lab = b["input_ids"].clone(), torch.full_like(b["labels"], -100)
    lab[3, 5, 2], ids[3, 5, 2] = 77, 1
    check(ids, lab)
    loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], torch.full_like(b["labels"], -100))
    e.backward()
    assert float(loss) == 0.0 and e.head_counts() == (0, 0)
    assert all(float(g.float().abs().max()) == 0.0 for g in e.grads().values())
    check(b["input_ids"], b["labels"])


def test_evaluate_pass_matches_oracle_losses():
    """training.evaluate (reference log_eval_dump_utils.evaluate :242-304) on the device model: the mean of the per-batch
    losses of an eval-mode forward (attention dropout configured but inactive) equals the oracle's, and the model returns to
    train mode."""
    import importlib
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    from _util import synth
    F, V = 4, 300
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                  num_attention_heads=2, max_position_embeddings=1024, causal_attention=False, stacked_feat=F,
                                  next_n_token=F, attention_dropout=0.1)
    model = modeling.GraphGPTPretrainBase(cfg, seed=2).cuda()
    loader = []
    for s in (1, 2, 3):
        b = synth.make_pretrain_batch(B=4, S=24, F=F, V=V, seed=40 + s)
        loader.append({k: torch.from_numpy(b[k]) for k in ("input_ids", "attention_mask", "labels", "position_ids")})
    loss, aux = tr.evaluate(model, loader, "valid")
    assert aux is None and model.training
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in sd.items()}
    p = O.to_params(st_bf, torch.float32, requires_grad=False)
    want = np.mean([O.pretrain_forward(model.spec, p, d["input_ids"], d["attention_mask"], d["labels"])["head1_loss"].item()
                    for d in loader])
    assert abs(float(loss) - want) <= 2e-3 * abs(want)


def _c1_engine_and_batch(seed=77, size="base"):
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 256, 32, 13, 756
    spec = spec_mod.spec_from_size(size, kind=spec_mod.KIND_PRETRAIN, vocab_size=V, stacked_feat=F, next_n_token=F)
    state = weights_mod.make_state_dict(spec, seed=3)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=seed)
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)
    return spec, state, batch, e


@pytest.mark.parametrize("layout", ["padded", "varlen"])
@pytest.mark.parametrize("size", ["base", "base24"])
def test_c1_full_size_loss_matches_oracle(size, layout):
    """BASELINE configs[1] (base d768/L12) and configs[2] (base24, 24 layers) at full size (B=256, S=32, F=13, V=756): SMTP loss of the HIP forward against the
    oracle forward on the same bf16-rounded weights, held to 5e-5 relative - inside north_star's 1e-4 and ~10 x what four rounds have
    measured (2e-6 ... 4e-6: a mean over ~37 k masked cells averages the bf16 rounding away) - on the padded grid and on the var-len
    token layout the bench runs (include/gget.h: gget_set_token_count)."""
    spec, state, batch, e = _c1_engine_and_batch(size=size)
    b = tb(batch)
    n_tok = int(batch["attention_mask"].sum()) if layout == "varlen" else None
    loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"), num_tokens=n_tok)
    assert e.varlen_status()[0] == (layout == "varlen")
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32, requires_grad=False)
    with torch.no_grad():
        want = O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"])["head1_loss"].item()
    record_error(f"c1_full_size_{size}_{layout}", "loss_rel_vs_oracle (B=256, S=32)", abs(float(loss) - want) / abs(want), 5e-5)
    assert abs(float(loss) - want) <= 5e-5 * abs(want), (float(loss), want)


@pytest.mark.parametrize("size,layout", [("base", "padded"), ("base", "varlen"), ("base24", "varlen")])
def test_c1_full_size_backward_matches_oracle(size, layout):
    """The headline configuration at FULL size (base d768 / L12, B = 256, S = 32, F = 13, V = 756: one oracle fwd + bwd, ~6 s on the
    box's host cores): loss and the gradients of eleven tensors spread over the stack - embedding, first / middle / last layer
    attention and MLP projections, a norm weight, n_token_proj and lm_head - against the oracle, on both token layouts; and
    BASELINE configs[2] (base24: the same width, 24 layers) on the layout the bench runs.  Tolerances: loss 5e-5 (measured 2.5e-6),
    gradients 3e-2 rel-L2 (measured 4e-3 ... 1.4e-2 on the 12-layer model)."""
    spec, state, batch, e = _c1_engine_and_batch(size=size, seed=79)
    Lm, Ll = spec.num_layers // 2 - 1, spec.num_layers - 1
    b = tb(batch)
    n_tok = int(batch["attention_mask"].sum()) if layout == "varlen" else None
    loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"), num_tokens=n_tok)
    e.backward()
    torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, _ = oracle_fn(spec, b, "pt")
    out, grads = O.loss_and_grads(fn, p, lk)
    want = out[lk].item()
    tag = f"c1_full_size_backward_{layout}" if size == "base" else f"c2_full_size_backward_{size}_{layout}"
    record_error(tag, "loss_rel_vs_oracle", abs(float(loss) - want) / abs(want), 5e-5)
    assert abs(float(loss) - want) <= 5e-5 * abs(want), (float(loss), want)
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.v_proj.weight",
              "model.layers.0.mlp.gate_proj.weight", f"model.layers.{Lm}.self_attn.o_proj.weight", f"model.layers.{Lm}.mlp.down_proj.weight",
              f"model.layers.{Lm}.post_attention_layernorm.weight", f"model.layers.{Ll}.self_attn.k_proj.weight",
              f"model.layers.{Ll}.mlp.up_proj.weight", "n_token_proj.weight", "lm_head.weight"):
        w = grads[k].numpy()
        # (standard init: q / k gradients are ~1e-3 of the largest - near-uniform attention - and sit at the bf16 noise floor of the
        #  signal that feeds them: tensors below 1 % of the largest norm are judged on that scale, like test_backward_matches_oracle)
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error(tag, "grad_rel_l2 " + k, err, 3e-2)
        assert err < 3e-2, f"{k}: {err}"


def test_c1_full_size_properties():
    """Size-independent properties at the full C1 size, no oracle needed: (1) permuting the samples of the batch leaves the
    loss and the gradients unchanged up to summation order; (2) appending padded positions (S 32 -> 40) changes nothing;
    (3) running the same batch again reproduces loss and gradients up to the order of the fp32-atomic reductions."""
    spec, state, batch, e = _c1_engine_and_batch(seed=78)
    b = tb(batch)
    loss0, _ = run_forward(e, spec, b, "pt")
    loss0 = float(loss0)
    e.backward()
    g0 = e.grad_bf16.float().clone()
    # (1) batch permutation
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(5))
    bp = {k: v[perm].contiguous() for k, v in b.items() if torch.is_tensor(v) and v.shape[:1] == (256,)}
    loss1, _ = run_forward(e, spec, bp, "pt")
    e.backward()
    g1 = e.grad_bf16.float()
    assert abs(float(loss1) - loss0) <= 2e-5 * abs(loss0)
    assert float((g1 - g0).norm() / g0.norm()) < 2e-2      # bf16 gradients, different accumulation order
    # (2) padding extension: same graphs in a wider (S = 40) batch
    S2 = 40
    ids = torch.zeros(256, S2, 13, dtype=torch.int64)
    ids[:, :32] = b["input_ids"]
    lab = torch.full((256, S2, 13), -100, dtype=torch.int64)
    lab[:, :32] = b["labels"]
    att = torch.zeros(256, S2, dtype=torch.int64)
    att[:, :32] = b["attention_mask"]
    e2 = eng_mod.Engine(spec, max_tokens=256 * S2, max_batch=256)
    e2.load_state_dict(state)
    loss2 = float(e2.forward_pretrain(ids, att, lab))
    assert abs(loss2 - loss0) <= 2e-5 * abs(loss0)    # mean over the labelled cells: independent of the padded width
    del e2
    # (3) idempotence: the same batch again gives the same loss (no dropout) and gradients up to the order of the few
    # fp32-atomic reductions (loss sum, norm weights)
    loss3, _ = run_forward(e, spec, b, "pt")
    e.backward()
    assert abs(float(loss3) - loss0) <= 2e-6 * abs(loss0)   # the loss sum itself is an fp32-atomic reduction over blocks
    assert float((e.grad_bf16.float() - g0).norm() / g0.norm()) < 1e-3


def _ft_loss_tol(fixture):
    """Loss tolerance of the full-width fine-tune checks against the ORACLE: max(1e-4, 3 x the gap between the reference's own bf16
    and fp32 losses on the reference fixture of the same architecture and sequence length (tools/make_golden.py ft_base_*))."""
    z = np.load(__import__("os").path.join(__import__("_util").GOLDEN, fixture + ".npz"))
    # factor 3, not the fixtures' 2: these checks run OTHER batches (B = 1 ... 8 pooled rows) than the one the reference's gap was
    # measured on, and the loss error of a bf16 path over a handful of 2-class samples scatters by that much from batch to batch
    # (measured here: 2.1e-3 on the B = 8 batch against the fixture's 1.05e-3 gap) -> 3.2e-3 for C3, 2.2e-2 for C4 (round 2: 3e-2 both)
    return loss_tolerance(z, factor=3.0)


def test_c4_long_sequence_full_model_matches_oracle():
    """The longest configuration of BASELINE.json (C4: base model, S = 2048, F = 4, V = 41245) at a batch the CPU oracle
    finishes in seconds (B = 2): fine-tune loss and logits of the full 12-layer HIP forward (4-wave attention kernels over 64
    key tiles, wide-vocabulary embedding path, position_ids up to 2047) against the oracle."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 2, 2048, 4, 41245
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                   max_position=2048)
    state = weights_mod.make_state_dict(spec, seed=8)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=91, lengths="uniform", min_len=S // 2)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, logits = run_forward(e, spec, b, "ft")
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32, requires_grad=False)
    fn, lk, gk = oracle_fn(spec, b, "ft")
    with torch.no_grad():
        out = fn(p)
    want = out[gk].float().numpy()
    got = logits.float().cpu().numpy()
    ltol = _ft_loss_tol("ft_base_s2048")
    record_error("c4_S2048_B2_forward", "loss_rel_vs_oracle", abs(float(loss) - out[lk].item()) / max(abs(out[lk].item()), 0.1), ltol)
    assert np.abs(got - want).max() <= 3e-2 * max(1.0, np.abs(want).max()), (got, want)
    assert abs(float(loss) - out[lk].item()) <= ltol * max(abs(out[lk].item()), 0.1)


def test_c3_shape_full_model_backward_matches_oracle():
    """C3 of BASELINE.json (ogbl-ppa-like fine-tune: base model with LayerScale, S = 256, F = 4, V = 41245) at B = 8: loss and
    gradients of the full 12-layer HIP forward + backward (4-wave attention, LayerScale residual kernels, sorted embedding
    backward for the wide vocabulary, pooled-row score head) against the oracle."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 8, 256, 4, 41245
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                   max_position=1024, layer_scale_init=1.0)
    state = weights_mod.make_state_dict(spec, seed=9)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=92, lengths="uniform", min_len=S // 4)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, _ = run_forward(e, spec, b, "ft")
    e.backward()
    torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, _ = oracle_fn(spec, b, "ft")
    out, grads = O.loss_and_grads(fn, p, lk)
    ltol = _ft_loss_tol("ft_base_ls_s256")
    record_error("c3_S256_B8_backward", "loss_rel_vs_oracle", abs(float(loss) - out[lk].item()) / max(abs(out[lk].item()), 0.1), ltol)
    assert abs(float(loss) - out[lk].item()) <= ltol * max(abs(out[lk].item()), 0.1)
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("score.weight", "model.layers.11.mlp.down_proj.weight", "model.layers.6.self_attn.q_proj.weight",
              "model.layers.0.lambda_1", "model.layers.0.input_layernorm.weight", "model.embed_tokens.weight"):
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        assert err < 8e-2, f"{k}: {err}"


def test_module_moves_after_engine_and_position_guard():
    """once the engine exists the parameters are views of its master arena - `.cuda()` / `.float()` are no-ops,
    `.cpu()` / `.half()` raise instead of silently detaching them; position_ids beyond the RoPE table raise (the reference
    evaluates the rotary embedding on the fly and would accept them - DESIGN.md section 7)."""
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=1,
                                  num_attention_heads=2, max_position_embeddings=64, causal_attention=False, stacked_feat=4,
                                  next_n_token=1, num_labels=2, problem_type="single_label_classification")
    m = modeling.GraphGPTTaskModel(cfg, seed=0).cuda()
    ptr = m._flat["model.embed_tokens.weight"].data_ptr()
    assert m.cuda() is m and m.float() is m and m.to(torch.device("cuda")) is m
    assert m._flat["model.embed_tokens.weight"].data_ptr() == ptr
    for bad in (lambda: m.cpu(), lambda: m.half(), lambda: m.to(torch.bfloat16)):
        with pytest.raises(RuntimeError):
            bad()
    ids = torch.randint(22, 300, (2, 16, 4))
    att = torch.ones(2, 16, dtype=torch.int64)
    pos = torch.arange(16)[None].repeat(2, 1)
    y = torch.tensor([0, 1])
    out = m(input_ids=ids, attention_mask=att, position_ids=pos, task_labels=y)
    assert torch.isfinite(out.task_loss)
    with pytest.raises(IndexError):
        m(input_ids=ids, attention_mask=att, position_ids=pos + 60, task_labels=y)        # host tensor: checked on the spot, for free
    # device tensors are not read back per step: the engine clamps them into the table, the forward stays finite,
    # and the sticky device flag becomes the same IndexError at the next check_deferred() - exactly once
    m.check_deferred()
    out = m(input_ids=ids.cuda(), attention_mask=att.cuda(), position_ids=(pos + 60).cuda(), task_labels=y.cuda())
    assert torch.isfinite(out.task_loss)
    with pytest.raises(IndexError):
        m.check_deferred()
    m.check_deferred()
    out = m(input_ids=ids.cuda(), attention_mask=att.cuda(), position_ids=pos.cuda(), task_labels=y.cuda())
    m.check_deferred()


@pytest.mark.parametrize("layout", ["padded", "varlen"])
@pytest.mark.parametrize("name", ["ft_base_ls_s256", "ft_base_s2048"])
def test_full_width_finetune_matches_reference(name, layout):
    """BASELINE's fine-tune configurations at full width and full sequence length against outputs of the REAL reference at a small
    batch (tools/make_golden.py): C3 = ogbl-ppa form (base model, LayerScale 1.0, S = 256, V = 41245, B = 4) and C4 (S = 2048, B =
    2).  The loss tolerance is derived, not a constant: max(1e-4, 2 x the reference's own bf16-vs-fp32 loss gap on the case) - 2.1e-3
    for C3, 1.5e-2 for C4 (the 3e-2 of round 2 was 3-30x wider than anything measured).  Pooled logits, every per-parameter gradient
    norm and 64x64 gradient blocks against the reference's; both token layouts."""
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    n_tok = int(batch["attention_mask"].sum()) if layout == "varlen" else None
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                     num_tokens=n_tok)
    e.backward()
    torch.cuda.synchronize()
    assert e.varlen_status()[0] == (layout == "varlen")
    tag = f"{name}_{layout}"
    got, want = float(loss.item()), float(z["loss"])
    tol = loss_tolerance(z, factor=FT_FACTOR)
    record_error(tag, "loss_rel_vs_reference_fp32", abs(got - want) / abs(want), tol)
    record_error(tag, "reference_bf16_vs_fp32_loss_gap", abs(float(z["loss_bf16"]) - want) / abs(want), float("nan"))
    assert abs(got - want) <= tol * abs(want), f"{tag}: loss {got} vs reference fp32 {want} (reference bf16 {float(z['loss_bf16'])})"
    err = np.abs(logits.cpu().numpy() - z["logits"]).max()
    ref_err = np.abs(z["logits_bf16"] - z["logits"]).max()
    record_error(tag, "task_logits_max_abs_vs_reference_fp32", err, max(2 * ref_err, 1e-2))
    assert err < max(2 * ref_err, 1e-2), f"{tag}: task logits max-abs {err} (reference bf16-vs-fp32 {ref_err})"
    grads = e.grads()
    names = list(state.keys())
    gn = np.array([float(grads[n].float().norm()) for n in names])
    ref, ref_bf = z["grad_norms"], z["grad_norms_bf16"]
    big = ref >= 1e-3 * ref.max()
    worst = float(np.max(np.abs(gn[big] - ref[big]) / ref[big]))
    worst_ref = float(np.max(np.abs(ref_bf[big] - ref[big]) / ref[big]))
    # (B = 2 ... 4 pooled rows: every gradient carries the common factor softmax(logits) of a handful of samples, which moves with
    #  the forward's bf16 rounding - the reference's own bf16 backward shows how much)
    tolg = max(5e-2, 1.5 * worst_ref)
    record_error(tag, "per_parameter_grad_norm_max_rel (norm >= 1e-3 of the largest)", worst, tolg)
    record_error(tag, "reference_bf16_vs_fp32 per_parameter_grad_norm_max_rel", worst_ref, float("nan"))
    assert worst < tolg, [(names[i], gn[i], ref[i]) for i in np.argsort(-np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max()))[:4]]
    for tagb, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l0_k", "model.layers.0.self_attn.k_proj.weight"),
                     ("l11_q", "model.layers.11.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                     ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
        blk = grads[pn].float().cpu().numpy()[:64, :64]
        ref_blk = z["gradblk_" + tagb]
        err = rel_l2(blk, ref_blk)
        ref_err = rel_l2(z["gradblk_bf16_" + tagb], ref_blk)
        tolb = max(6e-2, 1.5 * ref_err)
        record_error(tag, f"grad_block_rel_l2_own_norm {pn}[:64,:64]", err, tolb)
        record_error(tag, f"reference_bf16_vs_fp32 grad_block {pn}[:64,:64]", ref_err, float("nan"))
        assert err < tolb, f"{tag}: {pn} block rel-L2 {err} (reference bf16 {ref_err})"


@pytest.mark.parametrize("name", BASE_CASES)
def test_full_width_base_model_matches_reference(name):
    """The architecture of the headline run (base: d768 / L12 / ff3072, F=13, V=756) against outputs of the REAL reference at a
    small batch (tools/make_golden.py pt_base_*): standard init, and big weights (std 0.06 / heads 0.15: loss far from ln V,
    attention far from uniform, q/k gradients of ordinary size).  Loss held to max(1e-4, 1.5 x the reference's own bf16-vs-fp32
    gap); every per-parameter gradient norm and 64x64 gradient blocks of q / k / gate / down projections against the
    reference's; then three clip + AdamW steps against the reference's loss trajectory."""
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, _ = run_forward(e, spec, b, "pt")
    e.backward()
    torch.cuda.synchronize()
    got, want = float(loss.item()), float(z["loss"])
    tol = loss_tolerance(z)
    record_error(name, "loss_rel_vs_reference_fp32", abs(got - want) / abs(want), tol)
    record_error(name, "reference_bf16_vs_fp32_loss_gap", abs(float(z["loss_bf16"]) - want) / abs(want), float("nan"))
    assert abs(got - want) <= tol * abs(want), f"{name}: loss {got} vs reference fp32 {want} (reference bf16 {float(z['loss_bf16'])})"
    lg = e.head_logits().float().cpu().numpy()
    err, ref_err = rel_l2(lg[:64], z["logits"]), rel_l2(z["logits_bf16"], z["logits"])
    record_error(name, "logits_rel_l2_vs_reference_fp32", err, max(2.5 * ref_err, 1.5e-2))
    assert err < max(2.5 * ref_err, 1.5e-2), (err, ref_err)
    grads = e.grads()
    names = list(state.keys())
    gn = np.array([float(grads[n].float().norm()) for n in names])
    ref = z["grad_norms"]
    big = ref >= 1e-3 * ref.max()
    worst = float(np.max(np.abs(gn[big] - ref[big]) / ref[big]))
    record_error(name, "per_parameter_grad_norm_max_rel (norm >= 1e-3 of the largest)", worst, 5e-2)
    assert worst < 5e-2, [(names[i], gn[i], ref[i]) for i in np.argsort(-np.abs(gn - ref) / np.maximum(ref, 1e-3 * ref.max()))[:4]]
    for tag, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l0_k", "model.layers.0.self_attn.k_proj.weight"),
                    ("l11_q", "model.layers.11.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                    ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
        blk = grads[pn].float().cpu().numpy()[:64, :64]
        ref_blk = z["gradblk_" + tag]
        err = rel_l2(blk, ref_blk)
        ref_err = rel_l2(z["gradblk_bf16_" + tag], ref_blk)      # the reference's own bf16 backward on the same block
        tolb = max(6e-2, 1.5 * ref_err)
        record_error(name, f"grad_block_rel_l2_own_norm {pn}[:64,:64]", err, tolb)
        record_error(name, f"reference_bf16_vs_fp32 grad_block {pn}[:64,:64]", ref_err, float("nan"))
        assert err < tolb, f"{name}: {pn} block rel-L2 {err}"
    # three optimiser steps on the same batch: loss after each step against the reference trajectory
    losses = [got]
    e.adamw_step(ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"], ADAM["wd"], CLIP)
    for _ in range(2):
        l_, _ = run_forward(e, spec, b, "pt")
        losses.append(float(l_.item()))
        e.backward()
        e.adamw_step(ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"], ADAM["wd"], CLIP)
    l_, _ = run_forward(e, spec, b, "pt")
    losses.append(float(l_.item()))
    rel = np.abs(np.array(losses) - z["adamw_losses"]) / np.abs(z["adamw_losses"])
    # standard init: 5e-3.  The big-weight model sits at a loss of 55 where Adam's sign-like first steps (lr 1e-3 on every
    # weight) move the loss by 3-5 per step: gradient noise of a few % on near-zero entries flips update signs -> 2e-2
    ttol = 5e-3 if name == "pt_base_std" else 2e-2
    record_error(name, "loss_after_3_adamw_steps_rel_vs_reference", float(rel[-1]), ttol)
    record_error(name, "adamw_trajectory_losses_max_rel", float(rel.max()), ttol)
    assert rel.max() < ttol, (losses, z["adamw_losses"])


def test_c4_long_sequence_full_model_backward_matches_oracle():
    """Full 12-layer forward + BACKWARD at S = 2048 (C4 / C5 of BASELINE.json: base model, F = 4, V = 41245), B = 1, against the
    oracle: loss, pooled logits and the gradients of representative tensors of the first / middle / last layer, the score
    head and the embedding (multi-wave attention backward over 64 key tiles inside the full model, not only at op level)."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V = 1, 2048, 4, 41245
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                   max_position=2048)
    state = weights_mod.make_state_dict(spec, seed=8, std=0.04, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=93, lengths="full")
    batch["task_labels"][:] = 1
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    loss, logits = run_forward(e, spec, b, "ft")
    e.backward()
    torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn, lk, gk = oracle_fn(spec, b, "ft")
    out, grads = O.loss_and_grads(fn, p, lk)
    want = out[lk].item()
    ltol = _ft_loss_tol("ft_base_s2048")
    record_error("c4_S2048_B1_backward", "loss_rel_vs_oracle", abs(float(loss) - want) / max(abs(want), 0.1), ltol)
    assert abs(float(loss) - want) <= ltol * max(abs(want), 0.1), (float(loss), want)
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    # ONE sample, two classes: d loss / d logits = (-p0, p0), so every gradient of the model carries the factor p0 - and p0 comes
    # from the bf16 forward of 12 layers at S = 2048 (logit error ~1e-2 on a difference of ~3 moves p0 by several %).  That
    # common factor is the forward's rounding, not a backward error: it is measured from the two logit vectors and divided out,
    # the backward is then held to 4e-2; the raw error is recorded beside it (and bounded by the factor's own deviation).
    lg_e = logits.float().cpu().numpy().reshape(-1)[:2]
    lg_o = out[gk].detach().float().numpy().reshape(-1)[:2]

    def p0(lg):
        z = np.exp(lg - lg.max())
        return float(z[0] / z.sum())
    scale = p0(lg_e) / p0(lg_o)
    record_error("c4_S2048_B1_backward", "dlogit_scale_minus_1 (forward rounding through p0)", abs(scale - 1.0), 0.15)
    assert abs(scale - 1.0) < 0.15, (lg_e, lg_o)
    for k in ("score.weight", "model.layers.11.mlp.down_proj.weight", "model.layers.11.self_attn.q_proj.weight",
              "model.layers.6.self_attn.k_proj.weight", "model.layers.6.self_attn.v_proj.weight",
              "model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.o_proj.weight",
              "model.layers.0.mlp.gate_proj.weight", "model.layers.0.input_layernorm.weight", "model.embed_tokens.weight"):
        w = grads[k].numpy()
        g = got[k].float().cpu().numpy()
        den = max(float(np.linalg.norm(w)), 1e-2 * gmax)
        raw = float(np.linalg.norm(g - w)) / den
        err = float(np.linalg.norm(g / scale - w)) / den
        record_error("c4_S2048_B1_backward", "grad_rel_l2_raw " + k, raw, 4e-2 + 1.2 * abs(scale - 1.0))
        record_error("c4_S2048_B1_backward", "grad_rel_l2 " + k, err, 4e-2)
        assert err < 4e-2, f"{k}: {err} (raw {raw}, scale {scale})"
        assert raw < 4e-2 + 1.2 * abs(scale - 1.0), f"{k}: raw {raw}, scale {scale}"


def _attn_drop_keep(seed, B, H, S, p):
    """Python twin of drop_mul() in csrc/attention.hip (same as tests/test_gpu_ops.py:_drop_mask)."""
    bh = np.arange(B * H, dtype=np.uint64)[:, None, None]
    q = np.arange(S, dtype=np.uint64)[None, :, None]
    k = np.arange(S, dtype=np.uint64)[None, None, :]
    M32 = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) ^ ((bh * np.uint64(0x9E3779B1)) & M32)) & M32
    x = (x + q * np.uint64(0x85EBCA77) + (k >> np.uint64(1)) * np.uint64(0xC2B2AE3D)) & M32
    x ^= x >> np.uint64(16)
    x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & M32      # v_mul_u32_u24: the low 24 bits times a 24-bit constant
    w = x ^ (x >> np.uint64(15))
    f = np.where((k & np.uint64(1)) == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    thresh = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    keep = f >= thresh
    return torch.from_numpy((keep.astype(np.float32) / (1.0 - p)).reshape(B, H, S, S))


def test_c3_training_mode_dropouts_exact_mask():
    """C3 as the ogbl-ppa scripts train it: base model with LayerScale, stochastic depth path_pdrop = 0.2 AND attention dropout
    0.1, S = 256 (B = 4 so the CPU oracle finishes quickly).  Both masks are counter hashes; their Python twins feed the oracle
    the exact masks the kernels used, so loss and gradients must agree like in eval mode."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V, seed, p_attn, p_path = 4, 256, 4, 41245, 4242, 0.1, 0.2
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                   max_position=1024, layer_scale_init=1.0, path_pdrop=p_path)
    state = weights_mod.make_state_dict(spec, seed=9, std=0.04, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=94, lengths="uniform", min_len=S // 2)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    e.set_dropout(p_attn, p_path, seed)
    loss, _, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL)
    e.backward()
    torch.cuda.synchronize()
    L_, H = spec.num_layers, spec.num_heads
    pm = lambda l, w: _path_keep(seed, l, w, B, p_path * l / (L_ - 1))
    ak = lambda l: _attn_drop_keep((seed + 0x9E37 * l) & 0xFFFFFFFF, B, H, S, p_attn)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    out, grads = O.loss_and_grads(lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"],
                                                           b["task_labels"], path_mult=pm, attn_keep=ak), p, "task_loss")
    want = out["task_loss"].item()
    record_error("c3_train_mode_dropouts", "loss_rel_vs_oracle_same_masks", abs(float(loss) - want) / max(abs(want), 0.1), 3e-2)
    assert abs(float(loss) - want) <= 3e-2 * max(abs(want), 0.1), (float(loss), want)
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("score.weight", "model.layers.11.mlp.down_proj.weight", "model.layers.6.self_attn.q_proj.weight",
              "model.layers.6.self_attn.v_proj.weight", "model.layers.0.lambda_1", "model.layers.11.lambda_2",
              "model.layers.0.input_layernorm.weight", "model.embed_tokens.weight"):
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("c3_train_mode_dropouts", "grad_rel_l2 " + k, err, 8e-2)
        assert err < 8e-2, f"{k}: {err}"


@pytest.mark.gpu
@pytest.mark.parametrize("width", [512, 768, 1024])
def test_ls_norm_backward_lean_form_matches_the_wide_form(width):
    """The fused RMSNorm + LayerScale / DropPath backward has two lane mappings (engine.hip: rmsnorm_bwd_ls4_kernel - 8-byte pieces, all
    lanes busy, d = 512 / 768 / 1024 - and the 16-byte-chunk form of every other width): same expressions per element, another fp32 order
    of the row's dot product.  Same batch, same masks (LayerScale, stochastic depth, mlp_dropout, var-len rows): equal loss (the forward
    is not involved), gradients equal up to rounding flips."""
    from _util import spec_mod, weights_mod, synth
    import dataclasses
    B, S, F, V = 6, 64, 4, 2000
    spec = dataclasses.replace(spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1,
                                                       num_labels=2, max_position=256, layer_scale_init=1.0, path_pdrop=0.2, mlp_pdrop=0.1),
                               hidden_size=width, intermediate_size=2 * width, num_heads=width // 64, num_layers=3)
    state = weights_mod.make_state_dict(spec, seed=5, std=0.04, head_std=0.1)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=31, lengths="uniform", min_len=S // 3)
    b = tb(batch)
    runs = []
    for wide in (1, 0):
        L.check(L.load().gget_debug_set(11, wide))
        try:
            e = make_engine(spec, batch)
            e.load_state_dict(state)
            e.set_dropout(0.1, 0.2, 99)
            e.set_dropout_ex(0.0, 0.1, 0.0)
            loss, _, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                        num_tokens=int(batch["attention_mask"].sum()))
            e.backward()
            torch.cuda.synchronize()
            runs.append((float(loss), {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()}))
        finally:
            L.check(L.load().gget_debug_set(11, 0))
    (lw, gw), (ll, gl) = runs
    assert lw == ll
    gmax = max(float(np.linalg.norm(v)) for v in gw.values())
    worst = 0.0
    for k in gw:
        err = float(np.linalg.norm(gl[k] - gw[k])) / max(float(np.linalg.norm(gw[k])), 1e-2 * gmax)
        worst = max(worst, err)
        assert err < 4e-3, f"{k}: {err}"
    record_error(f"ls_norm_bwd_lean_d{width}", "grad_rel_l2_vs_wide_form_worst", worst, 4e-3)


@pytest.mark.gpu
def test_auc_loss_matches_oracle_with_the_same_pairs():
    """loss_type "auc" (modeling_finetune.py:203-207, src/utils/loss_utils.py:25-53) through the drop-in model class: the
    engine's counter-hash negative sampling is reproduced by its Python twin, the oracle (pinned to the reference by
    tests/golden/ft_tiny_auc.npz) evaluates the same pairs - loss on the engine's own logits exact to fp32 rounding, loss and
    gradients of the whole model within the fine-tune tolerances."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "ft_tiny_auc.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    num_neg = int(z["num_neg"])
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=1000, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=4, num_labels=2,
                           loss_type="auc", num_neg=num_neg, problem_type="single_label_classification")
    model = M.GraphGPTTaskModel(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.auc_seed = 77
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"],
                task_labels=b["task_labels"])
    loss = float(out.task_loss.item())
    out.task_loss.backward()
    idx = M.auc_pairs(b["task_labels"].numpy(), num_neg, model.last_auc_seed)
    lg = out.task_logits.float().cpu()
    on_own_logits = O.auc_loss(lg[:, 1] - lg[:, 0], b["task_labels"].view(-1), num_neg, idx).item()
    record_error("ft_tiny_auc", "loss_rel_vs_oracle_on_engine_logits", abs(loss - on_own_logits) / abs(on_own_logits), 2e-6)
    assert abs(loss - on_own_logits) <= 2e-6 * abs(on_own_logits), (loss, on_own_logits)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                  problem_type="single_label_classification", loss_type="auc", num_neg=num_neg, auc_idx=idx)
    o, grads = O.loss_and_grads(fn, p, "task_loss")
    want = o["task_loss"].item()
    record_error("ft_tiny_auc", "loss_rel_vs_oracle", abs(loss - want) / abs(want), 2e-2)
    assert abs(loss - want) <= 2e-2 * abs(want), (loss, want)
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("score.weight", "model.layers.1.mlp.down_proj.weight", "model.layers.0.self_attn.q_proj.weight",
              "model.embed_tokens.weight"):
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("ft_tiny_auc", "grad_rel_l2 " + k, err, 3e-2)
        assert err < 3e-2, f"{k}: {err}"
    # another call draws other pairs (the reference draws a fresh randperm every call)
    out2 = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"],
                 task_labels=b["task_labels"])
    assert float(out2.task_loss.item()) != loss


@pytest.mark.gpu
@pytest.mark.parametrize("gated,layer_scale", [(False, 0.0), (True, 1.0)])
def test_embed_and_mlp_dropouts_exact_mask(gated, layer_scale):
    """embed_pdrop = 0.1, mlp_pdrop = 0.2 in training mode (modeling_helpers.py:96-98, utils_graphgpt.py:69-80; positions pinned
    to the reference by tests/golden/pt_tiny_dropouts.npz through the oracle): the engine's counter-hash masks are regenerated
    by the Python twin and handed to the oracle - loss and gradients (incl. the embedding table's, which sees a different mask
    on every (cell, channel)) within the pre-train tolerances; evaluation mode is unaffected."""
    from _util import spec_mod, weights_mod, synth
    M = importlib.import_module("graph-gpt_amd.modeling")
    B, S, F, V = 4, 24, 13, 756
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=V, stacked_feat=F, next_n_token=F,
                                   gated_agg=gated, layer_scale_init=layer_scale, mlp_pdrop=0.2, embed_pdrop=0.1)
    state = weights_mod.make_state_dict(spec, seed=733, std=0.06, head_std=0.15)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=73)
    b = tb(batch)
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    l_eval = float(run_forward(e, spec, b, "pt")[0])     # (the returned loss is a view of the engine's device scalar)
    seed = 0x1234ABCD
    e.set_dropout(0.0, 0.0, seed)
    e.set_dropout_ex(0.1, 0.2)
    loss = float(run_forward(e, spec, b, "pt")[0])
    e.backward()
    torch.cuda.synchronize()
    d, ff = spec.hidden_size, spec.intermediate_size
    ek = torch.from_numpy(M.elem_drop_keep(seed, "embed", 0, B * S * F, d, 0.1)).view(B, S, F, d)
    mk = lambda i: (torch.from_numpy(M.elem_drop_keep(seed, "mlp_act", i, B * S, ff, 0.2)).view(B, S, ff),
                    torch.from_numpy(M.elem_drop_keep(seed, "mlp_out", i, B * S, d, 0.2)).view(B, S, d))
    assert abs(float((ek == 0).float().mean()) - 0.1) < 0.01 and abs(float((mk(1)[0] == 0).float().mean()) - 0.2) < 0.01
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], embed_keep=ek, mlp_keep=mk)
    out, grads = O.loss_and_grads(fn, p, "head1_loss")
    want = out["head1_loss"].item()
    name = f"pt_tiny_dropouts_gated{int(gated)}_ls{int(layer_scale > 0)}"
    record_error(name, "loss_rel_vs_oracle_same_masks", abs(float(loss) - want) / abs(want), 2e-3)
    assert abs(float(loss) - want) <= 2e-3 * abs(want), (float(loss), want)
    assert abs(float(loss) - float(l_eval)) > 1e-3 * abs(want)        # the masks are really applied
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    keys = ["model.embed_tokens.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.1.mlp.gate_proj.weight",
            "model.layers.0.mlp.up_proj.weight", "model.layers.0.self_attn.q_proj.weight", "lm_head.weight"]
    if gated:
        keys.append("stacked_feat_agg.weight")
    if layer_scale > 0:
        keys.append("model.layers.1.lambda_2")
    for k in keys:
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error(name, "grad_rel_l2 " + k, err, 3e-2)
        assert err < 3e-2, f"{k}: {err}"
    # evaluation mode again: same loss as before the training step
    e.set_dropout(0.0, 0.0, 0)
    e.set_dropout_ex(0.0, 0.0)
    assert abs(float(run_forward(e, spec, b, "pt")[0]) - l_eval) <= 1e-6 * abs(l_eval)     # (fp32 atomics in the loss reduction)


@pytest.mark.gpu
def test_model_class_applies_embed_and_mlp_dropout_only_in_training_mode():
    """GraphGPTConfig(hidden_act="gelu", embed_pdrop, mlp_pdrop) through the drop-in class: train() draws fresh masks every call (nn.Dropout),
    eval() is deterministic and equal to the dropout-free model."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    from _util import synth
    kw = dict(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
              max_position_embeddings=64, causal_attention=False, stacked_feat=4, next_n_token=4)
    batch = synth.make_pretrain_batch(B=4, S=24, F=4, V=300, seed=3)
    b = {k: torch.from_numpy(v) for k, v in batch.items()}
    call = lambda m: float(m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"]).head1_loss.item())
    plain = M.GraphGPTPretrainBase(M.GraphGPTConfig(hidden_act="gelu", **kw), seed=5)
    drop = M.GraphGPTPretrainBase(M.GraphGPTConfig(hidden_act="gelu", embed_pdrop=0.1, mlp_pdrop=0.1, **kw), seed=5)
    plain.eval(); drop.eval()
    ref = call(plain)
    ev = call(drop)
    # (with mlp_pdrop in the config the residual adds run as their own kernels: the branch output is rounded to bf16 before the
    # add, as in the reference's bf16 module, where the GEMM-epilogue add of the plain model adds the fp32 accumulator)
    same = lambda x, y: abs(x - y) <= 1e-6 * abs(y)          # (fp32 atomics in the loss reduction: last-bit differences)
    assert same(call(drop), ev) and abs(ev - ref) <= 2e-5 * abs(ref)
    drop.train()
    a, c = call(drop), call(drop)
    assert abs(a - ev) > 2e-5 * abs(ref) and abs(c - ev) > 2e-5 * abs(ref) and a != c     # (loss ~ ln V at this init: small but real shifts)
    drop.eval()
    assert same(call(drop), ev)


@pytest.mark.gpu
def test_mlp_score_head_matches_oracle_eval_and_training_dropout():
    """`MLP` score head (config.mlp = [48, 32], biases; modules_utils.py:8-34) through the drop-in class: evaluation-mode loss /
    logits against the reference fixture and the oracle's gradients; training mode with config.dropout = 0.25: the oracle is fed
    with the masks the Python twin regenerates from the engine's seed (exact-mask comparison)."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "ft_tiny_mlphead.npz"))
    hm = tuple(int(x) for x in z["head_mlp"])
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=1,
                                   score_bias=True, head_mlp=hm)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    for k in state:
        if k.startswith("score."):
            state[k] = z["w_" + k]
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13, num_labels=1,
                           mlp=list(hm), dropout=float(z["p"]), problem_type="regression")
    model = M.GraphGPTTaskModel(cfg, seed=1)
    assert sorted(k for k in model.state_dict() if k.startswith("score.")) == sorted(k for k in state if k.startswith("score."))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    call = lambda: model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"],
                         task_labels=b["task_labels"])
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    kw = dict(problem_type="regression", loss_type=None)
    keys = ("score.mlp_modules.0.weight", "score.mlp_modules.1.weight", "score.mlp_modules.2.bias", "score.mlp_modules.0.bias",
            "model.layers.1.mlp.down_proj.weight", "model.embed_tokens.weight")

    def compare(tag, out, head_keep):
        loss = float(out.task_loss.item())
        out.task_loss.backward()
        p = O.to_params(st_bf, torch.float32)
        fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                      head_keep=head_keep, **kw)
        o, grads = O.loss_and_grads(fn, p, "task_loss")
        want = o["task_loss"].item()
        record_error("ft_tiny_mlphead", tag + " loss_rel_vs_oracle", abs(loss - want) / abs(want), 2e-2)
        assert abs(loss - want) <= 2e-2 * abs(want), (tag, loss, want)
        got = model._engine.grads()
        gmax = max(float(g.norm()) for g in grads.values())
        for k in keys:
            w = grads[k].numpy()
            err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
            record_error("ft_tiny_mlphead", tag + " grad_rel_l2 " + k, err, 6e-2)
            assert err < 6e-2, f"{tag} {k}: {err}"
        return loss

    model.eval()
    out = call()
    lg = out.task_logits.float().cpu().numpy()
    err = float(np.abs(lg - z["logits"]).max() / np.abs(z["logits"]).max())
    record_error("ft_tiny_mlphead", "eval logits_max_rel_vs_reference", err, 2e-2)
    assert err < 2e-2
    l_eval = compare("eval", out, None)
    assert abs(l_eval - float(z["loss"])) <= 3e-2 * abs(float(z["loss"]))
    model.train()
    out = call()
    pd, B = float(z["p"]), b["input_ids"].shape[0]
    dims = [spec.hidden_size] + list(hm)
    hk = lambda i: torch.from_numpy(M.elem_drop_keep(model.last_dropout_seed, "head", i, B, dims[i], pd))
    l_train = compare("train", out, hk)
    assert abs(l_train - l_eval) > 1e-3 * abs(l_eval)


@pytest.mark.gpu
def test_focal_loss_matches_reference_and_oracle():
    """config.focal_gamma = 2 through the drop-in class (FocalLoss, utils_graphgpt.py:340-376): loss against the reference
    fixture at the pre-train tolerance rule's floor for bf16 (2e-3), gradients against the oracle."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "pt_tiny_focal.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    gamma = float(z["gamma"])
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13, next_n_token=13,
                           focal_gamma=gamma)
    model = M.GraphGPTPretrainBase(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.eval()
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
    loss = float(out.head1_loss.item())
    out.head1_loss.backward()
    record_error("pt_tiny_focal", "loss_rel_vs_reference_fp32", abs(loss - float(z["loss"])) / float(z["loss"]), 2e-3)
    assert abs(loss - float(z["loss"])) <= 2e-3 * float(z["loss"]), (loss, float(z["loss"]))
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], focal_gamma=gamma)
    o, grads = O.loss_and_grads(fn, p, "head1_loss")
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("lm_head.weight", "n_token_proj.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
              "model.embed_tokens.weight"):
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("pt_tiny_focal", "grad_rel_l2 " + k, err, 6e-2)
        assert err < 6e-2, f"{k}: {err}"
    # gamma = 0 is the plain cross-entropy again
    model.config.focal_gamma = 0.0
    plain = float(model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"]).head1_loss.item())
    assert plain - loss > 5e-4 * loss          # (1 - p_t)^2 < 1: slightly below the plain CE at this (near-uniform) init


@pytest.mark.gpu
def test_packed_long_rows_with_dropout_through_the_engine():
    """Packed rows of S = 320 (block-diagonal [B,S,S] mask, position ids running across graphs) in training mode with attention
    dropout 0.1: the engine's long-sequence attention kernels on per-token key ranges, with q / k stored ROTATED (the backward
    kernels rotate dq / dk back in their epilogues) - loss and gradients against the oracle fed with the same dropout mask."""
    from _util import spec_mod, weights_mod, synth
    B, S, F, V, seed, p_attn = 2, 320, 4, 300, 2024, 0.1
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=V, stacked_feat=F, next_n_token=F, max_position=512)
    state = weights_mod.make_state_dict(spec, seed=17, std=0.06, head_std=0.15)
    batch = synth.make_packed_pretrain_batch(B=B, S=S, F=F, V=V, seed=57, mean_len=40, min_len=8)
    b = tb(batch)
    assert b["attention_mask"].dim() == 3
    e = make_engine(spec, batch)
    e.load_state_dict(state)
    e.set_dropout(p_attn, 0.0, seed)
    loss = float(e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], None, b["position_ids"]))
    e.backward()
    torch.cuda.synchronize()
    H = spec.num_heads
    ak = lambda l: _attn_drop_keep((seed + 0x9E37 * l) & 0xFFFFFFFF, B, H, S, p_attn)
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)

    def fn(q):
        x, _ = O.stacked_embed(q["model.embed_tokens.weight"], b["input_ids"], q.get("stacked_feat_agg.weight"))
        hidden = O.backbone(spec, q, x, b["attention_mask"], b["position_ids"], attn_keep=ak)
        l_, lg = O.smtp_head(spec, q, hidden, b["labels"])
        return dict(head1_loss=l_, head1_logits=lg)
    out, grads = O.loss_and_grads(fn, p, "head1_loss")
    want = out["head1_loss"].item()
    record_error("pt_tiny_packed_S320_dropout", "loss_rel_vs_oracle_same_mask", abs(loss - want) / abs(want), 2e-3)
    assert abs(loss - want) <= 2e-3 * abs(want), (loss, want)
    got = e.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight", "model.layers.1.self_attn.k_proj.weight",
              "model.layers.1.self_attn.v_proj.weight", "model.layers.0.mlp.down_proj.weight", "model.embed_tokens.weight", "lm_head.weight"):
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("pt_tiny_packed_S320_dropout", "grad_rel_l2 " + k, err, 6e-2)
        assert err < 6e-2, f"{k}: {err}"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_stack_method_long_matches_reference_and_oracle(kind):
    """config.stack_method = "long" through the drop-in classes (proteins_supervised.sh:31): the per-token 1 / (non-zero ids) embedding
    ratio in forward and backward (modeling_helpers.py:106-110) and, for pre-training, the per-feature-level loss weights
    (:327-342).  The batch has empty (0-valued) feature cells in real rows.  Loss against the reference fixture, gradients against
    the oracle on the bf16-rounded weights; the pad row of the embedding table takes no gradient (nn.Embedding padding_idx)."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    pt = kind == "pt"
    z = np.load(os.path.join(GOLDEN, "pt_tiny_long.npz" if pt else "ft_tiny_long.npz"))
    if pt:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    else:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13,
                           next_n_token=13 if pt else 1, stack_method="long", num_labels=2)
    model = (M.GraphGPTPretrainBase if pt else M.GraphGPTTaskModel)(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.eval()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    if pt:
        out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
        loss_t, lk = out.head1_loss, "head1_loss"
        fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], stack_long=True)
        short = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"])
    else:
        out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"],
                    task_labels=b["task_labels"])
        loss_t, lk = out.task_loss, "task_loss"
        fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], stack_long=True)
        short = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"])
    loss = float(loss_t.item())
    loss_t.backward()
    tag = f"{kind}_tiny_long"
    ref = float(z["loss"])
    tol = 2e-3 if pt else 2e-2       # (fine-tune: a 12-sample CE at head_std 0.3 - the rule's FT factor on the reference's own bf16 gap)
    record_error(tag, "loss_rel_vs_reference_fp32", abs(loss - ref) / ref, tol)
    assert abs(loss - ref) <= tol * ref, (loss, ref)
    with torch.no_grad():
        assert abs(float(short(p)[lk]) - ref) > 5 * abs(loss - ref)      # the engine is on the "long" arithmetic, not near "short"
    o, grads = O.loss_and_grads(fn, p, lk)
    if not pt:
        lg = out.task_logits.float().cpu().numpy()
        err = float(np.abs(lg - z["logits"]).max()) / float(np.abs(z["logits"]).max())
        record_error(tag, "logits_max_rel_vs_reference_fp32", err, 3e-2)
        assert err < 3e-2, err
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    keys = ["model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.embed_tokens.weight"]
    keys += ["lm_head.weight", "n_token_proj.weight"] if pt else ["score.weight"]
    for k in keys:
        w = grads[k].numpy()
        err = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error(tag, "grad_rel_l2 " + k, err, 6e-2)
        assert err < 6e-2, f"{k}: {err}"
    assert float(got["model.embed_tokens.weight"][0].abs().max()) == 0.0


@pytest.mark.gpu
def test_token_level_ce_matches_reference_and_oracle():
    """config.loss_type = "token_ce" through the drop-in class (node-level tasks, modeling_finetune.py:162-164, :198-202): labels [B,S]
    with -100 on unlabelled rows, `task_logits` for every row.  Logits and loss against the reference fixture, gradients against the
    oracle on the bf16-rounded weights; evaluation without labels returns the same all-row logits."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "ft_tiny_tokence.npz"))
    C = 7
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=C)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13, next_n_token=1,
                           num_labels=C, loss_type="token_ce", problem_type="single_label_classification")
    model = M.GraphGPTTaskModel(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.eval()
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], task_labels=b["task_labels"])
    loss = float(out.task_loss.item())
    out.task_loss.backward()
    ref = float(z["loss"])
    record_error("ft_tiny_tokence", "loss_rel_vs_reference_fp32", abs(loss - ref) / ref, 5e-3)
    assert abs(loss - ref) <= 5e-3 * ref, (loss, ref)
    lg = out.task_logits.float().cpu().numpy()
    assert lg.shape == z["logits"].shape == (10, 24, C)
    real = (b["input_ids"][:, :, 0] != 0).numpy()
    err = float(np.abs(lg - z["logits"])[real].max()) / float(np.abs(z["logits"][real]).max())
    record_error("ft_tiny_tokence", "logits_max_rel_vs_reference_fp32 (real rows)", err, 3e-2)
    assert err < 3e-2, err
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], loss_type="token_ce")
    o, grads = O.loss_and_grads(fn, p, "task_loss")
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("score.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.embed_tokens.weight"):
        w = grads[k].numpy()
        e = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("ft_tiny_tokence", "grad_rel_l2 " + k, e, 6e-2)
        assert e < 6e-2, f"{k}: {e}"
    with torch.no_grad():
        ev = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"])
    assert ev.task_loss is None and torch.equal(ev.task_logits, out.task_logits)


@pytest.mark.gpu
def test_rope_range_matches_reference_and_oracle():
    """config.rope_range = 6 through the drop-in class (utils_graphgpt.reset_pos_ids :574-581): uneven position ids per row, rescaled to
    [0, 6) - the engine evaluates the rotary angles per token.  Logits / loss against the reference fixture, q / k projection
    gradients (the rotation's backward) against the oracle on the bf16-rounded weights."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "ft_tiny_roperange.npz"))
    rr = float(z["rope_range"])
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2, rope_range=rr)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13, next_n_token=1,
                           num_labels=2, rope_range=rr)
    model = M.GraphGPTTaskModel(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.eval()
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], task_labels=b["task_labels"])
    loss = float(out.task_loss.item())
    out.task_loss.backward()
    ref = float(z["loss"])
    record_error("ft_tiny_roperange", "loss_rel_vs_reference_fp32", abs(loss - ref) / ref, 2e-2)
    assert abs(loss - ref) <= 2e-2 * ref, (loss, ref)
    lg = out.task_logits.float().cpu().numpy()
    err = float(np.abs(lg - z["logits"]).max()) / float(np.abs(z["logits"]).max())
    record_error("ft_tiny_roperange", "logits_max_rel_vs_reference_fp32", err, 3e-2)
    assert err < 3e-2, err
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"])
    o, grads = O.loss_and_grads(fn, p, "task_loss")
    import dataclasses
    with torch.no_grad():
        plain = float(O.task_forward(dataclasses.replace(spec, rope_range=0.0), p, b["input_ids"], b["attention_mask"], b["position_ids"],
                                     b["task_labels"])["task_loss"])
    assert abs(plain - ref) > 5 * abs(loss - ref)          # the engine is on the rescaled positions, not on the raw ones
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight", "model.layers.1.self_attn.k_proj.weight",
              "model.layers.1.mlp.down_proj.weight", "score.weight"):
        w = grads[k].numpy()
        e = float(np.linalg.norm(got[k].float().cpu().numpy() - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("ft_tiny_roperange", "grad_rel_l2 " + k, e, 6e-2)
        assert e < 6e-2, f"{k}: {e}"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_raw_embedding_inputs_match_reference_and_oracle(kind):
    """config.embed_dim = 64 through the drop-in classes: `inputs_raw_embeds` [B,S,64] -> (pre-train: emb_mask_token on the rows whose labels
    are all set) -> embed_layernorm -> embed_proj -> added to the stacked token embeddings (modeling_pretrain.py:131-149,
    modeling_helpers.py:127-139).  Loss against the reference fixture, the branch's three gradients against the oracle."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    pt = kind == "pt"
    z = np.load(os.path.join(GOLDEN, f"{kind}_tiny_rawembed.npz"))
    E = int(z["embed_dim"])
    if pt:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13, embed_dim=E)
    else:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2, embed_dim=E)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    state["embed_layernorm.weight"] = z["w_embed_ln"]
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13,
                           next_n_token=13 if pt else 1, num_labels=2, embed_dim=E)
    model = (M.GraphGPTPretrainBase if pt else M.GraphGPTTaskModel)(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    assert tuple(model.state_dict()["embed_proj.weight"].shape) == (spec.hidden_size, E)
    model.eval()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    if pt:
        assert tuple(model.state_dict()["emb_mask_token"].shape) == (1, 1, E)
        out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"], inputs_raw_embeds=b["inputs_raw_embeds"])
        loss_t, lk = out.head1_loss, "head1_loss"
        fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], inputs_raw_embeds=b["inputs_raw_embeds"])
        without = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"])
    else:
        out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], task_labels=b["task_labels"],
                    inputs_raw_embeds=b["inputs_raw_embeds"])
        loss_t, lk = out.task_loss, "task_loss"
        fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                      inputs_raw_embeds=b["inputs_raw_embeds"])
        without = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"])
    loss = float(loss_t.item())
    loss_t.backward()
    ref = float(z["loss"])
    tag = f"{kind}_tiny_rawembed"
    tol = 2e-3 if pt else 2e-2
    record_error(tag, "loss_rel_vs_reference_fp32", abs(loss - ref) / ref, tol)
    assert abs(loss - ref) <= tol * ref, (loss, ref)
    with torch.no_grad():
        assert abs(float(without(p)[lk]) - ref) > 5 * abs(loss - ref)       # the branch matters on this batch
    o, grads = O.loss_and_grads(fn, p, lk)
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    keys = ["embed_proj.weight", "embed_layernorm.weight", "model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight"]
    keys += ["emb_mask_token", "lm_head.weight"] if pt else ["score.weight"]
    for k in keys:
        w = grads[k].numpy().reshape(-1)
        e = float(np.linalg.norm(got[k].float().cpu().numpy().reshape(-1) - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error(tag, "grad_rel_l2 " + k, e, 6e-2)
        assert e < 6e-2, f"{k}: {e}"
    with pytest.raises(AssertionError):      # the forward needs the raw embeddings, like the reference's
        model(input_ids=b["input_ids"], attention_mask=b["attention_mask"])


@pytest.mark.gpu
def test_raw_embedding_dropout_exact_mask():
    """Training mode with embed_pdrop = 0.2 on a model with raw-embedding inputs: `raw_embed_dropout` (modeling_pretrain.py:71-72, :146-147)
    sits between embed_layernorm and embed_proj; its counter-hash mask (stream 60) and the token-embedding mask (stream 48) are
    regenerated by the Python twin and handed to the oracle."""
    import os
    from _util import GOLDEN, spec_mod, weights_mod
    M = importlib.import_module("graph-gpt_amd.modeling")
    z = np.load(os.path.join(GOLDEN, "pt_tiny_rawembed.npz"))
    E, pdrop = int(z["embed_dim"]), 0.2
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13, embed_dim=E, embed_pdrop=pdrop)
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    state["embed_layernorm.weight"] = z["w_embed_ln"]
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    B, S, F = b["input_ids"].shape
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads,
                           max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=13, next_n_token=13,
                           embed_dim=E, embed_pdrop=pdrop)
    model = M.GraphGPTPretrainBase(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.train()
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"], inputs_raw_embeds=b["inputs_raw_embeds"])
    loss = float(out.head1_loss.item())
    out.head1_loss.backward()
    sd = model.last_dropout_seed
    ek = torch.from_numpy(M.elem_drop_keep(sd, "embed", 0, B * S * F, spec.hidden_size, pdrop)).view(B, S, F, spec.hidden_size)
    rk = torch.from_numpy(M.elem_drop_keep(sd, "raw", 0, B * S, E, pdrop)).view(B, S, E)
    assert abs(float((rk == 0).float().mean()) - pdrop) < 0.02
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], inputs_raw_embeds=b["inputs_raw_embeds"],
                                      embed_keep=ek, raw_keep=rk)
    o, grads = O.loss_and_grads(fn, p, "head1_loss")
    want = float(o["head1_loss"])
    record_error("pt_tiny_rawembed_dropout", "loss_rel_vs_oracle_same_masks", abs(loss - want) / want, 2e-3)
    assert abs(loss - want) <= 2e-3 * want, (loss, want)
    assert abs(loss - float(z["loss"])) > 1e-3 * want          # the masks are really applied
    got = model._engine.grads()
    gmax = max(float(g.norm()) for g in grads.values())
    for k in ("embed_proj.weight", "embed_layernorm.weight", "emb_mask_token", "model.embed_tokens.weight"):
        w = grads[k].numpy().reshape(-1)
        e = float(np.linalg.norm(got[k].float().cpu().numpy().reshape(-1) - w)) / max(float(np.linalg.norm(w)), 1e-2 * gmax)
        record_error("pt_tiny_rawembed_dropout", "grad_rel_l2 " + k, e, 6e-2)
        assert e < 6e-2, f"{k}: {e}"


@pytest.mark.gpu
def test_token_level_ce_without_any_labelled_row_is_nan_like_the_reference():
    """CrossEntropyLoss over rows that are all ignore_index is nan in torch (mean over nothing); the engine reports the same and a
    zero gradient instead of dividing by zero."""
    from _util import spec_mod, synth
    M = importlib.import_module("graph-gpt_amd.modeling")
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=300, stacked_feat=1, next_n_token=1, num_labels=5)
    cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=300, hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                           num_hidden_layers=spec.num_layers, num_attention_heads=spec.num_heads, max_position_embeddings=64,
                           causal_attention=False, stacked_feat=1, next_n_token=1, num_labels=5, loss_type="token_ce",
                           problem_type="single_label_classification")
    model = M.GraphGPTTaskModel(cfg, seed=3)
    batch = synth.make_task_batch(B=4, S=16, F=1, V=300, seed=5, num_labels=5)
    ids = torch.from_numpy(batch["input_ids"]); att = torch.from_numpy(batch["attention_mask"])
    lab = torch.full((4, 16), -100, dtype=torch.int64)
    out = model(input_ids=ids, attention_mask=att, task_labels=lab)
    assert torch.isnan(out.task_loss).item() and tuple(out.task_logits.shape) == (4, 16, 5)
    assert torch.isfinite(out.task_logits).all()


@pytest.mark.gpu
def test_gradient_norm_from_backward_partials_matches_full_pass(monkeypatch):
    """in a single-rank step the clip + AdamW launch takes the squared norm of the decoder layers' weight-gradient
    matrices from the per-tile sums their grouped weight-gradient launches left behind (gget_set_option GGET_OPT_NORM_FROM_BACKWARD,
    csrc/gemm.hip GemmGroup::sq_partials) and re-reads only the other tensors.  Same sum in another fixed order: the reported global
    norm (what torch.nn.utils.clip_grad_norm_ returns in the reference's step, training_utils.py:72-76) must equal the full pass to
    fp32 rounding, with the clip active (max_grad_norm below the norm) and the parameters after three steps must agree; the full-width
    base model is used because only its launch is the one-tile-per-CU kernel that leaves the partials."""
    M = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
               max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
    batch = synth.make_pretrain_batch(B=64, S=32, F=13, V=756, seed=21)
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}

    def run(full_pass):
        if full_pass:
            monkeypatch.delenv("GGET_NORM_FROM_BACKWARD", raising=False)
        else:
            monkeypatch.setenv("GGET_NORM_FROM_BACKWARD", "1")
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=4).cuda().eval()
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
        norms = []
        for _ in range(3):
            tr.batch_training(dev, eng)
            norms.append(float(eng.last_grad_norm))
        torch.cuda.synchronize()
        e = model._engine
        manual = float(e.grad_bf16.float().pow(2).sum().sqrt())          # the last step's gradients are still in the arena
        return norms, manual, e.master.detach().cpu().numpy().copy()

    w0 = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=4).cuda()._engine.master.detach().cpu().numpy().copy()
    nf, mf, wf = run(True)
    np_, mp_, wp = run(False)
    nf2, _, wf2 = run(True)                                               # a second full-pass run: the run-to-run noise floor
    assert all(n > 0.05 for n in nf)                                      # the clip is active in every step
    assert abs(np_[0] - nf[0]) <= 3e-6 * nf[0]                            # same gradients, two summation orders of their squares
    np.testing.assert_allclose(np_, nf, rtol=5e-4)                        # (later steps inherit run-to-run parameter differences)
    assert abs(np_[-1] - mp_) <= 2e-5 * mp_ and abs(nf[-1] - mf) <= 2e-5 * mf     # both are the norm of what is in the arena
    # parameters: noise-dominated gradients (fp32-atomic reductions of the norm / embedding gradients differ in their last bits from
    # run to run, and AdamW normalises every update) move single entries by a fraction of lr between ANY two runs - the two norm
    # paths must not differ by more than two full-pass runs do
    upd = float(np.linalg.norm(wf - w0))
    noise = float(np.linalg.norm(wf2 - wf))
    assert float(np.linalg.norm(wp - wf)) <= max(4.0 * noise, 2e-2 * upd), (float(np.linalg.norm(wp - wf)), noise, upd)


@pytest.mark.gpu
def test_reserved_cus_launch_menu_gives_the_same_step():
    """Data-parallel runs leave CUs free for the collective's workgroups (gget_debug_set(15, R): every GEMM tile plan, persistent grid and
    split-K fit counts CUs - R; the o projection's weight gradient leaves the grouped 256-tile launch for the split-K slab path so that the
    rest is one tile per CU again; csrc/gemm.hip g_gemm_cu_reserve).  Same arithmetic in other tilings: loss bit-equal is not promised,
    but loss and every gradient must agree to bf16 rounding with the full-chip menu - at the headline width, where the menus differ."""
    from _util import spec_mod, weights_mod, synth
    lib = L.load()
    B, S, F, V, d = 256, 32, 13, 756, 768
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=V, hidden_size=d, intermediate_size=4 * d, num_layers=2, num_heads=d // 64,
                              head_dim=64, stacked_feat=F, next_n_token=F, causal=False, max_position=1024)
    state = weights_mod.make_state_dict(spec, seed=9, std=0.02, head_std=0.05)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=77)
    b = tb(batch)
    n_tok = int(batch["attention_mask"].sum())
    out = {}
    try:
        for name, r in (("full", 0), ("reserved", 16), ("reserved32", 32)):
            L.check(lib.gget_debug_set(15, r))
            e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
            e.load_state_dict(state)
            loss = float(e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=n_tok))
            e.backward()
            torch.cuda.synchronize()
            out[name] = (loss, {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()})
    finally:
        L.check(lib.gget_debug_set(15, 0))
    lf, gf = out["full"]
    gmax = max(float(np.linalg.norm(g)) for g in gf.values())
    for name in ("reserved", "reserved32"):      # 16: the o weight gradient leaves the group; 32: q|k|v and o do
        lr, gr = out[name]
        assert abs(lf - lr) <= 2e-5 * abs(lf), (name, lf, lr)
        for k in gf:
            den = max(float(np.linalg.norm(gf[k])), 1e-2 * gmax)
            err = float(np.linalg.norm(gr[k] - gf[k])) / den
            record_error(name + "_cus_menu", "grad_rel_l2_vs_full_chip " + k, err, 8e-3)
            assert err <= 8e-3, (name, k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [False, True])
def test_cross_entropy_block_partials_match_the_atomic_sum(weighted):
    """The engine's cross-entropy launch leaves ONE partial loss sum per block in its workspace and the finalising launch adds them in block
    order (kernels.hip ce_rows_kernel / finalize_loss_kernel; 2048 same-address atomics cost the headline launch a third of its time);
    gget_debug_set(14, 0) = the atomic form.  Same row losses, two summation orders: the loss equal to fp32 rounding, the lm_head gradient -
    dlogits - untouched (the plain and the dLM-weighted normalisation, modeling_pretrain.py:210-236)."""
    from _util import spec_mod, weights_mod, synth
    lib = L.load()
    B, S, F, V, d = 96, 32, 13, 756, 256
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=V, hidden_size=d, intermediate_size=4 * d, num_layers=2, num_heads=d // 64,
                              head_dim=64, stacked_feat=F, next_n_token=F, causal=False, max_position=1024)
    state = weights_mod.make_state_dict(spec, seed=5, std=0.05, head_std=0.1)
    b = tb(synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=31, dlm_wgt=weighted))
    out = []
    try:
        for form in (0, 1, 1):
            L.check(lib.gget_debug_set(14, form))
            e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
            e.load_state_dict(state)
            loss = float(e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt")))
            e.backward()
            torch.cuda.synchronize()
            out.append((loss, e.grads()["lm_head.weight"].float().cpu().numpy().copy()))
    finally:
        L.check(lib.gget_debug_set(14, 1))
    assert out[0][0] > 0.5
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * abs(out[0][0]), (out[0][0], out[1][0])
    assert out[1][0] == out[2][0], "the block-ordered sum must reproduce itself"
    assert rel_l2(out[1][1], out[0][1]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode,d", [("labels", 768), ("labels_varlen", 768), ("dlm_weights", 768), ("inference", 768),
                                    ("labels", 576), ("labels_varlen", 576)])
def test_slot_sorted_head_matches_dense_head(mode, d):
    """Round 4: n_token_proj on the labelled cells only (cells sorted by slot, one GEMM with a weight block per row tile; kernels.hip
    "Slot-sorted SMTP head") against the dense form of rounds 1-3 - every selected row through all n slots, then the gather
    (modeling_helpers.py:263-301, the reference's order of operations) - on the full-width model: same loss and head logits (both forms
    run the K = d reduction of a cell's row in one tile-K order), and the gradients of every tensor within bf16 rounding (the sorted
    backward rounds one input-gradient row per cell to bf16 before a token's cells are summed; the dense one sums in the accumulator).
    The dense form itself is what the reference fixtures pin on the narrow models (d = 128 takes it: the sorted form needs d % 192 == 0).
    d = 576: d / 8 = 72 channels, so head_cell_sum_kernel's last pass over the channels has 8 live lanes while tokens carry up
    to 13 labelled cells - the cell rows must still come from all 13 lanes."""
    from _util import spec_mod, weights_mod, synth
    lib = L.load()
    B, S, F, V = 64, 32, 13, 756
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=V, hidden_size=d, intermediate_size=4 * d, num_layers=2, num_heads=d // 64,
                              head_dim=64, stacked_feat=F, next_n_token=F, causal=False, max_position=1024)
    state = weights_mod.make_state_dict(spec, seed=5, std=0.05, head_std=0.1)
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=31, dlm_wgt=(mode == "dlm_weights"))
    b = tb(batch)
    n_tok = int(batch["attention_mask"].sum()) if mode == "labels_varlen" else None
    out = {}
    try:
        for name, dense in (("sorted", 0), ("dense", 1)):
            L.check(lib.gget_debug_set(8, dense))
            e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
            e.load_state_dict(state)
            labels = None if mode == "inference" else b["labels"]
            loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], labels, b.get("wgt"), num_tokens=n_tok)
            logits = e.head_logits().float().cpu().numpy()
            grads = None
            if labels is not None:
                e.backward()
                torch.cuda.synchronize()
                grads = {k: v.float().cpu().numpy().copy() for k, v in e.grads().items()}
            out[name] = (None if loss is None else float(loss), logits, grads, e.head_counts())
    finally:
        L.check(lib.gget_debug_set(8, 0))
    (ls, gs, grs, cs), (ld, gd, grd, cd) = out["sorted"], out["dense"]
    assert cs == cd and gs.shape == gd.shape and gs.shape[0] == cs[1]
    if mode == "inference":
        assert cs[1] == B * S * F                                   # every cell is predicted
    np.testing.assert_array_equal(gs, gd)                           # head logits: bit-equal
    if ls is not None:
        assert abs(ls - ld) <= 2e-6 * abs(ld)
        gmax = max(float(np.linalg.norm(g)) for g in grd.values())
        for k in grd:
            den = max(float(np.linalg.norm(grd[k])), 1e-2 * gmax)
            err = float(np.linalg.norm(grs[k] - grd[k])) / den
            tol = 1e-6 if k in ("lm_head.weight",) else 8e-3        # lm_head's gradient does not pass through n_token_proj's backward
            record_error("slot_sorted_head_" + mode + ("" if d == 768 else f"_d{d}"), "grad_rel_l2_vs_dense " + k, err, tol)
            assert err <= tol, (k, err)
