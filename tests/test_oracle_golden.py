"""Pins the CPU oracle (oracle/gget_oracle.py) against golden vectors captured from the real
reference (tools/make_golden.py).  fp32: tight; bf16: the reference's own bf16 module path."""
import importlib
import os

import numpy as np
import pytest
import torch

from _util import ADAM, BASE_CASES, CLIP, FT_CASES, PT_CASES, ft_problem, load_case, rel_l2, tb
from oracle import gget_oracle as O


def _fwd_fn(spec, b, kind, name=""):
    if kind == "pt":
        def fn(p):
            return O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"))
        return fn, "head1_loss", "head1_logits"
    problem, loss_type = ft_problem(spec, b, name)

    def fn(p):
        return O.task_forward(spec, p, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                              sample_wgt=b.get("wgt"), problem_type=problem, loss_type=loss_type)
    return fn, "task_loss", "task_logits"


@pytest.mark.parametrize("name", PT_CASES + FT_CASES)
def test_oracle_fp32_matches_reference(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    p = O.to_params(state, torch.float32)
    fn, lk, gk = _fwd_fn(spec, b, kind, name)
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(out[lk].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"])) + 1e-6
    lg = out[gk].detach().float().numpy()
    assert tuple(lg.shape) == tuple(int(x) for x in z["logits_shape"])
    np.testing.assert_allclose(lg[:64], z["logits"], rtol=2e-4, atol=2e-5)
    names = list(state.keys())
    gn = np.array([float(grads[n].norm()) for n in names])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=2e-4, atol=1e-7)
    assert rel_l2(grads["model.embed_tokens.weight"].numpy(), z["grad_embed"]) < 1e-4
    assert rel_l2(grads["model.layers.0.self_attn.q_proj.weight"].numpy(), z["grad_l0_q"]) < 1e-4
    assert rel_l2(grads["model.layers.1.mlp.down_proj.weight"].numpy(), z["grad_l1_down"]) < 1e-4


@pytest.mark.parametrize("name", BASE_CASES)
def test_oracle_fp32_matches_reference_full_width(name):
    """The full-width base model (d768 / L12, the headline run's architecture) at a small batch: loss, logits, every
    per-parameter gradient norm and 64 x 64 gradient blocks of q / k / gate / down projections."""
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    p = O.to_params(state, torch.float32)
    fn, lk, gk = _fwd_fn(spec, b, "pt", name)
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(out[lk].item() - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out[gk].detach().float().numpy()[:64], z["logits"], rtol=5e-4, atol=5e-4)
    gn = np.array([float(grads[n].norm()) for n in state.keys()])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=5e-4, atol=1e-7)
    for tag, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l0_k", "model.layers.0.self_attn.k_proj.weight"),
                    ("l11_q", "model.layers.11.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                    ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
        assert rel_l2(grads[pn].numpy()[:64, :64], z["gradblk_" + tag]) < 5e-4, tag


@pytest.mark.parametrize("name", ["ft_base_ls_s256", "ft_base_s2048"])
def test_oracle_fp32_matches_reference_full_width_finetune(name):
    """BASELINE's fine-tune configurations at full width and full sequence length, small batch (round 3 fixtures): the ogbl-ppa form
    (base model + LayerScale, S = 256, V = 41245) and the long-sequence form (S = 2048) - oracle loss, pooled logits, per-parameter
    gradient norms and gradient blocks against the REAL reference."""
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    p = O.to_params(state, torch.float32)
    fn, lk, gk = _fwd_fn(spec, b, "ft", name)
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(out[lk].item() - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out[gk].detach().float().numpy()[:64], z["logits"], rtol=5e-4, atol=5e-5)
    gn = np.array([float(grads[n].norm()) for n in state.keys()])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=1e-3, atol=1e-7)
    for tag, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                    ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
        assert rel_l2(grads[pn].numpy()[:64, :64], z["gradblk_" + tag]) < 1e-3, tag


@pytest.mark.parametrize("name", ["pt_tiny_f13_a", "pt_tiny_bigw", "pt_tiny_s72", "ft_tiny_f4", "ft_tiny_ls"])
def test_oracle_bf16_tracks_reference_bf16(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    p = O.to_params(state, torch.bfloat16, requires_grad=False)
    fn, lk, gk = _fwd_fn(spec, b, kind, name)
    with torch.no_grad():
        out = fn(p)
    # same arithmetic as the reference's bf16 module path -> agreement far below bf16 noise vs fp32
    assert abs(out[lk].item() - float(z["loss_bf16"])) <= 2e-3 * abs(float(z["loss_bf16"])) + 1e-3
    assert rel_l2(out[gk].float().numpy()[:64], z["logits_bf16"]) < 2e-2


@pytest.mark.parametrize("name", ["pt_tiny_f13_a", "pt_tiny_bigw", "pt_tiny_wgt", "ft_tiny_f4", "ft_tiny_reg", "ft_tiny_mse", "ft_tiny_wce"])
def test_oracle_adamw_trajectory(name):
    z, spec, state, batch = load_case(name)
    kind = "pt" if name.startswith("pt") else "ft"
    b = tb(batch)
    p = O.to_params(state, torch.float32)
    fn, lk, _ = _fwd_fn(spec, b, kind, name)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in p.items()}
    losses, gns = [], []
    for step in range(1, 4):
        out, grads = O.loss_and_grads(fn, p, lk)
        losses.append(out[lk].item())
        with torch.no_grad():
            gns.append(O.adamw_step({k: t for k, t in p.items()}, grads, m, v, step, ADAM["lr"], ADAM["beta1"],
                                    ADAM["beta2"], ADAM["eps"], ADAM["wd"], CLIP))
    with torch.no_grad():
        losses.append(fn(p)[lk].item())
    np.testing.assert_allclose(losses, z["adamw_losses"], rtol=3e-4, atol=1e-5)
    np.testing.assert_allclose(gns, z["adamw_gnorms"], rtol=3e-4)
    fin = np.array([float(p[n].detach().norm()) for n in state.keys()])
    np.testing.assert_allclose(fin, z["adamw_final_norms"], rtol=1e-4)


def test_lr_schedules():
    import os
    from _util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "lr_schedules.npz"))
    for tag in ("a", "b"):
        max_lr, min_lr, total, warm = z["onecycle_" + tag + "_params"]
        got = [O.one_cycle_lr(s, max_lr, int(total) + 1, warm / total, min_lr) for s in range(int(total))]
        np.testing.assert_allclose(got, z["onecycle_" + tag], rtol=1e-9, atol=1e-15)


def test_inputs_regenerate():
    """The committed inputs are exactly what our seeded generator produces (so the GPU box can rebuild
    larger batches from seeds alone)."""
    from _util import synth
    z, spec, state, batch = load_case("pt_tiny_f13_a")
    again = synth.make_pretrain_batch(B=4, S=24, F=13, V=756, seed=0)
    for k in ("input_ids", "labels", "attention_mask", "position_ids"):
        np.testing.assert_array_equal(again[k], batch[k])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_smtp2d_masking_matches_reference(tag):
    """In-model SMTP masking (row A9 / N1): the oracle fed with the reference's own random draws (recorded in call order
    by tools/make_golden.py) reproduces the reference's masked ids and labels exactly."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "smtp2d.npz"))
    T = lambda k: torch.from_numpy(g[f"{tag}_{k}"])
    rate, power, rep, V, glob = g[f"{tag}_params"]
    ids, labels = O.smtp_2d_inputs_labels(T("ids"), T("node_idx"), T("u_sample"), T("u_rate"), T("u_cell"), T("token_shift"),
                                          T("u_replace"), smtp_2d_rate=float(rate), power=float(power),
                                          replace_rate=float(rep), vocab=int(V), global_2d_mask=bool(glob))
    assert torch.equal(ids, T("out_ids"))
    assert torch.equal(labels, T("out_labels"))
    assert (labels != -100).any() and (ids == 1).any()


@pytest.mark.parametrize("alg", ["maskgit_plus", "topk_margin", "entropy"])
def test_generation_loop_matches_reference(alg):
    """Generation (next item N3): the oracle's restatement of sample_per_batch / _batch_unmask_without_for_loop, driven by
    the oracle forward (labels=None -> logits for every feature token), reproduces the reference's token grid after every
    iteration (fixture: reference loop + reference model on CPU)."""
    import importlib
    import os
    spec_mod = importlib.import_module("graph-gpt_amd.spec")
    weights = importlib.import_module("graph-gpt_amd.weights")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=300, stacked_feat=4, next_n_token=4)
    assert list(g["meta_spec"]) == spec.as_c_ints()
    std, head_std, seed = g["meta_init"]
    p = O.to_params(weights.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(head_std)), torch.float32,
                    requires_grad=False)
    ids, att = torch.from_numpy(g["in_input_ids"]), torch.from_numpy(g["in_attention_mask"])

    def logits_fn(x):
        with torch.no_grad():
            return O.pretrain_forward(spec, p, x, att, labels=None)["head1_logits"]

    x, hist = O.sample_per_batch(logits_fn, ids, alg=alg, steps=6, eps=1e-3, mask_token_id=1)
    assert len(hist) == len(g[f"{alg}_hist"])
    for a, b in zip(hist, g[f"{alg}_hist"]):
        assert torch.equal(a, torch.from_numpy(b))
    assert torch.equal(x, torch.from_numpy(g[f"{alg}_x"]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_host_masking_matches_reference(tag):
    """Host SMTP masking (row A0): the oracle's _mask_stacked_input_ids_v2 with the reference's sampled cell list gives the
    reference's ids / labels (incl. a pad-valued cell that is labelled but not overwritten); k = ceil(cells * ratio)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hostmask.npz"))
    ids, idx = g[f"{tag}_ids"], g[f"{tag}_idx"]
    assert len(idx) == int(np.ceil(ids.size * float(g[f"{tag}_ratio"])))
    out_ids, out_lab = O.mask_stacked_input_ids_v2(ids, idx)
    np.testing.assert_array_equal(out_ids, g[f"{tag}_out_ids"])
    np.testing.assert_array_equal(out_lab, g[f"{tag}_out_labels"])
    alpha, wgt = O.smtp_mask_ratio(0.25, 0.01, 0.99, 2.0)
    t = 0.01 + 0.98 * 0.25
    assert alpha == 1 - t ** 2.0 and wgt == 2.0 / t


@pytest.mark.parametrize("tag,alg", [("origin", "origin"), ("gumbel", "maskgit_plus"), ("margin_t", "topk_margin")])
def test_stochastic_generation_loop_matches_reference(tag, alg):
    """The stochastic settings of the generation loop (N3): the oracle's restatement, fed with the random draws the reference
    took (recorded by tools/make_golden.py: categorical samples, transfer-mask uniforms, Gumbel uniforms), reproduces the
    reference's token grid after every iteration - temperature, top-p, top-k, the "origin" update and the Gumbel ranking."""
    import importlib
    import os
    spec_mod = importlib.import_module("graph-gpt_amd.spec")
    weights = importlib.import_module("graph-gpt_amd.weights")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation_stochastic.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=300, stacked_feat=4, next_n_token=4)
    std, head_std, seed = g["meta_init"]
    p = O.to_params(weights.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(head_std)), torch.float32,
                    requires_grad=False)
    ids, att = torch.from_numpy(g["in_input_ids"]), torch.from_numpy(g["in_attention_mask"])
    temperature, top_p, top_k, alg_temp = [float(v) for v in g[f"{tag}_cfg"]]

    def logits_fn(x):
        with torch.no_grad():
            return O.pretrain_forward(spec, p, x, att, labels=None)["head1_logits"]

    def draw_fn(it):
        d = {"x0": torch.from_numpy(g[f"{tag}_x0"][it])}
        if f"{tag}_u_transfer" in g.files:
            d["u_transfer"] = torch.from_numpy(g[f"{tag}_u_transfer"][it])
        if f"{tag}_u_gumbel" in g.files:
            d["u_gumbel"] = torch.from_numpy(g[f"{tag}_u_gumbel"][it])
        return d

    x, hist = O.sample_per_batch(logits_fn, ids, alg=alg, steps=6, eps=1e-3, mask_token_id=1, temperature=temperature,
                                 top_p=top_p if top_p > 0 else None, top_k=int(top_k) if top_k > 0 else None,
                                 alg_temp=alg_temp if alg_temp > 0 else None, draw_fn=draw_fn)
    assert len(hist) == len(g[f"{tag}_hist"])
    for it, (a, b) in enumerate(zip(hist, g[f"{tag}_hist"])):
        assert torch.equal(a, torch.from_numpy(b)), f"iteration {it}"


def test_oracle_inverse_cdf_sampler_and_filters():
    """sample_tokens' explicit categorical draw (inverse CDF, the HIP kernel's convention) has the right distribution, and the
    top-p / top-k filters keep exactly what the reference keeps (reference functions restated with a stable sort)."""
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(1, 12, generator=g) * 2
    probs = torch.softmax(logits / 0.7, dim=-1)[0]
    u = torch.rand(200000, generator=g)
    _, x0 = O.sample_tokens(logits.expand(200000, 12), temperature=0.7, u=u)
    freq = torch.bincount(x0, minlength=12).float() / 200000
    assert float((freq - probs).abs().max()) < 5e-3
    kept = O.top_k_logits(logits, 4) > torch.finfo(torch.float32).min
    assert int(kept.sum()) == 4 and torch.equal(kept[0].nonzero().view(-1).sort().values, logits[0].topk(4).indices.sort().values)
    lp = O.top_p_logits(logits, 0.6)[0]
    order = logits[0].argsort(descending=True)
    cum = torch.softmax(logits[0][order], dim=-1).cumsum(0)
    n_keep = int((cum <= 0.6).sum()) + 1          # everything up to and including the first token that crosses top_p
    assert torch.equal((lp > torch.finfo(torch.float32).min).nonzero().view(-1).sort().values, order[:n_keep].sort().values)


def _auc_case():
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "ft_tiny_auc.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_auc_loss_matches_reference():
    """loss_type "auc" (src/utils/loss_utils.py:25-53): with the negative-sample indices the reference's randperm drew, the
    restatement reproduces the reference's loss, logits and gradients."""
    z, spec, state, b = _auc_case()
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                  problem_type="single_label_classification", loss_type="auc", num_neg=int(z["num_neg"]),
                                  auc_idx=z["idx"])
    out, grads = O.loss_and_grads(fn, p, "task_loss")
    assert abs(out["task_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out["task_logits"].detach().numpy(), z["logits"], rtol=2e-4, atol=2e-5)
    for k, want in (("score.weight", z["grad_score"]), ("model.layers.1.mlp.down_proj.weight", z["grad_l1_down"])):
        g = grads[k].numpy()
        assert np.linalg.norm(g - want) <= 2e-4 * np.linalg.norm(want), k
    gn = np.array([float(grads[str(n)].norm()) for n in z["names"]])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=5e-4, atol=1e-7)


def test_auc_pairs_twin_is_a_balanced_permutation_draw():
    """graph-gpt_amd.modeling.auc_pairs (twin of the device sampling): perm(P * num_neg) % N - every negative is used
    floor or ceil of cnt / N times, like the reference's randperm % N."""
    m = importlib.import_module("graph-gpt_amd.modeling")
    y = np.array([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1, 1])
    idx = m.auc_pairs(y, 3, seed=5)
    P, N = 8, 4
    assert idx.shape == (P * 3,) and idx.min() >= 0 and idx.max() < N
    cnt = np.bincount(idx, minlength=N)
    assert cnt.min() >= (P * 3) // N and cnt.max() <= -(-(P * 3) // N)
    assert not np.array_equal(idx, m.auc_pairs(y, 3, seed=6))


def test_oracle_embed_and_mlp_dropouts_match_reference():
    """embed_pdrop / mlp_pdrop in train() mode (modeling_helpers.py:96-98, utils_graphgpt.py:69-80): fed with the keep masks the
    reference's dropout calls drew (recorded, tools/make_golden.py:dropout_fixture), the restatement reproduces the reference's
    loss and gradients - which pins WHERE each dropout sits."""
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "pt_tiny_dropouts.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    pe, pm = [float(x) for x in z["p"]]
    B, S = b["input_ids"].shape[:2]

    def unpack(key, shape, p):
        n = int(np.prod(shape))
        return torch.from_numpy(np.unpackbits(z[key])[:n].reshape(shape).astype(np.float32) / (1.0 - p))
    ek = unpack("embed_keep", tuple(int(x) for x in z["embed_shape"]), pe)
    mk = lambda i: (unpack(f"act_keep_{i}", (B, S, spec.intermediate_size), pm), unpack(f"out_keep_{i}", (B, S, spec.hidden_size), pm))
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], embed_keep=ek, mlp_keep=mk)
    out, grads = O.loss_and_grads(fn, p, "head1_loss")
    assert abs(out["head1_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    for k, want in (("model.embed_tokens.weight", z["grad_embed"]), ("model.layers.0.mlp.down_proj.weight", z["grad_l0_down"]),
                    ("model.layers.1.mlp.gate_proj.weight", z["grad_l1_gate"])):
        g = grads[k].numpy()
        assert np.linalg.norm(g - want) <= 2e-4 * np.linalg.norm(want), k
    gn = np.array([float(grads[str(n)].norm()) for n in z["names"]])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=5e-4, atol=1e-7)
    # and the masks matter: eval-mode arithmetic gives another loss
    assert abs(O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"])["head1_loss"].item() - float(z["loss"])) > 1e-3


def _mlphead_case():
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "ft_tiny_mlphead.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=1,
                                   score_bias=True, head_mlp=tuple(int(x) for x in z["head_mlp"]))
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    for k in state:
        if k.startswith("score."):
            state[k] = z["w_" + k]
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_mlp_score_head_matches_reference():
    """config.mlp = [48, 32] (src/utils/modules_utils.py:8-34): evaluation mode, and training mode with config.dropout = 0.25 fed
    with the keep masks the reference's dropout calls drew on the pooled rows."""
    z, spec, state, b = _mlphead_case()
    p = O.to_params(state, torch.float32)
    kw = dict(problem_type="regression", loss_type=None)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], **kw)
    out, grads = O.loss_and_grads(fn, p, "task_loss")
    assert abs(out["task_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out["task_logits"].detach().numpy(), z["logits"], rtol=2e-4, atol=2e-5)
    for k, want in (("score.mlp_modules.0.weight", z["grad_w0"]), ("score.mlp_modules.1.bias", z["grad_b1"]),
                    ("model.layers.1.mlp.down_proj.weight", z["grad_l1_down"])):
        assert np.linalg.norm(grads[k].numpy() - want) <= 2e-4 * np.linalg.norm(want), k
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)
    pd = float(z["p"])
    hk = lambda i: torch.from_numpy(z[f"train_keep_{i}"].astype(np.float32) / (1.0 - pd))
    fn2 = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], head_keep=hk, **kw)
    out2, grads2 = O.loss_and_grads(fn2, p, "task_loss")
    assert abs(out2["task_loss"].item() - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    assert np.linalg.norm(grads2["score.mlp_modules.0.weight"].numpy() - z["train_grad_w0"]) <= 2e-4 * np.linalg.norm(z["train_grad_w0"])
    np.testing.assert_allclose(np.array([float(grads2[str(n)].norm()) for n in z["names"]]), z["train_grad_norms"], rtol=5e-4, atol=1e-7)


def _focal_case():
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "pt_tiny_focal.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_focal_loss_matches_reference():
    """config.focal_gamma = 2 (utils_graphgpt.FocalLoss through _get_ce_loss): loss and gradients of the reference."""
    z, spec, state, b = _focal_case()
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], focal_gamma=float(z["gamma"]))
    out, grads = O.loss_and_grads(fn, p, "head1_loss")
    assert abs(out["head1_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    for k, want in (("lm_head.weight", z["grad_lm_head"]), ("model.layers.0.self_attn.q_proj.weight", z["grad_l0_q"])):
        assert np.linalg.norm(grads[k].numpy() - want) <= 2e-4 * np.linalg.norm(want), k
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)


def _long_case(kind):
    from _util import GOLDEN, spec_mod, weights_mod
    if kind == "pt":
        z = np.load(os.path.join(GOLDEN, "pt_tiny_long.npz"))
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    else:
        z = np.load(os.path.join(GOLDEN, "ft_tiny_long.npz"))
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_stack_long_pretrain_matches_reference():
    """config.stack_method = "long": 1 / (non-zero ids) embedding ratio + per-feature-level loss weights, on a batch with empty
    (0-valued) feature cells in real rows - loss and gradients of the reference."""
    z, spec, state, b = _long_case("pt")
    assert ((b["input_ids"][:, :, 1:] == 0) & (b["input_ids"][:, :, :1] != 0)).any()
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], stack_long=True)
    out, grads = O.loss_and_grads(fn, p, "head1_loss")
    assert abs(out["head1_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    for k, want in (("lm_head.weight", z["grad_lm_head"]), ("model.layers.0.self_attn.q_proj.weight", z["grad_l0_q"])):
        assert np.linalg.norm(grads[k].numpy() - want) <= 2e-4 * np.linalg.norm(want), k
    got = grads["model.embed_tokens.weight"].numpy()[:64]
    assert np.linalg.norm(got - z["grad_embed_rows"]) <= 2e-4 * np.linalg.norm(z["grad_embed_rows"])
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)
    # and it is not the "short" loss
    short = O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"])["head1_loss"].item()
    assert abs(short - float(z["loss"])) > 1e-3 * abs(float(z["loss"]))


def test_oracle_stack_long_finetune_matches_reference():
    z, spec, state, b = _long_case("ft")
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], stack_long=True)
    out, grads = O.loss_and_grads(fn, p, "task_loss")
    assert abs(out["task_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out["task_logits"].detach().numpy(), z["logits"], rtol=1e-4, atol=2e-5)
    got = grads["model.embed_tokens.weight"].numpy()[:64]
    assert np.linalg.norm(got - z["grad_embed_rows"]) <= 2e-4 * np.linalg.norm(z["grad_embed_rows"])
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)


def _tokence_case():
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "ft_tiny_tokence.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=7)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_token_ce_matches_reference():
    """config.loss_type = "token_ce" (node-level tasks): score + cross-entropy on every labelled row, all-row logits returned."""
    z, spec, state, b = _tokence_case()
    assert tuple(b["task_labels"].shape) == tuple(b["input_ids"].shape[:2]) and (b["task_labels"] == -100).any()
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], loss_type="token_ce")
    out, grads = O.loss_and_grads(fn, p, "task_loss")
    assert abs(out["task_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    assert tuple(out["task_logits"].shape) == tuple(z["logits"].shape)
    np.testing.assert_allclose(out["task_logits"].detach().numpy(), z["logits"], rtol=1e-4, atol=3e-5)
    for k, want in (("score.weight", z["grad_score"]), ("model.layers.1.mlp.down_proj.weight", z["grad_l1_down"])):
        assert np.linalg.norm(grads[k].numpy() - want) <= 2e-4 * np.linalg.norm(want), k
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)


def _roperange_case():
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, "ft_tiny_roperange.npz"))
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2,
                                   rope_range=float(z["rope_range"]))
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


def test_oracle_rope_range_matches_reference():
    """config.rope_range = 6: position ids rescaled per row to [0, 6) (fractional rotary positions), fine-tune model."""
    z, spec, state, b = _roperange_case()
    p = O.to_params(state, torch.float32)
    fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"])
    out, grads = O.loss_and_grads(fn, p, "task_loss")
    assert abs(out["task_loss"].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(out["task_logits"].detach().numpy(), z["logits"], rtol=1e-4, atol=3e-5)
    for k, want in (("model.layers.0.self_attn.q_proj.weight", z["grad_l0_q"]), ("model.layers.0.self_attn.k_proj.weight", z["grad_l0_k"])):
        assert np.linalg.norm(grads[k].numpy() - want) <= 2e-4 * np.linalg.norm(want), k
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)
    # ... and the rescaling matters on this batch
    import dataclasses
    plain = O.task_forward(dataclasses.replace(spec, rope_range=0.0), p, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"])
    assert abs(plain["task_loss"].item() - float(z["loss"])) > 1e-3 * abs(float(z["loss"]))


def _rawembed_case(kind):
    from _util import GOLDEN, spec_mod, weights_mod
    z = np.load(os.path.join(GOLDEN, f"{kind}_tiny_rawembed.npz"))
    E = int(z["embed_dim"])
    if kind == "pt":
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13, embed_dim=E)
    else:
        spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2, embed_dim=E)
    assert [int(x) for x in z["meta_spec"]] == list(spec.as_c_ints())
    seed, std, hstd = z["meta_init"]
    state = weights_mod.make_state_dict(spec, seed=int(seed), std=float(std), head_std=float(hstd))
    state["embed_layernorm.weight"] = z["w_embed_ln"]
    b = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, spec, state, b


@pytest.mark.parametrize("kind", ["pt", "ft"])
def test_oracle_raw_embeds_matches_reference(kind):
    """config.embed_dim = 64: raw-embedding inputs (mask-token blend in pre-training, RMSNorm, projection, sum with the token embeddings)."""
    z, spec, state, b = _rawembed_case(kind)
    p = O.to_params(state, torch.float32)
    if kind == "pt":
        fn = lambda q: O.pretrain_forward(spec, q, b["input_ids"], b["attention_mask"], b["labels"], inputs_raw_embeds=b["inputs_raw_embeds"])
        lk = "head1_loss"
    else:
        fn = lambda q: O.task_forward(spec, q, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                      inputs_raw_embeds=b["inputs_raw_embeds"])
        lk = "task_loss"
    out, grads = O.loss_and_grads(fn, p, lk)
    assert abs(out[lk].item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    pairs = [("embed_proj.weight", z["grad_embed_proj"]), ("embed_layernorm.weight", z["grad_embed_ln"])]
    if kind == "pt":
        pairs.append(("emb_mask_token", z["grad_mask_token"]))
    for k, want in pairs:
        got = grads[k].numpy().reshape(want.shape)
        assert np.linalg.norm(got - want) <= 2e-4 * np.linalg.norm(want), k
    np.testing.assert_allclose(np.array([float(grads[str(n)].norm()) for n in z["names"]]), z["grad_norms"], rtol=5e-4, atol=1e-7)
