"""Is the fine-tune loss deviation of the bf16 engine NOISE or BIAS?  (VERDICT r4 "What's weak" 1, "Next round" 3.)

The fine-tune fixtures are means over 2-32 pooled rows and sit 1e-3 ... 1.4e-2 from the fp32 loss under tolerances derived from the
reference's own bf16-vs-fp32 gap.  That argument needs evidence that the deviation is zero-mean rounding noise - it must shrink like
1/sqrt(rows) and carry no sign:

  (i)   C3 (BASELINE configs[3]: base model + LayerScale, S = 256, F = 4, V = 41245) at B = 64 and B = 128, forward, HIP vs oracle:
        loss held to 2e-3 / 1e-3 (first measurement, round 5: 1.02e-3 at B = 64, 1.0e-4 at B = 128; the 4-row groups of the same rows
        scatter with sigma 2.6e-3 ... 3.9e-3 around -8e-4 +- 6e-4 / -2e-4 +- 7e-4, i.e. around zero) - a mean over 64 / 128 pooled rows
        beats the 4-row fixture (1.05e-3 reference gap, 2.1e-3 measured at B = 8) as noise must;
  (ii)  a 64-seed sign test of (HIP loss - fp32 oracle loss) on ft_tiny_f4-shaped batches (and the 32-row shape): |mean| <= 3 sigma/sqrt(64);
  (iii) the same C3 rows split into groups of 4: the group errors scatter around zero with the spread the small fixtures show;
  (iv)  C3 in TRAINING mode (DropPath 0.2 + attention dropout 0.1, identical masks through the Python twins) at B = 32 next to its eval-mode
        twin on the same weights: the 2.3e-2 of the B = 4 check (std-0.04 weights) is the few-row scatter of that weight scale, not the dropouts.
Reference: /root/reference/src/models/graphgpt/modeling_finetune.py:167-234 (calculate_task_loss), :236-326 (forward)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from _util import ROOT, record_error, spec_mod, synth, tb, weights_mod
from oracle import gget_oracle as O

pytestmark = pytest.mark.gpu

eng_mod = importlib.import_module("graph-gpt_amd.engine")
L = importlib.import_module("graph-gpt_amd._lib")


def _bf16_weights(state):
    return {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}


def _dump(name, rec):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", name), "w") as fh:
        json.dump(rec, fh, indent=1)


def _c3(B, seed_w=9, seed_b=192, std=0.02, head_std=None, path_pdrop=0.0):
    S, F, V = 256, 4, 41245
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2,
                                   max_position=1024, layer_scale_init=1.0, path_pdrop=path_pdrop)
    state = weights_mod.make_state_dict(spec, seed=seed_w, std=std, head_std=head_std)
    batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=seed_b, lengths="uniform", min_len=S // 4)
    return spec, state, batch


def _oracle_task(spec, state_bf, b, chunk=16, **kw):
    """fp32 oracle forward in chunks of `chunk` samples (memory: [chunk,H,S,S] score tensors); logits [B, num_labels]"""
    p = O.to_params(state_bf, torch.float32, requires_grad=False)
    B = b["input_ids"].shape[0]
    outs = []
    with torch.no_grad():
        for a in range(0, B, chunk):
            sl = slice(a, min(B, a + chunk))
            kk = {k: (lambda f, sl: (lambda *args: f(*args)[sl]))(f, sl) for k, f in kw.items()}      # per-sample masks: this chunk's rows
            out = O.task_forward(spec, p, b["input_ids"][sl], b["attention_mask"][sl], b["position_ids"][sl], b["task_labels"][sl], **kk)
            outs.append(out["task_logits"].float())
    return torch.cat(outs)


_C3_POINTS = [(64, 2e-3), (128, 1e-3)] + [(int(x), 1e-3) for x in os.environ.get("GGET_C3_EXTRA_B", "").split(",") if x]   # (one-off B = 512 pass on file)


@pytest.mark.parametrize("B,tol", _C3_POINTS)
def test_c3_large_batch_loss_deviation_shrinks_like_noise(B, tol):
    spec, state, batch = _c3(B)
    b = tb(batch)
    e = eng_mod.Engine(spec, max_tokens=B * 256, max_batch=B)
    e.load_state_dict(state)
    loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL,
                                     num_tokens=int(batch["attention_mask"].sum()))
    torch.cuda.synchronize()
    want_logits = _oracle_task(spec, _bf16_weights(state), b)
    y = b["task_labels"]
    want = float(Fnn.cross_entropy(want_logits, y))
    got = float(loss)
    rel = abs(got - want) / abs(want)
    lg = logits.float().cpu()
    dlogit = float((lg - want_logits).abs().max())
    # per-row CE of both, groups of 4 rows = the size of the small fixtures
    ce_g, ce_w = Fnn.cross_entropy(lg, y, reduction="none"), Fnn.cross_entropy(want_logits, y, reduction="none")
    assert abs(float(ce_g.mean()) - got) <= 2e-5 * abs(got)          # the engine's loss IS the mean CE of its logits
    grp = ((ce_g - ce_w).view(-1, 4).mean(1) / ce_w.view(-1, 4).mean(1)).numpy()
    n = len(grp)
    mean, sd = float(grp.mean()), float(grp.std(ddof=1))
    rec = {"B": B, "loss_engine": got, "loss_oracle_fp32_on_bf16_weights": want, "loss_rel": rel, "logits_max_abs_dev": dlogit,
           "groups_of_4_rows": {"n": n, "mean_rel_err": mean, "std_rel_err": sd, "stderr_of_mean": sd / np.sqrt(n),
                                "max_abs_rel_err": float(np.abs(grp).max()), "positive_fraction": float((grp > 0).mean())}}
    _dump(f"parity_stats_c3_B{B}.json", rec)
    record_error(f"c3_S256_B{B}_forward", "loss_rel_vs_oracle", rel, tol)
    record_error(f"c3_S256_B{B}_forward", "task_logits_max_abs_vs_oracle", dlogit, 5e-2)
    record_error(f"c3_S256_B{B}_forward", f"groups_of_4_rows mean_rel_err (std {sd:.2e}, n {n})", abs(mean), 2.5 * sd / np.sqrt(n) + 1e-5)
    assert rel <= tol, rec
    assert dlogit <= 5e-2, rec           # (max over 2 B logits of |bf16 engine - fp32|: 0.030 at B = 64, 0.022 at B = 128; the fixtures' bound is 3 x the reference's own bf16 gap)
    assert abs(mean) <= 2.5 * sd / np.sqrt(n) + 1e-5, rec            # no sign: the group errors scatter around zero


@pytest.mark.parametrize("shape", ["f4", "f4_b32"])
def test_sign_test_of_finetune_loss_deviation(shape):
    """NSEED = 64 independent (weights, batch) draws of the fixture shape (tiny d128 / L2, F = 4, S = 24; B = 4 with standard init like ft_tiny_f4,
    B = 32 with the wide init of ft_tiny_f4_b32): d_i = (HIP loss - oracle loss) / oracle loss, oracle = fp32 arithmetic on the same bf16-rounded
    weights.  Fail if |mean d| > 3 sigma / sqrt(NSEED) (a bias shows as a mean that does not shrink with the number of draws; the first 16
    draws alone - round 5's first run - gave -2.6e-4 +- 1.2e-4 on the B = 4 shape, 2.1 sigma: not decidable at n = 16, hence 64)."""
    NSEED = int(os.environ.get("GGET_SIGN_TEST_SEEDS", "64"))     # (a one-off 512-draw pass is on file: profiles/r05_parity_stats_sign_test_*_n512.json)
    B, std, head_std = (4, 0.02, None) if shape == "f4" else (32, 0.06, 0.15)
    S, F, V = 24, 4, 1000
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2)
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    ds, dl = [], []
    for i in range(NSEED):
        state = weights_mod.make_state_dict(spec, seed=1000 + i, std=std, head_std=head_std)
        batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=2000 + i)
        b = tb(batch)
        e.load_state_dict(state)
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL)
        want_logits = _oracle_task(spec, _bf16_weights(state), b, chunk=B)
        want = float(Fnn.cross_entropy(want_logits, b["task_labels"]))
        ds.append((float(loss) - want) / want)
        dl.append(float((logits.float().cpu() - want_logits).abs().max()))
    ds = np.asarray(ds)
    mean, sd = float(ds.mean()), float(ds.std(ddof=1))
    bound = 3 * sd / np.sqrt(NSEED)
    rec = {"shape": shape, "B": B, "n_seeds": NSEED, "rel_dev_by_seed": ds.tolist(), "mean": mean, "std": sd, "stderr_of_mean": sd / np.sqrt(NSEED),
           "three_sigma_over_sqrt_n": bound, "mean_first_16": float(ds[:16].mean()), "max_abs": float(np.abs(ds).max()),
           "positive": int((ds > 0).sum()), "logits_max_abs_dev_max": max(dl)}
    _dump(f"parity_stats_sign_test_{shape}.json" if NSEED == 64 else f"parity_stats_sign_test_{shape}_n{NSEED}.json", rec)
    record_error(f"ft_tiny_{shape}_{NSEED}_seed_sign_test", f"abs_mean_rel_loss_dev (std {sd:.2e}, +{rec['positive']}/{NSEED})", abs(mean), bound)
    record_error(f"ft_tiny_{shape}_{NSEED}_seed_sign_test", "max_abs_rel_loss_dev", float(np.abs(ds).max()), 2e-2)
    assert abs(mean) <= bound, rec
    assert np.abs(ds).max() <= 2e-2, rec


def test_sign_test_of_s2048_loss_deviation():
    """`ft_base_s2048` (base model, S = 2048, a TWO-row mean) sits at 95 % of its derived tolerance, and the sign tests
    above cover the tiny model only.  Here: NSEED = 32 independent (weights, batch) draws of that fixture's shape - base d768 / L12, F = 4,
    V = 41245, B = 2 full-length rows - forward, HIP loss against the fp32 oracle on the same bf16-rounded weights.  The deviation must
    carry no sign (|mean d| <= 3 sigma / sqrt(NSEED)), and the fixture's own 1.4e-2 must be an ordinary draw of the measured scatter
    (<= 3.5 sigma): it is a two-row coin flip, not a bias of the long-sequence kernels.  The per-seed logits are held to the fixtures' bound."""
    # (the 32-draw pass with both oracle passes takes 12 minutes of host time - 64 fp32 forwards of the base model at S = 2048 - and is on file:
    #  profiles/r06_parity_stats_sign_test_s2048_n32.json: mean -1.3e-3 +- 1.2e-3 (14 / 32 positive) against bf16-rounded weights, -1.2e-3 +-
    #  1.4e-3 (15 / 32) against fp32 weights, sigma 6.8e-3 / 7.9e-3, the fixture's 1.4e-2 = 1.8 sigma.  The suite's default is 8 draws, one pass.)
    NSEED = int(os.environ.get("GGET_S2048_SIGN_TEST_SEEDS", "8"))
    both = NSEED >= 32
    B, S, F, V = 2, 2048, 4, 41245
    spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_TASK, vocab_size=V, stacked_feat=F, next_n_token=1, num_labels=2, max_position=2048)
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    ds, dl, df = [], [], []
    for i in range(NSEED):
        state = weights_mod.make_state_dict(spec, seed=3000 + i)
        batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=4000 + i, lengths="full")
        b = tb(batch)
        e.load_state_dict(state)
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None, L.PROBLEM_SINGLE_LABEL)
        want_logits = _oracle_task(spec, _bf16_weights(state), b, chunk=1)
        want = float(Fnn.cross_entropy(want_logits, b["task_labels"]))
        ds.append((float(loss) - want) / want)
        dl.append(float((logits.float().cpu() - want_logits).abs().max()))
        # ... and against fp32 arithmetic on the UN-rounded weights - what the fixture compares with (the reference's fp32 run): the bf16
        # rounding of the weights is half of the deviation there (cast-point model)
        full = float(Fnn.cross_entropy(_oracle_task(spec, state, b, chunk=1), b["task_labels"])) if both else want
        df.append((float(loss) - full) / full)
    ds, df = np.asarray(ds), np.asarray(df)
    mean, sd = float(ds.mean()), float(ds.std(ddof=1))
    mean_f, sd_f = float(df.mean()), float(df.std(ddof=1))
    bound, bound_f = 3 * sd / np.sqrt(NSEED), 3 * sd_f / np.sqrt(NSEED)
    fixture_dev = 1.404e-2          # ft_base_s2048: |HIP - reference fp32| / reference fp32 (profiles/r05_parity_errors.json)
    rec = {"shape": "base d768 / L12, S 2048, B 2, F 4, V 41245", "n_seeds": NSEED,
           "vs_fp32_on_bf16_rounded_weights": {"rel_dev_by_seed": ds.tolist(), "mean": mean, "std": sd, "stderr_of_mean": sd / np.sqrt(NSEED),
                                               "three_sigma_over_sqrt_n": bound, "max_abs": float(np.abs(ds).max()), "positive": int((ds > 0).sum())},
           "vs_fp32_on_fp32_weights": {"rel_dev_by_seed": df.tolist(), "mean": mean_f, "std": sd_f, "stderr_of_mean": sd_f / np.sqrt(NSEED),
                                       "three_sigma_over_sqrt_n": bound_f, "max_abs": float(np.abs(df).max()), "positive": int((df > 0).sum())},
           "logits_max_abs_dev_max": max(dl), "fixture_ft_base_s2048_dev": fixture_dev, "fixture_dev_in_sigmas": fixture_dev / sd_f}
    rec["second_oracle_pass_on_fp32_weights"] = both
    _dump("parity_stats_sign_test_s2048.json" if NSEED == 8 else f"parity_stats_sign_test_s2048_n{NSEED}.json", rec)
    tag = f"ft_base_s2048_{NSEED}_seed_sign_test"
    record_error(tag, f"abs_mean_rel_loss_dev vs fp32 on bf16-rounded weights (std {sd:.2e}, +{int((ds > 0).sum())}/{NSEED})", abs(mean), bound)
    record_error(tag, f"abs_mean_rel_loss_dev vs fp32 on fp32 weights (std {sd_f:.2e}, +{int((df > 0).sum())}/{NSEED})", abs(mean_f), bound_f)
    record_error(tag, "fixture_dev_in_sigmas_of_the_two_row_scatter", fixture_dev / sd_f, 3.5)
    record_error(tag, "task_logits_max_abs_vs_oracle (max over seeds)", max(dl), 5e-2)
    assert abs(mean) <= bound and abs(mean_f) <= bound_f, rec
    assert fixture_dev <= (3.5 if both else 5.0) * sd_f, rec      # (8 draws estimate sigma to +-25 %)
    assert max(dl) <= 5e-2, rec


def test_c3_training_mode_dropouts_large_batch_and_eval_twin():
    """The B = 4 check of test_gpu_model.py::test_c3_training_mode_dropouts_exact_mask measures 2.3e-2 with identical masks.  Same weights
    (std 0.04, head 0.1: twice the standard init, logits of +-3), B = 32: (a) eval mode, (b) training mode with both masks handed to
    the oracle.  Measured (round 5): eval 9.3e-3, training 6.5e-3; the logits deviate by up to 0.28 of a +-7 range and the 4-row
    groups scatter with sigma ~0.09 (single groups up to 0.2) - weights at twice the standard init under LayerScale 1 amplify the bf16
    rounding of the residual stream to 4 % of the logit range, so ONE 4-row group landing at 2.3e-2 is an ordinary draw, with or without
    the dropouts.  Held: both 32-row means under 1.5e-2, the training-mode one no worse than 2 x the eval one + 2e-3."""
    from test_gpu_model import _attn_drop_keep, _path_keep
    B, S, seed, p_attn, p_path = 32, 256, 4242, 0.1, 0.2
    spec, state, batch = _c3(B, seed_w=9, seed_b=94, std=0.04, head_std=0.1, path_pdrop=p_path)
    b = tb(batch)
    y = b["task_labels"]
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)
    sb = _bf16_weights(state)
    res = {}
    for mode in ("eval", "train"):
        e.set_dropout(p_attn if mode == "train" else 0.0, p_path if mode == "train" else 0.0, seed)
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], y, None, L.PROBLEM_SINGLE_LABEL)
        torch.cuda.synchronize()
        kw = {}
        if mode == "train":
            L_, H = spec.num_layers, spec.num_heads
            cache = {}

            def attn_keep(l):
                if l not in cache:
                    cache.clear()           # (one layer's [B,H,S,S] mask at a time)
                    cache[l] = _attn_drop_keep((seed + 0x9E37 * l) & 0xFFFFFFFF, B, H, S, p_attn)
                return cache[l]
            kw = dict(path_mult=lambda l, w: _path_keep(seed, l, w, B, p_path * l / (L_ - 1)), attn_keep=attn_keep)
        want_logits = _oracle_task(spec, sb, b, chunk=8, **kw)
        want = float(Fnn.cross_entropy(want_logits, y))
        lg = logits.float().cpu()
        ce_g, ce_w = Fnn.cross_entropy(lg, y, reduction="none"), Fnn.cross_entropy(want_logits, y, reduction="none")
        grp = ((ce_g - ce_w).view(-1, 4).mean(1) / ce_w.view(-1, 4).mean(1)).numpy()
        res[mode] = {"loss_engine": float(loss), "loss_oracle": want, "rel": abs(float(loss) - want) / want,
                     "logits_max_abs_dev": float((lg - want_logits).abs().max()), "logits_abs_max": float(want_logits.abs().max()),
                     "groups_of_4_rel_err": grp.tolist()}
    _dump("parity_stats_c3_train_vs_eval_B32.json", res)
    record_error("c3_train_mode_dropouts_B32", "eval loss_rel_vs_oracle", res["eval"]["rel"], 1.5e-2)
    record_error("c3_train_mode_dropouts_B32", "train loss_rel_vs_oracle_same_masks", res["train"]["rel"], 1.5e-2)
    for mode in ("eval", "train"):
        g_ = np.asarray(res[mode]["groups_of_4_rel_err"])
        record_error("c3_train_mode_dropouts_B32", f"{mode}: std of the 4-row group errors (max |err| {np.abs(g_).max():.2e})", float(g_.std(ddof=1)), float("nan"))
    assert res["eval"]["rel"] <= 1.5e-2 and res["train"]["rel"] <= 1.5e-2, res
    assert res["train"]["rel"] <= 2 * res["eval"]["rel"] + 2e-3, res


def test_training_curve_tracks_the_oracle_over_many_steps():
    """The SMTP loss CURVE, not single steps: 40 clip + AdamW steps (GGET_CURVE_STEPS) of the small pre-train model (d 256 / L 4, F = 13, V = 756,
    the C1 tokenisation) over eight batches in rotation, HIP engine (bf16 compute copy, fp32 master, var-len rows) against the oracle's
    fp32 training with the same AdamW restatement (bf16-rounded weights in its forward: the engine's cast point).  The two trajectories
    see different rounding (activations, gradients) at every step, so they separate slowly; held: every step's loss within 1 % of the
    oracle's to 2e-3 (measured 3.6e-4 at the worst step), the mean signed gap within 3 standard errors of zero (no drift), and the weight UPDATE
    (final - initial master weights) of the two runs pointing the same way (cosine >= 0.95; element-wise Adam amplifies gradient rounding,
    see the comment at the assert)."""
    steps = int(os.environ.get("GGET_CURVE_STEPS", "40"))
    B, S, F, V = 32, 32, 13, 756
    spec = spec_mod.spec_from_size("mini", kind=spec_mod.KIND_PRETRAIN, vocab_size=V, stacked_feat=F, next_n_token=F)
    state = weights_mod.make_state_dict(spec, seed=11, std=0.02)
    batches = [synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=500 + i) for i in range(8)]
    lr, b1, b2, eps, wd, clip = 1e-3, 0.9, 0.95, 1e-8, 0.1, 1.0
    e = eng_mod.Engine(spec, max_tokens=B * S, max_batch=B)
    e.load_state_dict(state)
    p = O.to_params(state, torch.float32)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    got, want = [], []
    for it in range(steps):
        batch = batches[it % len(batches)]
        b = tb({k: x for k, x in batch.items() if k != "lengths"})
        loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], num_tokens=int(batch["attention_mask"].sum()))
        e.backward()
        e.adamw_step(lr, b1, b2, eps, wd, clip)
        got.append(float(loss))

        def fn(q):
            qb = {k: (x.detach().to(torch.bfloat16).float() - x.detach() + x) for k, x in q.items()}     # bf16 compute copy, straight-through
            return O.pretrain_forward(spec, qb, b["input_ids"], b["attention_mask"], b["labels"])
        out, grads = O.loss_and_grads(fn, p, "head1_loss")
        with torch.no_grad():
            O.adamw_step(p, grads, m, v, it + 1, lr, b1, b2, eps, wd, clip)
        want.append(float(out["head1_loss"].detach()))
    got, want = np.asarray(got), np.asarray(want)
    rel = (got - want) / want
    mean, sd = float(rel.mean()), float(rel.std(ddof=1))
    # where the weights went: update = final - initial master weights, engine vs oracle (whole model and the worst tensor; tensors whose
    # update is tiny against the model's largest are measured against that floor)
    upd_g = {k: e.view(k, "master").float().cpu().numpy().ravel() - np.asarray(state[k], np.float32).ravel() for k in state}
    upd_w = {k: p[k].detach().numpy().ravel() - np.asarray(state[k], np.float32).ravel() for k in state}
    umax = max(float(np.linalg.norm(x)) for x in upd_w.values())
    per = {k: float(np.linalg.norm(upd_g[k] - upd_w[k])) / max(float(np.linalg.norm(upd_w[k])), 1e-2 * umax) for k in state}
    worst = max(per, key=per.get)
    tot = float(np.sqrt(sum(float(np.sum((upd_g[k] - upd_w[k]) ** 2)) for k in state)) / np.sqrt(sum(float(np.sum(upd_w[k] ** 2)) for k in state)))
    dot = sum(float(np.dot(upd_g[k], upd_w[k])) for k in state)
    cos = dot / np.sqrt(sum(float(np.sum(upd_g[k] ** 2)) for k in state) * sum(float(np.sum(upd_w[k] ** 2)) for k in state))
    rec = {"steps": steps, "loss_engine": got.tolist(), "loss_oracle": want.tolist(), "rel_gap_max_abs": float(np.abs(rel).max()),
           "rel_gap_mean": mean, "rel_gap_std": sd, "loss_first": float(want[0]), "loss_last": float(want[-1]),
           "weight_update_cosine_whole_model": float(cos), "weight_update_rel_l2_whole_model": tot, "weight_update_rel_l2_worst_tensor": per[worst], "worst_tensor": worst}
    _dump("parity_stats_training_curve.json", rec)
    record_error("pt_mini_training_curve", f"max_abs_rel_loss_gap_over_{steps}_steps", rec["rel_gap_max_abs"], 1e-2)
    record_error("pt_mini_training_curve", "abs_mean_rel_loss_gap", abs(mean), 3 * sd / np.sqrt(steps) + 1e-5)
    record_error("pt_mini_training_curve", "one_minus_weight_update_cosine", 1.0 - float(cos), 5e-2)
    assert want[-1] < want[0] - 0.1, rec                         # the curve actually goes somewhere
    assert rec["rel_gap_max_abs"] <= 2e-3, rec                   # (measured 3.6e-4)
    assert abs(mean) <= 3 * sd / np.sqrt(steps) + 1e-5, rec
    # Adam normalises every element's step to ~lr whatever the size of its gradient: elements whose gradient is at the level of the bf16
    # rounding of the gradient array move in rounding-decided directions in BOTH runs (the reference's bf16 training has the same property
    # against fp32), so the element-wise update differs by ~0.18 rel-L2 after 40 steps while the direction of the whole update agrees
    assert cos >= 0.95 and tot <= 0.35, rec
