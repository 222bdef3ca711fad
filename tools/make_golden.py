#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REAL reference (alibaba/graph-gpt at /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Fixtures hold data
only: seeded inputs produced by our own generators (graph-gpt_amd/synth.py, weights.py) and the
reference's outputs on them (losses, logits, hidden-state slices, per-parameter gradient norms,
AdamW trajectories, LR-schedule samples).  Import recipe = SURVEY.md Appendix A.

    python tools/make_golden.py            # rewrites every fixture
"""
import importlib
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
gg = importlib.import_module("graph-gpt_amd")
from importlib import import_module  # noqa: E402

spec_mod = import_module("graph-gpt_amd.spec")
weights_mod = import_module("graph-gpt_amd.weights")
synth = import_module("graph-gpt_amd.synth")


# ----------------------------------------------------------------------------- reference import
def import_reference():
    sys.path.insert(0, REF)

    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            full = self.__name__ + "." + name
            return sys.modules.get(full) or type(name, (), {"__init__": lambda self, *a, **k: None})

    def stub(name):
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            n = ".".join(parts[:i])
            if n not in sys.modules:
                m = _Stub(n)
                m.__path__ = []
                m.__spec__ = importlib.machinery.ModuleSpec(n, None, is_package=True)
                sys.modules[n] = m

    def bare_pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m

    # our repo also has a top-level `src` package (the drop-in surface); make sure the reference's wins here
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    bare_pkg("src", REF + "/src")
    bare_pkg("src.utils", REF + "/src/utils")
    for n in ["torch_geometric", "torch_geometric.data", "ogb", "ogb.utils", "ogb.utils.features",
              "omegaconf", "timm", "timm.utils", "timm.models", "timm.utils.model"]:
        stub(n)
    sys.modules["omegaconf"].MISSING = "???"
    import transformers.utils.import_utils as iu
    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    from src.models import GraphGPTPretrainBase, GraphGPTTaskModel, GraphGPTConfig
    from src.models.graphgpt import utils_graphgpt
    # transformers>=5 decoder layers must return a tensor; the reference's LayerScale layer returns a tuple
    _orig = utils_graphgpt.LlamaDecoderLayer.forward

    def _fwd(self, *a, **k):
        return _orig(self, *a, **k)[0]

    utils_graphgpt.LlamaDecoderLayer.forward = _fwd
    return GraphGPTPretrainBase, GraphGPTTaskModel, GraphGPTConfig


def ref_config(GraphGPTConfig, spec, **extra):
    kw = dict(vocab_size=spec.vocab_size, hidden_size=spec.hidden_size,
              intermediate_size=spec.intermediate_size, num_hidden_layers=spec.num_layers,
              num_attention_heads=spec.num_heads, head_dim=spec.head_dim, hidden_act="gelu",
              max_position_embeddings=spec.max_position, rms_norm_eps=spec.rms_eps,
              causal_attention=spec.causal, stacked_feat=spec.stacked_feat, stack_method="short",
              stacked_feat_agg_method="gated" if spec.gated_agg else "sum",
              next_n_token=spec.next_n_token, attention_dropout=0.0, use_cache=False,
              layer_scale_init_value=spec.layer_scale_init, pad_token_id=0, tie_word_embeddings=False)
    kw.update(extra)
    return GraphGPTConfig(**kw)


def load_weights(model, state):
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "rotary_emb" not in m]
    assert not missing and not unexpected, (missing, unexpected)


def grad_norms(model, names):
    g = dict(model.named_parameters())
    return np.array([float(g[n].grad.float().norm()) if g[n].grad is not None else 0.0 for n in names], np.float64)


CASES = [
    # name, kind, spec kwargs, batch kwargs, init kwargs
    ("pt_tiny_f13_a", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=4, S=24, seed=0), {}),
    ("pt_tiny_f13_b", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=2, S=40, seed=1), {}),
    ("pt_tiny_f1", "pt", dict(vocab_size=300, stacked_feat=1, next_n_token=1), dict(B=4, S=24, seed=2, lengths="uniform"), {}),
    ("pt_tiny_causal", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13, causal=True), dict(B=4, S=24, seed=3), {}),
    ("pt_tiny_gated", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13, gated_agg=True), dict(B=4, S=24, seed=4), {}),
    ("pt_tiny_wgt", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=4, S=24, seed=5, dlm_wgt=True), {}),
    ("pt_tiny_bigw", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=4, S=24, seed=6),
     dict(std=0.06, head_std=0.15)),
    ("pt_tiny_s72", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=3, S=72, seed=7, lengths="uniform", min_len=20),
     dict(std=0.06, head_std=0.15)),
    ("pt_tiny_packed", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=3, S=72, seed=14, packed=True),
     dict(std=0.06, head_std=0.15)),
    ("ft_tiny_f4", "ft", dict(vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2), dict(B=4, S=24, seed=8), {}),
    ("ft_tiny_ls", "ft", dict(vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2, layer_scale_init=1.0),
     dict(B=4, S=40, seed=9), dict(std=0.06, head_std=0.15)),
    ("ft_tiny_reg", "ft", dict(vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=1, score_bias=True),
     dict(B=4, S=24, seed=12, regression=True), dict(std=0.06, head_std=0.15)),
    ("ft_tiny_ml", "ft", dict(vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=5),
     dict(B=6, S=24, seed=13, multi_label=True), dict(std=0.06, head_std=0.15)),
    # round 2: larger fine-tune batches (the 2-class CE of 4 samples is too coarse a probe), MSE regression, sample-weighted CE
    ("ft_tiny_f4_b32", "ft", dict(vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2), dict(B=32, S=24, seed=15),
     dict(std=0.06, head_std=0.15)),
    ("ft_tiny_mse", "ft", dict(vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=1, score_bias=True),
     dict(B=32, S=24, seed=16, regression=True, mse=True), dict(std=0.06, head_std=0.15)),
    ("ft_tiny_wce", "ft", dict(vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2),
     dict(B=32, S=24, seed=17, sample_wgt=True), dict(std=0.06, head_std=0.15)),
    # round 2: the full-width base model (d768 / L12) of the headline run at a small batch, big weights (loss far from
    # ln V, attention far from uniform) - size "base"; only norms and slices of the gradients are stored
    ("pt_base_bigw", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=4, S=32, seed=18),
     dict(std=0.06, head_std=0.15, size="base")),
    ("pt_base_std", "pt", dict(vocab_size=756, stacked_feat=13, next_n_token=13), dict(B=4, S=32, seed=19),
     dict(size="base")),
    # round 3: the fine-tune configurations of BASELINE.json at full width and full sequence length, small batch - C3 (ogbl-ppa: base
    # model + LayerScale, S = 256, V = 41245) and C4 (S = 2048) - so that their loss tolerances derive from the reference's own
    # bf16-vs-fp32 gap instead of a constant
    ("ft_base_ls_s256", "ft", dict(vocab_size=41245, stacked_feat=4, next_n_token=1, num_labels=2, layer_scale_init=1.0),
     dict(B=4, S=256, seed=20, lengths="uniform", min_len=64), dict(size="base")),
    ("ft_base_s2048", "ft", dict(vocab_size=41245, stacked_feat=4, next_n_token=1, num_labels=2, max_position=2048),
     dict(B=2, S=2048, seed=21, lengths="uniform", min_len=1024), dict(size="base")),
]

ADAM = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
CLIP = 1.0


def run_case(name, kind, skw, bkw, ikw, classes):
    PT, FT, Cfg = classes
    size = ikw.pop("size", "tiny")
    big = size != "tiny"
    spec = spec_mod.spec_from_size(size, kind=spec_mod.KIND_PRETRAIN if kind == "pt" else spec_mod.KIND_TASK, **skw)
    state = weights_mod.make_state_dict(spec, seed=100 + bkw["seed"], **ikw)
    names = list(state.keys())
    out = {}
    if kind == "pt":
        if bkw.pop("packed", False):
            batch = synth.make_packed_pretrain_batch(F=spec.stacked_feat, V=spec.vocab_size, **bkw)
        else:
            batch = synth.make_pretrain_batch(F=spec.stacked_feat, V=spec.vocab_size, **bkw)
        cfg = ref_config(Cfg, spec)
        model = PT(cfg)
    else:
        reg = bkw.pop("regression", False)
        ml = bkw.pop("multi_label", False)
        mse = bkw.pop("mse", False)
        swgt = bkw.pop("sample_wgt", False)
        batch = synth.make_task_batch(F=spec.stacked_feat, V=spec.vocab_size, num_labels=spec.num_labels,
                                      regression=reg, multi_label=ml, **bkw)
        if swgt:   # per-sample weights of the sample-weighted CE (modeling_finetune.py:215-226)
            batch["wgt"] = np.random.RandomState(1000 + bkw["seed"]).uniform(0.2, 3.0, size=(bkw["B"],)).astype(np.float32)
        extra = dict(num_labels=spec.num_labels, loss_type=("l1" if reg and not mse else None), mlp=[],
                     problem_type="regression" if reg else ("multi_label_classification" if ml else "single_label_classification"))
        cfg = ref_config(Cfg, spec, **extra)
        model = FT(cfg)
    load_weights(model, state)
    model.eval()  # dropout-free arithmetic (attention_dropout=0 anyway; DropPath = identity)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}

    def fwd(m, dtype=None):
        if kind == "pt":
            return m(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"],
                     inputs_raw_embeds=None, sample_wgt=tb.get("wgt"))
        return m(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"],
                 position_ids=tb["position_ids"], task_labels=tb["task_labels"], sample_wgt=tb.get("wgt"))

    o = fwd(model)
    loss = o.head1_loss if kind == "pt" else o.task_loss
    logits = o.head1_logits if kind == "pt" else o.task_logits
    model.zero_grad()
    loss.backward()
    out["loss"] = np.float64(loss.item())
    if kind == "ft" and spec.num_labels == 1 and cfg.loss_type == "l1":
        # L1 regression: the gradient is sign(pred - y); keep every sample far from the kink so that bf16-level
        # differences in `pred` cannot flip a sign (the parity tests compare gradients)
        margin = (logits.detach().float().view(-1) - tb["task_labels"].float().view(-1)).abs().min().item()
        assert margin > 0.15, f"{name}: |pred-y| margin {margin} too small, pick another seed"
    out["logits"] = logits.detach().float().numpy()[:64]
    out["logits_shape"] = np.array(logits.shape)
    out["grad_norms"] = grad_norms(model, names)
    # one full gradient for a mid-stack matrix and the embedding (tight check of backward)
    g = dict(model.named_parameters())
    if not big:
        out["grad_embed"] = g["model.embed_tokens.weight"].grad.numpy().copy()
        out["grad_l0_q"] = g["model.layers.0.self_attn.q_proj.weight"].grad.numpy().copy()
        out["grad_l1_down"] = g["model.layers.1.mlp.down_proj.weight"].grad.numpy().copy()
    else:   # full-width model: 64 x 64 corner blocks of a few matrices (q / k of the first and last layer, a down projection)
        for tag, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l0_k", "model.layers.0.self_attn.k_proj.weight"),
                        ("l11_q", "model.layers.11.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                        ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
            out["gradblk_" + tag] = g[pn].grad.numpy()[:64, :64].copy()
    # final hidden states
    with torch.no_grad():
        if kind == "pt":
            ids3, emb, _ = model.prepare_inputs_embeds(tb["input_ids"], None, None, tb["labels"])
        else:
            ids3, emb, _ = model.prepare_inputs_embeds(tb["input_ids"][:, :, : spec.stacked_feat], None, None)
        out["embeds"] = emb.numpy()[:, :4, :16].copy()
        if kind == "ft":
            out["task_hidden"] = o.task_hidden_states.detach().numpy().copy()
    # bf16 module path of the reference itself
    mb = (PT if kind == "pt" else FT)(cfg)
    load_weights(mb, state)
    mb = mb.to(torch.bfloat16).eval()
    with torch.no_grad():
        ob = fwd(mb)
    lb = ob.head1_loss if kind == "pt" else ob.task_loss
    out["loss_bf16"] = np.float64(lb.item())
    if big:   # the reference's own bf16 backward: how far a bf16 implementation's gradients sit from the fp32 ones
        mb.zero_grad()
        ob2 = fwd(mb)
        (ob2.head1_loss if kind == "pt" else ob2.task_loss).backward()
        gb = dict(mb.named_parameters())
        for tag, pn in (("l0_q", "model.layers.0.self_attn.q_proj.weight"), ("l0_k", "model.layers.0.self_attn.k_proj.weight"),
                        ("l11_q", "model.layers.11.self_attn.q_proj.weight"), ("l11_k", "model.layers.11.self_attn.k_proj.weight"),
                        ("l5_down", "model.layers.5.mlp.down_proj.weight"), ("l5_gate", "model.layers.5.mlp.gate_proj.weight")):
            out["gradblk_bf16_" + tag] = gb[pn].grad.float().numpy()[:64, :64].copy()
        out["grad_norms_bf16"] = np.array([float(gb[n].grad.float().norm()) if gb[n].grad is not None else 0.0 for n in names], np.float64)
    out["logits_bf16"] = (ob.head1_logits if kind == "pt" else ob.task_logits).float().numpy()[:64]
    # three clip+AdamW steps on the same batch (training_utils.py:53-86 order: clip then step)
    model.zero_grad()
    opt = torch.optim.AdamW(model.parameters(), **ADAM)
    traj = []
    gn = []
    for _ in range(3):
        opt.zero_grad()
        o = fwd(model)
        l_ = o.head1_loss if kind == "pt" else o.task_loss
        l_.backward()
        gn.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)))
        opt.step()
        traj.append(l_.item())
    with torch.no_grad():
        o = fwd(model)
    traj.append((o.head1_loss if kind == "pt" else o.task_loss).item())
    out["adamw_losses"] = np.array(traj, np.float64)
    out["adamw_gnorms"] = np.array(gn, np.float64)
    sd = model.state_dict()
    out["adamw_final_norms"] = np.array([float(sd[n].float().norm()) for n in names], np.float64)
    # meta (so the tests rebuild inputs from the same generators and cross-check them against the stored copy)
    for k, v in batch.items():
        out["in_" + k] = v
    out["meta_spec"] = np.array(spec.as_c_ints(), np.int64)
    out["meta_layer_scale"] = np.float64(spec.layer_scale_init)
    out["meta_init"] = np.array([ikw.get("std", 0.02), ikw.get("head_std", -1.0), 100 + bkw["seed"]], np.float64)
    out["meta_size"] = np.array(size)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(f"{name}: loss {out['loss']:.6f} bf16 {out['loss_bf16']:.6f} logits {tuple(out['logits_shape'])} "
          f"adamw {np.round(out['adamw_losses'], 5)}")


def lr_fixture():
    """OneCycleLR samples with the parameters `_py_one_cycle` (reference loss_utils.py:322-367) sets."""
    res = {}
    for tag, (max_lr, min_lr, total, warm) in {"a": (3e-4, 0.0, 1000, 100), "b": (1e-3, 1e-5, 50, 5)}.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=max_lr)
        div = 25.0
        initial = max_lr / div
        fdf = initial / min_lr if min_lr > 0 else 1e4
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=[max_lr], total_steps=total + 1,
                                                   pct_start=warm / total, anneal_strategy="cos",
                                                   cycle_momentum=False, div_factor=div, final_div_factor=fdf)
        lrs = []
        for _ in range(total):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        res["onecycle_" + tag] = np.array(lrs, np.float64)
        res["onecycle_" + tag + "_params"] = np.array([max_lr, min_lr, total, warm], np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lr_schedules.npz"), **res)
    print("lr_schedules written")


def smtp2d_fixture():
    """In-model SMTP masking: the reference function run under a recording torch.rand / torch.randn, so the fixture holds
    the draws (in call order) next to inputs and outputs."""
    import_reference()
    helpers = sys.modules["src.models.graphgpt.modeling_helpers"]
    res = {}
    cases = {"a": dict(B=5, S=24, F=13, V=756, smtp_2d_rate=1.0, power=1.0, replace_rate=0.0, global_2d_mask=False),
             "b": dict(B=4, S=40, F=4, V=300, smtp_2d_rate=0.5, power=2.0, replace_rate=0.3, global_2d_mask=False),
             "c": dict(B=3, S=16, F=1, V=97, smtp_2d_rate=0.3, power=0.5, replace_rate=0.5, global_2d_mask=True)}
    for tag, c in cases.items():
        g = torch.Generator().manual_seed(100 + ord(tag))
        B, S, F, V = c["B"], c["S"], c["F"], c["V"]
        lens = torch.randint(S // 2, S + 1, (B,), generator=g)
        ids = torch.randint(2, V, (B, S, F), generator=g)
        node_idx = torch.zeros(B, S, dtype=torch.long)
        for b in range(B):
            ids[b, lens[b]:] = 0
            # Eulerian-path style node indices: tokens revisit nodes, values < number of distinct nodes <= len
            nn_ = max(2, int(lens[b]) * 2 // 3)
            node_idx[b, : lens[b]] = torch.randint(0, nn_, (int(lens[b]),), generator=g)
        rec = []
        real_rand, real_randn = torch.rand, torch.randn

        def rand(*a, **k):
            k.pop("device", None)
            t = real_rand(*a, generator=g, **k)
            rec.append(t.clone())
            return t

        def randn(*a, **k):
            k.pop("device", None)
            t = real_randn(*a, generator=g, **k)
            rec.append(t.clone())
            return t

        torch.rand, torch.randn = rand, randn
        try:
            out_ids, out_lab = helpers.prepare_for_2d_smtp_inputs_labels(
                ids.clone(), node_idx, smtp_2d_rate=c["smtp_2d_rate"], power=c["power"], replace_rate=c["replace_rate"],
                vocab=V, global_2d_mask=c["global_2d_mask"])
        finally:
            torch.rand, torch.randn = real_rand, real_randn
        assert len(rec) == 5, len(rec)
        res.update({f"{tag}_ids": ids.numpy(), f"{tag}_node_idx": node_idx.numpy(),
                    f"{tag}_u_sample": rec[0].reshape(B).numpy(), f"{tag}_u_rate": rec[1].reshape(B).numpy(),
                    f"{tag}_u_cell": rec[2].numpy(), f"{tag}_token_shift": (rec[3] * 10).numpy(), f"{tag}_u_replace": rec[4].numpy(),
                    f"{tag}_out_ids": out_ids.numpy(), f"{tag}_out_labels": out_lab.numpy(),
                    f"{tag}_params": np.array([c["smtp_2d_rate"], c["power"], c["replace_rate"], V, int(c["global_2d_mask"])], np.float64)})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "smtp2d.npz"), **res)
    print("smtp2d written")


def generation_fixture():
    """Deterministic MaskGIT-style generation: the reference `sample_per_batch` (src/utils/generation_utils.py:84-237) driven
    by the reference tiny pre-train model on CPU (fp32), one run per confidence algorithm."""
    import types
    classes = import_reference()
    PT, FT, Cfg = classes
    import importlib
    gen = importlib.import_module("src.utils.generation_utils")
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=300, stacked_feat=4, next_n_token=4)
    state = weights_mod.make_state_dict(spec, seed=321, std=0.08, head_std=0.2)
    model = PT(ref_config(Cfg, spec))
    load_weights(model, state)
    model.eval()
    batch = synth.make_pretrain_batch(B=3, S=16, F=4, V=300, seed=77)
    ids = torch.from_numpy(batch["input_ids"])
    att = torch.from_numpy(batch["attention_mask"])
    res = {"in_input_ids": batch["input_ids"], "in_attention_mask": batch["attention_mask"],
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([0.08, 0.2, 321], np.float64)}
    for alg in ("maskgit_plus", "topk_margin", "entropy"):
        cfg = types.SimpleNamespace(eps=1e-3, steps=6, mask_token_id=1, output_history=True, temperature=0.0, top_p=None,
                                    top_k=None, alg=alg, alg_temp=None)
        x, hist = gen.sample_per_batch(model, cfg, input_ids=ids.clone(), attention_mask=att, inputs_raw_embeds=None)
        res[f"{alg}_x"] = x.numpy()
        res[f"{alg}_hist"] = np.stack([h.numpy() for h in hist])
        print(alg, "steps run", len(hist), "masked left", int((x == 1).sum()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "generation.npz"), **res)
    print("generation written")


def generation_stochastic_fixture():
    """Stochastic generation settings of the reference loop (src/utils/generation_utils.py:45-82, :139-237): alg "origin"
    with temperature / top-p / top-k candidate sampling, and confidence ranking with sampled candidates + Gumbel `alg_temp`.
    The reference runs under recorders of its random draws - the categorical samples (`dists.Categorical.sample`), torch.rand
    (transfer mask) and torch.rand_like (Gumbel uniforms) - which are stored per iteration next to the token grid after every
    iteration, so a restatement fed with the same draws must reproduce the grids exactly."""
    import types
    PT, FT, Cfg = import_reference()
    gen = importlib.import_module("src.utils.generation_utils")
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=300, stacked_feat=4, next_n_token=4)
    state = weights_mod.make_state_dict(spec, seed=321, std=0.08, head_std=0.2)
    model = PT(ref_config(Cfg, spec))
    load_weights(model, state)
    model.eval()
    batch = synth.make_pretrain_batch(B=3, S=16, F=4, V=300, seed=77)
    ids = torch.from_numpy(batch["input_ids"])
    att = torch.from_numpy(batch["attention_mask"])
    res = {"in_input_ids": batch["input_ids"], "in_attention_mask": batch["attention_mask"],
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([0.08, 0.2, 321], np.float64)}
    runs = {"origin": dict(alg="origin", temperature=0.8, top_p=0.9, top_k=20, alg_temp=None),
            "gumbel": dict(alg="maskgit_plus", temperature=0.5, top_p=None, top_k=30, alg_temp=0.4),
            "margin_t": dict(alg="topk_margin", temperature=1.0, top_p=0.95, top_k=None, alg_temp=None)}
    real_rand, real_rand_like, RealCat = torch.rand, torch.rand_like, gen.dists.Categorical
    for tag, kw in runs.items():
        g = torch.Generator().manual_seed(500 + len(tag))
        rec = {"x0": [], "rand": [], "rand_like": []}

        class Cat(RealCat):
            def sample(self, *a, **k):
                out = super().sample(*a, **k)
                rec["x0"].append(out.clone())
                return out

        def rand(*a, **k):
            k.pop("device", None)
            t = real_rand(*a, generator=g, **k)
            rec["rand"].append(t.clone())
            return t

        def rand_like(t, **k):
            out = real_rand(t.shape, generator=g, dtype=t.dtype)
            rec["rand_like"].append(out.clone())
            return out

        torch.manual_seed(900 + len(tag))
        gen.dists.Categorical, torch.rand, torch.rand_like = Cat, rand, rand_like
        try:
            cfg = types.SimpleNamespace(eps=1e-3, steps=6, mask_token_id=1, output_history=True, **kw)
            x, hist = gen.sample_per_batch(model, cfg, input_ids=ids.clone(), attention_mask=att, inputs_raw_embeds=None)
        finally:
            gen.dists.Categorical, torch.rand, torch.rand_like = RealCat, real_rand, real_rand_like
        res[f"{tag}_hist"] = np.stack([h.numpy() for h in hist])
        res[f"{tag}_x0"] = np.stack([t.numpy() for t in rec["x0"]])
        if rec["rand"]:
            res[f"{tag}_u_transfer"] = np.stack([t.numpy() for t in rec["rand"]])
        if rec["rand_like"]:
            res[f"{tag}_u_gumbel"] = np.stack([t.numpy() for t in rec["rand_like"]])
        res[f"{tag}_cfg"] = np.array([kw["temperature"], kw["top_p"] or 0.0, kw["top_k"] or 0, kw["alg_temp"] or 0.0], np.float64)
        print(tag, "iterations", len(hist), "categorical draws", len(rec["x0"]), "masked left", int((x == 1).sum()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "generation_stochastic.npz"), **res)
    print("generation_stochastic written")


def hostmask_fixture():
    """Host-side SMTP masking: the reference _mask_stacked_input_ids_v2 under a recording random.sample."""
    import random as _random
    import_reference()
    tu = sys.modules["src.utils.tokenizer_utils"]
    res = {}
    for tag, (seq, dim, ratio, seed) in {"a": (17, 13, 0.37, 1), "b": (40, 4, 0.9, 2), "c": (9, 1, 0.05, 3)}.items():
        rng = np.random.RandomState(seed)
        ids = rng.randint(2, 700, size=(seq, dim)).astype(np.int64)
        if tag == "b":
            ids[5, 2] = 0        # a pad-valued cell: keeps 0 but is labelled when sampled
        rec = {}
        real = _random.sample

        def sample(pop, k):
            out = real(pop, k)
            rec["idx"] = list(out)
            return out

        _random.seed(100 + seed)
        _random.sample = sample
        try:
            new_ids, labels = tu._mask_stacked_input_ids_v2(ids.tolist(), 1, list(range(2, 700)), mask_ratio=ratio,
                                                            mask_token_precent=(1, 0, 0), pad_token_id=0)
        finally:
            _random.sample = real
        if tag == "b" and (5 * dim + 2) not in rec["idx"]:
            rec["idx"][0] = 5 * dim + 2   # make sure the pad-valued cell is exercised: rerun with the edited list
            _random.sample = lambda pop, k: list(rec["idx"])
            try:
                new_ids, labels = tu._mask_stacked_input_ids_v2(ids.tolist(), 1, list(range(2, 700)), mask_ratio=ratio,
                                                                mask_token_precent=(1, 0, 0), pad_token_id=0)
            finally:
                _random.sample = real
        res.update({f"{tag}_ids": ids, f"{tag}_idx": np.array(rec["idx"], np.int64), f"{tag}_ratio": np.float64(ratio),
                    f"{tag}_out_ids": np.array(new_ids, np.int64), f"{tag}_out_labels": np.array(labels, np.int64)})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hostmask.npz"), **res)
    print("hostmask written")


def ckpt_fixture():
    """Checkpoint interchange: a checkpoint WRITTEN BY THE REFERENCE - a reference-initialised model (HF
    `_init_weights`, seeded), saved the way `save_ddp_ckp` saves it under DDP (torch.save of a state dict whose keys carry the
    `module.` prefix, src/utils/misc_utils.py:105-121) into tests/golden/ref_ckpt/epoch_3/model.pt - next to the reference's
    own loss / logits / gradient norms on a seeded batch with those weights.  Also the reverse direction: a model.pt written
    by this package's `save_model` is loaded by the reference with strict=True and must reproduce the same loss."""
    PT, FT, Cfg = import_reference()
    spec = spec_mod.ModelSpec(kind=spec_mod.KIND_PRETRAIN, vocab_size=300, hidden_size=128, intermediate_size=256, num_layers=1,
                              num_heads=2, head_dim=64, stacked_feat=4, next_n_token=4)
    torch.manual_seed(4242)
    model = PT(ref_config(Cfg, spec))          # reference init: normal(0, initializer_range), pad row zero, norms one
    model.eval()
    d = os.path.join(ROOT, "tests", "golden", "ref_ckpt", "epoch_3")
    os.makedirs(d, exist_ok=True)
    sd = {"module." + k: v.detach().clone() for k, v in model.state_dict().items()}    # = DDP(model).state_dict()
    torch.save(sd, os.path.join(d, "model.pt"))
    batch = synth.make_pretrain_batch(B=6, S=24, F=4, V=300, seed=55)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=None)
    model.zero_grad()
    o.head1_loss.backward()
    names = [k for k in model.state_dict().keys() if "rotary_emb" not in k]
    res = {"loss": np.float64(o.head1_loss.item()), "logits": o.head1_logits.detach().float().numpy()[:64],
           "grad_norms": grad_norms(model, names), "names": np.array(names), "meta_spec": np.array(spec.as_c_ints(), np.int64)}
    for k, v in batch.items():
        res["in_" + k] = v
    # reverse direction: our writer -> reference reader
    ours = import_module("graph-gpt_amd.modeling")
    ck = import_module("graph-gpt_amd.checkpoint")
    import tempfile
    mine = ours.GraphGPTPretrainBase(ours.GraphGPTConfig(hidden_act="gelu",
        vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
        max_position_embeddings=spec.max_position, causal_attention=False, stacked_feat=4, next_n_token=4), seed=9)
    with tempfile.TemporaryDirectory() as td:
        ck.save_model(mine, td, ddp_prefix=True)
        back = torch.load(os.path.join(td, "model.pt"), map_location="cpu")
        back = {(k[7:] if k.startswith("module.") else k): v for k, v in back.items()}      # loader_utils.py:192-194
        m2 = PT(ref_config(Cfg, spec))
        missing, unexpected = m2.load_state_dict(back, strict=False)
        assert not unexpected and all("rotary_emb" in m for m in missing), (missing, unexpected)
        m2.eval()
        with torch.no_grad():
            o2 = m2(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=None)
        res["roundtrip_loss_of_seed9_model"] = np.float64(o2.head1_loss.item())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_ckpt.npz"), **res)
    print(f"ref_ckpt written: loss {res['loss']:.6f}, {len(names)} tensors, model.pt {os.path.getsize(os.path.join(d, 'model.pt'))} bytes; "
          f"our save_model -> reference strict load ok, loss {res['roundtrip_loss_of_seed9_model']:.6f}")


def auc_fixture():
    """Fine-tune head with loss_type "auc" (src/utils/loss_utils.py:25-53 through modeling_finetune.py:203-207), num_neg = 2:
    the reference's loss / gradients on a seeded batch, with the negative-sample indices its torch.randperm call drew
    (re-derived from the same generator state) stored next to them - a restatement fed with those indices must match."""
    PT, FT, Cfg = import_reference()
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=1000, stacked_feat=4, next_n_token=1, num_labels=2)
    state = weights_mod.make_state_dict(spec, seed=611, std=0.06, head_std=0.15)
    batch = synth.make_task_batch(B=24, S=24, F=4, V=1000, num_labels=2, seed=61)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    num_neg = 2
    model = FT(ref_config(Cfg, spec, num_labels=2, loss_type="auc", num_neg=num_neg, mlp=[], problem_type="single_label_classification"))
    load_weights(model, state)
    model.eval()
    y = tb["task_labels"].view(-1)
    P, N = int((y != 0).sum()), int((y == 0).sum())
    torch.manual_seed(8642)
    idx = torch.randperm(P * num_neg, dtype=torch.int64) % N          # what auc_loss will draw next
    torch.manual_seed(8642)
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
              task_labels=tb["task_labels"])
    model.zero_grad()
    o.task_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.task_loss.item()), "logits": o.task_logits.detach().float().numpy(), "idx": idx.numpy(),
           "num_neg": np.int64(num_neg), "grad_norms": grad_norms(model, names), "names": np.array(names),
           "grad_score": g["score.weight"].grad.numpy().copy(),
           "grad_l1_down": g["model.layers.1.mlp.down_proj.weight"].grad.numpy().copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([611, 0.06, 0.15])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ft_tiny_auc.npz"), **res)
    print(f"ft_tiny_auc written: loss {res['loss']:.6f}, P {P} N {N}, pairs {P * num_neg}")


def dropout_fixture():
    """Training-mode dropouts of the reference outside attention: embed_pdrop (modeling_helpers.py:96-98) and mlp_pdrop (the
    reference's own LlamaMLP, utils_graphgpt.py:69-80).  The reference runs in train() mode with torch.nn.functional.dropout
    replaced by a recorder that draws the keep mask itself and stores it (call order: embedding, then per layer the gated
    activations and the down projection's output); a restatement fed with the stored masks must match loss and gradients."""
    import torch.nn.functional as F
    PT, FT, Cfg = import_reference()
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    state = weights_mod.make_state_dict(spec, seed=733, std=0.06, head_std=0.15)
    batch = synth.make_pretrain_batch(B=4, S=24, F=13, V=756, seed=73)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = PT(ref_config(Cfg, spec, embed_pdrop=0.1, mlp_pdrop=0.2))
    load_weights(model, state)
    model.train()
    rec = []
    gen = torch.Generator().manual_seed(97531)
    orig = F.dropout

    def recording_dropout(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = (torch.rand(input.shape, generator=gen) >= p)
        rec.append((tuple(input.shape), float(p), keep.numpy().copy()))
        return input * keep.to(input.dtype) / (1.0 - p)

    F.dropout = recording_dropout
    try:
        o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=None)
        model.zero_grad()
        o.head1_loss.backward()
    finally:
        F.dropout = orig
    L_ = spec.num_layers
    assert len(rec) == 1 + 2 * L_, [r[:2] for r in rec]
    assert rec[0][0] == (4, 24, 13, spec.hidden_size) and rec[0][1] == 0.1
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.head1_loss.item()), "grad_norms": grad_norms(model, names), "names": np.array(names),
           "grad_embed": g["model.embed_tokens.weight"].grad.numpy().copy(),
           "grad_l0_down": g["model.layers.0.mlp.down_proj.weight"].grad.numpy().copy(),
           "grad_l1_gate": g["model.layers.1.mlp.gate_proj.weight"].grad.numpy().copy(),
           "embed_keep": np.packbits(rec[0][2]), "embed_shape": np.array(rec[0][0]),
           "p": np.array([0.1, 0.2]), "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([733, 0.06, 0.15])}
    for i in range(L_):
        assert rec[1 + 2 * i][0] == (4, 24, spec.intermediate_size) and rec[2 + 2 * i][0] == (4, 24, spec.hidden_size)
        res[f"act_keep_{i}"] = np.packbits(rec[1 + 2 * i][2])
        res[f"out_keep_{i}"] = np.packbits(rec[2 + 2 * i][2])
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pt_tiny_dropouts.npz"), **res)
    print(f"pt_tiny_dropouts written: loss {res['loss']:.6f}, {len(rec)} dropout calls recorded")


def mlp_head_fixture():
    """Fine-tune model with the `MLP` score head (config.mlp = [48, 32], src/utils/modules_utils.py:8-34 chosen at
    modeling_finetune.py:88-97), regression => every Linear has a bias.  Evaluation mode: loss / logits / gradients; training
    mode with config.dropout = 0.25: the same with the keep masks of the head's three dropout calls recorded (see dropout_fixture)."""
    import torch.nn.functional as F
    PT, FT, Cfg = import_reference()
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=1,
                                   score_bias=True, head_mlp=(48, 32))
    state = weights_mod.make_state_dict(spec, seed=811, std=0.06, head_std=0.3)
    rs = np.random.RandomState(5)
    for k in state:
        if k.startswith("score.mlp_modules.") and k.endswith(".bias"):
            state[k] = rs.uniform(-0.2, 0.2, size=state[k].shape).astype(np.float32)
    batch = synth.make_task_batch(B=16, S=24, F=13, V=756, num_labels=1, regression=True, seed=81)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    cfg = ref_config(Cfg, spec, num_labels=1, loss_type=None, mlp=[48, 32], dropout=0.25, problem_type="regression")
    model = FT(cfg)
    load_weights(model, state)
    names = list(state.keys())
    res = {"meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([811, 0.06, 0.3]), "names": np.array(names),
           "head_mlp": np.array([48, 32]), "p": np.float64(0.25)}
    for k in names:
        if k.startswith("score."):
            res["w_" + k] = state[k]
    fwd = lambda: model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
                        task_labels=tb["task_labels"])
    model.eval()
    o = fwd()
    model.zero_grad()
    o.task_loss.backward()
    g = dict(model.named_parameters())
    res.update(loss=np.float64(o.task_loss.item()), logits=o.task_logits.detach().float().numpy(), grad_norms=grad_norms(model, names),
               grad_w0=g["score.mlp_modules.0.weight"].grad.numpy().copy(), grad_b1=g["score.mlp_modules.1.bias"].grad.numpy().copy(),
               grad_l1_down=g["model.layers.1.mlp.down_proj.weight"].grad.numpy().copy())
    model.train()
    rec = []
    gen = torch.Generator().manual_seed(24680)
    orig = F.dropout

    def recording_dropout(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = (torch.rand(input.shape, generator=gen) >= p)
        rec.append((tuple(input.shape), float(p), keep.numpy().copy()))
        return input * keep.to(input.dtype) / (1.0 - p)

    F.dropout = recording_dropout
    try:
        o = fwd()
        model.zero_grad()
        o.task_loss.backward()
    finally:
        F.dropout = orig
    assert [r[0][-1] for r in rec] == [spec.hidden_size, 48, 32] and all(r[1] == 0.25 for r in rec), [r[:2] for r in rec]
    B, S = tb["input_ids"].shape[:2]
    seq_len = (tb["input_ids"][:, :, 0] != 0).sum(-1) - 1
    for i, (shape, _, keep) in enumerate(rec):           # the reference runs the head on every row: keep the pooled rows' masks
        res[f"train_keep_{i}"] = keep.reshape(B, S, -1)[np.arange(B), seq_len.numpy()].copy()
    res.update(train_loss=np.float64(o.task_loss.item()), train_grad_norms=grad_norms(model, names),
               train_grad_w0=g["score.mlp_modules.0.weight"].grad.numpy().copy())
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ft_tiny_mlphead.npz"), **res)
    print(f"ft_tiny_mlphead written: eval loss {res['loss']:.6f}, train loss {res['train_loss']:.6f}")


def focal_fixture():
    """config.focal_gamma = 2 (FocalLoss, utils_graphgpt.py:340-376, through _get_ce_loss :158-160): the reference's pre-train
    loss and gradients on a seeded batch."""
    PT, FT, Cfg = import_reference()
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    state = weights_mod.make_state_dict(spec, seed=911, std=0.06, head_std=0.15)
    batch = synth.make_pretrain_batch(B=4, S=24, F=13, V=756, seed=91)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = PT(ref_config(Cfg, spec, focal_gamma=2.0))
    load_weights(model, state)
    model.eval()
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=None)
    model.zero_grad()
    o.head1_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.head1_loss.item()), "gamma": np.float64(2.0), "grad_norms": grad_norms(model, names),
           "names": np.array(names), "grad_lm_head": g["lm_head.weight"].grad.numpy().copy(),
           "grad_l0_q": g["model.layers.0.self_attn.q_proj.weight"].grad.numpy().copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([911, 0.06, 0.15])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pt_tiny_focal.npz"), **res)
    print(f"pt_tiny_focal written: loss {res['loss']:.6f}")


def long_fixture():
    """config.stack_method = "long" (examples/node_lvl/proteins_supervised.sh:31): the 1 / (non-zero ids) embedding ratio
    (modeling_helpers.py:106-110) in both models and the per-feature-level SMTP loss (:327-342, :368-374).  Real rows get empty
    (0-valued) feature cells - only this stacking has them; a labelled empty cell keeps id 0 and label 0 (tokenizer_utils.py:112-148)."""
    PT, FT, Cfg = import_reference()
    rs = np.random.RandomState(77)

    def with_empty_cells(ids, labels=None):
        ids = ids.copy()
        empty = (rs.uniform(size=ids.shape) < 0.3) & (ids[:, :, :1] != 0)
        empty[:, :, 0] = False
        ids[empty] = 0
        if labels is not None:
            labels = labels.copy()
            labels[empty & (labels != -100)] = 0
        return ids, labels

    # pre-train
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13)
    state = weights_mod.make_state_dict(spec, seed=1011, std=0.06, head_std=0.15)
    batch = synth.make_pretrain_batch(B=5, S=24, F=13, V=756, seed=101)
    batch["input_ids"], batch["labels"] = with_empty_cells(batch["input_ids"], batch["labels"])
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = PT(ref_config(Cfg, spec, stack_method="long"))
    load_weights(model, state)
    model.eval()
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=None)
    model.zero_grad()
    o.head1_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.head1_loss.item()), "grad_norms": grad_norms(model, names), "names": np.array(names),
           "grad_lm_head": g["lm_head.weight"].grad.numpy().copy(),
           "grad_l0_q": g["model.layers.0.self_attn.q_proj.weight"].grad.numpy().copy(),
           "grad_embed_rows": g["model.embed_tokens.weight"].grad.numpy()[:64].copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([1011, 0.06, 0.15])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pt_tiny_long.npz"), **res)
    print(f"pt_tiny_long written: loss {res['loss']:.6f}")

    # fine-tune
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2)
    state = weights_mod.make_state_dict(spec, seed=1012, std=0.06, head_std=0.3)
    batch = synth.make_task_batch(B=12, S=24, F=13, V=756, seed=102)
    batch["input_ids"], _ = with_empty_cells(batch["input_ids"])
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = FT(ref_config(Cfg, spec, stack_method="long", num_labels=2, loss_type=None))
    load_weights(model, state)
    model.eval()
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
              task_labels=tb["task_labels"])
    model.zero_grad()
    o.task_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.task_loss.item()), "logits": o.task_logits.detach().float().numpy(),
           "grad_norms": grad_norms(model, names), "names": np.array(names),
           "grad_embed_rows": g["model.embed_tokens.weight"].grad.numpy()[:64].copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([1012, 0.06, 0.3])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ft_tiny_long.npz"), **res)
    print(f"ft_tiny_long written: loss {res['loss']:.6f}")


def token_ce_fixture():
    """Token-level task (config.loss_type = "token_ce", the nodev2 labelling of src/utils/tokenizer_utils.py:688-745): `score` on every
    row, labels [B,S] with -100 on unlabelled rows (modeling_finetune.py:162-164, :198-202).  7 classes, roughly a third of the real
    rows labelled."""
    PT, FT, Cfg = import_reference()
    C = 7
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=C)
    state = weights_mod.make_state_dict(spec, seed=1311, std=0.06, head_std=0.3)
    batch = synth.make_task_batch(B=10, S=24, F=13, V=756, seed=131)
    rs = np.random.RandomState(13)
    real = batch["input_ids"][:, :, 0] != 0
    lab = rs.randint(0, C, size=real.shape).astype(np.int64)
    lab[~real | (rs.uniform(size=real.shape) > 0.35)] = -100
    batch["task_labels"] = lab
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = FT(ref_config(Cfg, spec, num_labels=C, loss_type="token_ce", problem_type="single_label_classification"))
    load_weights(model, state)
    model.eval()
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
              task_labels=tb["task_labels"])
    model.zero_grad()
    o.task_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.task_loss.item()), "logits": o.task_logits.detach().float().numpy(),
           "grad_norms": grad_norms(model, names), "names": np.array(names), "grad_score": g["score.weight"].grad.numpy().copy(),
           "grad_l1_down": g["model.layers.1.mlp.down_proj.weight"].grad.numpy().copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([1311, 0.06, 0.3])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ft_tiny_tokence.npz"), **res)
    print(f"ft_tiny_tokence written: loss {res['loss']:.6f}, logits {res['logits'].shape}, labelled rows {(lab >= 0).sum()}")


def rope_range_fixture():
    """config.rope_range = 6 (utils_graphgpt.reset_pos_ids :574-581): the fine-tune model with arbitrary position ids per row,
    rescaled to [0, 6) before the rotary embedding - fractional positions."""
    PT, FT, Cfg = import_reference()
    spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2)
    state = weights_mod.make_state_dict(spec, seed=1411, std=0.06, head_std=0.3)
    batch = synth.make_task_batch(B=12, S=24, F=13, V=756, seed=141)
    rs = np.random.RandomState(14)
    pos = batch["position_ids"].copy()
    pos[::2] = pos[::2] * 3 + rs.randint(0, 5, size=(pos[::2].shape[0], 1))       # uneven ranges per row
    batch["position_ids"] = pos
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    model = FT(ref_config(Cfg, spec, num_labels=2, loss_type=None, rope_range=6))
    load_weights(model, state)
    model.eval()
    o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
              task_labels=tb["task_labels"])
    model.zero_grad()
    o.task_loss.backward()
    names = list(state.keys())
    g = dict(model.named_parameters())
    res = {"loss": np.float64(o.task_loss.item()), "logits": o.task_logits.detach().float().numpy(), "rope_range": np.float64(6),
           "grad_norms": grad_norms(model, names), "names": np.array(names),
           "grad_l0_q": g["model.layers.0.self_attn.q_proj.weight"].grad.numpy().copy(),
           "grad_l0_k": g["model.layers.0.self_attn.k_proj.weight"].grad.numpy().copy(),
           "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([1411, 0.06, 0.3])}
    for k, v in batch.items():
        res["in_" + k] = v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ft_tiny_roperange.npz"), **res)
    print(f"ft_tiny_roperange written: loss {res['loss']:.6f}")


def raw_embeds_fixture():
    """config.embed_dim = 64: raw-embedding inputs [B,S,64] (modeling_pretrain.py:69-84, :131-149; modeling_helpers.py:127-139).  Pre-train: the
    batch is masked token-wise for half of the samples (every label of a masked token set - those rows take emb_mask_token) and cell-wise
    for the rest (rows with some label unset keep their raw embedding).  Fine-tune: no mask token.  Evaluation mode."""
    PT, FT, Cfg = import_reference()
    E = 64
    rs = np.random.RandomState(151)
    for kind in ("pt", "ft"):
        if kind == "pt":
            spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_PRETRAIN, vocab_size=756, stacked_feat=13, next_n_token=13, embed_dim=E)
            batch = synth.make_pretrain_batch(B=6, S=24, F=13, V=756, seed=151)
            ids, lab = batch["input_ids"], batch["labels"]
            real = batch["attention_mask"] != 0
            for b in range(0, ids.shape[0], 2):           # token-wise masking on the even samples
                tok = real[b] & (rs.uniform(size=real[b].shape) < 0.4)
                orig = np.where(lab[b] != -100, lab[b], ids[b])
                lab[b] = np.where(tok[:, None], orig, -100)
                ids[b] = np.where(tok[:, None], 1, orig)
        else:
            spec = spec_mod.spec_from_size("tiny", kind=spec_mod.KIND_TASK, vocab_size=756, stacked_feat=13, next_n_token=1, num_labels=2, embed_dim=E)
            batch = synth.make_task_batch(B=12, S=24, F=13, V=756, seed=152)
        state = weights_mod.make_state_dict(spec, seed=1511 if kind == "pt" else 1512, std=0.06, head_std=0.15 if kind == "pt" else 0.3)
        state["embed_layernorm.weight"] = rs.uniform(0.5, 1.5, size=state["embed_layernorm.weight"].shape).astype(np.float32)
        raw = rs.standard_normal(size=batch["input_ids"].shape[:2] + (E,)).astype(np.float32)
        batch["inputs_raw_embeds"] = raw
        tb = {k: torch.from_numpy(v) for k, v in batch.items()}
        model = (PT if kind == "pt" else FT)(ref_config(Cfg, spec, embed_dim=E, **({} if kind == "pt" else dict(num_labels=2, loss_type=None))))
        load_weights(model, state)
        model.eval()
        if kind == "pt":
            o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], labels=tb["labels"], inputs_raw_embeds=tb["inputs_raw_embeds"])
            loss = o.head1_loss
        else:
            o = model(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], position_ids=tb["position_ids"],
                      task_labels=tb["task_labels"], inputs_raw_embeds=tb["inputs_raw_embeds"])
            loss = o.task_loss
        model.zero_grad()
        loss.backward()
        names = list(state.keys())
        g = dict(model.named_parameters())
        res = {"loss": np.float64(loss.item()), "grad_norms": grad_norms(model, names), "names": np.array(names), "embed_dim": np.int64(E),
               "grad_embed_proj": g["embed_proj.weight"].grad.numpy().copy(), "grad_embed_ln": g["embed_layernorm.weight"].grad.numpy().copy(),
               "w_embed_ln": state["embed_layernorm.weight"],
               "meta_spec": np.array(spec.as_c_ints(), np.int64), "meta_init": np.array([1511 if kind == "pt" else 1512, 0.06, 0.15 if kind == "pt" else 0.3])}
        if kind == "pt":
            res["grad_mask_token"] = g["emb_mask_token"].grad.numpy().reshape(-1).copy()
            full = (batch["labels"] != -100).all(-1)
            assert full.any() and ((batch["labels"] != -100).any(-1) & ~full).any()
        else:
            res["logits"] = o.task_logits.detach().float().numpy()
        for k, v in batch.items():
            res["in_" + k] = v
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{kind}_tiny_rawembed.npz"), **res)
        print(f"{kind}_tiny_rawembed written: loss {res['loss']:.6f}")


def config_convert_fixture():
    """The reference's OWN `convert_to_legacy_config` (configuration_graphgpt.py:210-342) run on its OWN nested
    `GraphGPTModelConfig` (src/conf/model/model_configs.py:247-287) with the settings of examples/graph_lvl/pcqm4m_v2_pretrain.sh
    (base), examples/edge_lvl/ppa_supervised.sh (base) and a stress case that sets every hot-path field to a non-default.
    The fixture holds, per case, the nested input (dataclasses.asdict) and every key of the reference's flat result."""
    import dataclasses
    import json
    from src.conf.model.model_configs import GraphGPTModelConfig
    from src.models.graphgpt.configuration_graphgpt import convert_to_legacy_config as ref_convert
    from src.utils.modules_utils import set_up_model_architect

    def base(hidden, layers):
        c = GraphGPTModelConfig(hidden_size=hidden, num_hidden_layers=layers, intermediate_size=0, num_attention_heads=0,
                                max_position_embeddings=1024)
        c.intermediate_size, c.num_attention_heads, c.head_dim = set_up_model_architect(hidden_size=hidden)   # modules_utils.py:63-70
        return c

    cases = {}
    c = base(768, 12)                           # pcqm4m_v2_pretrain.sh + set_model_config (modules_utils.py:57-81)
    c.vocab_size, c.causal_attention = 756, False
    c.graph_input.stack_method, c.graph_input.stacked_feat_agg_method, c.graph_input.stacked_feat = "short", "sum", 13
    c.pt_head.next_n_token = 13
    c.dropout_settings.attention_dropout = 0.1
    c.bos_token_id, c.eos_token_id = 19, 20
    cases["pcqm4m_v2_pretrain_base"] = c
    c = base(768, 12)                           # ppa_supervised.sh + set_ft_model_config (modules_utils.py:84-92)
    c.vocab_size, c.causal_attention = 41245, False
    c.graph_input.stack_method, c.graph_input.stacked_feat_agg_method, c.graph_input.stacked_feat = "short", "sum", 4
    c.dropout_settings.attention_dropout, c.dropout_settings.path_dropout = 0.1, 0.2
    c.layer_scale_init_value = 1.0
    c.ft_head.num_labels, c.ft_head.problem_type, c.ft_head.loss_type, c.ft_head.task_ratio = 2, "single_label_classification", "", 1
    c.num_key_value_heads, c.tie_word_embeddings, c.pt_head.next_n_token = c.num_attention_heads, False, 1
    cases["ogbl_ppa_supervised_base"] = c
    c = base(256, 4)                            # every hot-path field away from its default
    c.vocab_size, c.causal_attention, c.rope_range, c.layer_scale_init_value = 1000, True, 6, 0.5
    c.dropout_settings.embed_dropout, c.dropout_settings.path_dropout = 0.05, 0.2
    c.dropout_settings.mlp_dropout, c.dropout_settings.attention_dropout = 0.15, 0.1
    c.graph_input.stack_method, c.graph_input.stacked_feat_agg_method = "long", "gated"
    c.graph_input.stacked_feat, c.graph_input.embed_dim = 7, 64
    c.pt_head.next_n_token, c.pt_head.focal_gamma, c.pt_head.smtp_inside = 7, 2.0, True
    c.ft_head.mlp, c.ft_head.dropout, c.ft_head.pooling_method = [256, 64], 0.25, "last"
    c.ft_head.loss_type, c.ft_head.num_neg, c.ft_head.num_labels, c.ft_head.problem_type = "auc", 3, 1, "regression"
    c.ft_head.task_ratio = 0.5
    c.pad_token_id, c.cls_token_id, c.rope_theta, c.rms_norm_eps, c.initializer_range = 0, 5, 50000.0, 1e-5, 0.01
    c.pos_pt_head.smtp_power = 1.25
    cases["stress_all_fields"] = c
    out = {}
    for name, c in cases.items():
        flat = ref_convert(c).to_dict()
        nested = dataclasses.asdict(c)
        json.dumps(flat, default=str)
        out[name] = {"nested": nested, "flat": flat}
    with open(os.path.join(ROOT, "tests", "golden", "config_convert.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True, default=str)
    print("config_convert.json written:", {k: len(v["flat"]) for k, v in out.items()})


def pipeline_config_fixture():
    """What the reference's OWN config plumbing makes of a `Config`-shaped tree on the way into `TrainingPipeline.run()`
    (pipeline.py:60-95): base_configs.update_num_steps / update_epochs / update_ft_num_steps / set_finetune_cfg, modules_utils
    .set_model_config / set_ft_model_config with a stand-in tokenizer (three id getters), convert_to_legacy_config, and
    conf_utils.parse_deepspeed_config(_for_ft) on the reference's own examples/ds_config2_pt.json / ds_config2.json - the last one
    records the scheduler parameters DeepSpeed would be handed (warmup_min_lr == warmup_max_lr: SURVEY row A12's constant-lr quirk)
    and the torch OneCycleLR the fine-tune / DDP paths build, sampled.  Data only: inputs + the reference's outputs."""
    import copy
    import dataclasses
    import json
    import types
    from src.conf import base_configs as BC
    from src.conf.model.model_configs import GraphGPTModelConfig
    from src.models.graphgpt.configuration_graphgpt import convert_to_legacy_config as ref_convert
    from src.utils import modules_utils, conf_utils, loss_utils

    class Tok:                                   # what set_model_config reads from the tokenizer (modules_utils.py:78-80)
        def __init__(self, v, b, e): self.vocab_size, self._b, self._e = v, b, e
        def get_bos_token_id(self): return self._b
        def get_eos_token_id(self): return self._e

    out = {}
    # ---- pre-training: pcqm4m_v2_pretrain.sh (base), 8 ranks
    os.environ["WORLD_SIZE"] = "8"
    mc = GraphGPTModelConfig(hidden_size=768, num_hidden_layers=12, intermediate_size=0, num_attention_heads=0, max_position_embeddings=1024)
    mc.graph_input.stack_method, mc.graph_input.stacked_feat_agg_method, mc.graph_input.stacked_feat = "short", "sum", 13
    mc.dropout_settings.attention_dropout = 0.1
    tc = BC.TrainingConfig(task_type="pretrain-mlm", batch_size=256, deepspeed_conf_file=REF + "/examples/ds_config2_pt.json",
                           output_dir="/tmp/gget_fixture_out", pretrain_mlm=None)
    tc.schedule.total_tokens, tc.schedule.warmup_tokens, tc.schedule.samples_per_saving = 4e9, 1e8, 1000000
    tc.optimizer.lr, tc.optimizer.eps, tc.optimizer.weight_decay, tc.optimizer.max_grad_norm = 3e-4, 1e-8, 0.1, 1.0
    nested_in, train_in = dataclasses.asdict(mc), dataclasses.asdict(tc)
    cfg = types.SimpleNamespace(model=mc, training=tc)
    tps, samples_per_gpu, world = 22.37, 3378606, 8
    BC.update_num_steps(tc.schedule, tps, tc.batch_size, world)
    BC.update_epochs(tc.schedule, tps, samples_per_gpu, world)
    mc2 = modules_utils.set_model_config(cfg, Tok(756, 19, 20))
    flat = ref_convert(mc2).to_dict()
    tc.optimizer.min_lr = tc.optimizer.lr * 0.1                       # pretrain_mode.py:108 (use_deepspeed)
    ds = conf_utils.parse_deepspeed_config(training=tc, loss_utils=loss_utils)
    out["pretrain_ds"] = {"model_nested": nested_in, "training": train_in, "tokens_per_sample": tps, "samples_per_gpu": samples_per_gpu,
                          "world_size": world, "tokenizer": {"vocab_size": 756, "bos_token_id": 19, "eos_token_id": 20},
                          "total_num_steps": tc.schedule.total_num_steps, "warmup_num_steps": tc.schedule.warmup_num_steps,
                          "epochs": tc.schedule.epochs, "flat": flat,
                          "ds_optimizer": ds["optimizer"], "ds_scheduler": ds["scheduler"], "ds_gradient_clipping": ds["gradient_clipping"],
                          "ds_train_batch_size": ds["train_batch_size"]}
    # ---- the same run on the DDP path (no DeepSpeed JSON): AdamW + OneCycleLR (opt_utils.py:18-33)
    tc2 = copy.deepcopy(tc)
    tc2.deepspeed_conf_file, tc2.optimizer.min_lr = "", 0
    tc2.schedule.total_num_steps, tc2.schedule.warmup_num_steps = 2000, 150
    gen, _ = loss_utils.set_py_scheduler("OneCycleLR", {"scheduler": {"params": {}}}, max_lr=tc2.optimizer.lr, min_lr=tc2.optimizer.min_lr,
                                         total_steps=tc2.schedule.total_num_steps + 1,
                                         pct_start=tc2.schedule.warmup_num_steps / tc2.schedule.total_num_steps, last_step_index=-1)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=tc2.optimizer.lr)
    sch = gen(opt)
    lrs = []
    for _ in range(tc2.schedule.total_num_steps):
        lrs.append(opt.param_groups[0]["lr"]); opt.step(); sch.step()
    out["pretrain_ddp"] = {"training": dataclasses.asdict(tc2), "lr_by_step": lrs[::50] + lrs[-3:], "lr_stride": 50}
    # ---- fine-tuning: ppa_supervised.sh (base) under DeepSpeed: torch OneCycleLR built by parse_deepspeed_config_for_ft
    os.environ["WORLD_SIZE"] = "4"
    mc = GraphGPTModelConfig(hidden_size=768, num_hidden_layers=12, intermediate_size=0, num_attention_heads=0, max_position_embeddings=1024)
    mc.graph_input.stack_method, mc.graph_input.stacked_feat_agg_method, mc.graph_input.stacked_feat = "short", "sum", 4
    mc.dropout_settings.attention_dropout, mc.dropout_settings.path_dropout, mc.layer_scale_init_value = 0.1, 0.2, 1.0
    mc.ft_head.num_labels, mc.ft_head.problem_type, mc.ft_head.loss_type = 2, "single_label_classification", ""
    tc = BC.TrainingConfig(task_type="edge", batch_size=64, deepspeed_conf_file=REF + "/examples/ds_config2.json",
                           output_dir="/tmp/gget_fixture_out", pretrain_cpt="", pretrain_mlm=None)
    tc.schedule.epochs, tc.schedule.warmup_epochs = 8, 0.6
    tc.optimizer.lr, tc.optimizer.min_lr, tc.optimizer.eps, tc.optimizer.weight_decay = 1e-4, 0.0, 1e-10, 0.02
    tc.finetune.task_ratio = 1.0
    nested_in, train_in = dataclasses.asdict(mc), dataclasses.asdict(tc)
    cfg = types.SimpleNamespace(model=mc, training=tc)
    samples_per_gpu = 10000
    BC.update_ft_num_steps(tc, samples_per_gpu)
    mc2 = modules_utils.set_ft_model_config(cfg, Tok(41245, 1, 2))
    flat = ref_convert(mc2).to_dict()
    BC.set_finetune_cfg(tc.finetune)
    ds, gen, sconf = conf_utils.parse_deepspeed_config_for_ft(tc, loss_utils)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=tc.optimizer.lr)
    sch = gen(opt)
    lrs = []
    for _ in range(tc.schedule.total_num_steps - 1):
        lrs.append(opt.param_groups[0]["lr"]); opt.step(); sch.step()
    out["finetune_ds"] = {"model_nested": nested_in, "training": train_in, "samples_per_gpu": samples_per_gpu, "world_size": 4,
                          "tokenizer": {"vocab_size": 41245, "bos_token_id": 1, "eos_token_id": 2},
                          "total_num_steps": tc.schedule.total_num_steps, "warmup_num_steps": tc.schedule.warmup_num_steps,
                          "finetune": dataclasses.asdict(tc.finetune), "flat": flat, "ds_optimizer": ds["optimizer"],
                          "scheduler_conf": sconf, "lr_by_step": lrs[::25] + lrs[-3:], "lr_stride": 25}
    with open(os.path.join(ROOT, "tests", "golden", "pipeline_config.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True, default=str)
    print("pipeline_config.json written:", {k: sorted(v)[:4] for k, v in out.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    classes = import_reference()
    only = set(sys.argv[1:])
    for name, kind, skw, bkw, ikw in CASES:
        if only and name not in only:
            continue
        run_case(name, kind, dict(skw), dict(bkw), dict(ikw), classes)
    if not only:
        lr_fixture()
    if not only or "smtp2d" in only:
        smtp2d_fixture()
    if not only or "generation" in only:
        generation_fixture()
    if not only or "generation_stochastic" in only:
        generation_stochastic_fixture()
    if not only or "hostmask" in only:
        hostmask_fixture()
    if not only or "ref_ckpt" in only:
        ckpt_fixture()
    if not only or "ft_tiny_auc" in only:
        auc_fixture()
    if not only or "pt_tiny_dropouts" in only:
        dropout_fixture()
    if not only or "ft_tiny_mlphead" in only:
        mlp_head_fixture()
    if not only or "pt_tiny_focal" in only:
        focal_fixture()
    if not only or "tiny_long" in only:
        long_fixture()
    if not only or "ft_tiny_tokence" in only:
        token_ce_fixture()
    if not only or "ft_tiny_roperange" in only:
        rope_range_fixture()
    if not only or "tiny_rawembed" in only:
        raw_embeds_fixture()
    if not only or "config_convert" in only:
        config_convert_fixture()
    if not only or "pipeline_config" in only:
        pipeline_config_fixture()


if __name__ == "__main__":
    main()
