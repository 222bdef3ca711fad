"""`TrainingPipeline(cfg, mode)` driven by the reference's `Config` shape (reference src/training/pipeline.py
:60-95, :97-139, pretrain_mode.py:96-230, finetune_mode.py:181-192, conf_utils.py:49-131, opt_utils.py:7-36).

tests/golden/pipeline_config.json holds what the REFERENCE's own plumbing (base_configs.update_num_steps / update_epochs /
update_ft_num_steps / set_finetune_cfg, modules_utils.set_model_config / set_ft_model_config, convert_to_legacy_config,
conf_utils.parse_deepspeed_config(_for_ft), loss_utils.set_py_scheduler) made of three configurations
(tools/make_golden.py:pipeline_config_fixture).  The CPU tests run the pipeline's config phases (everything before the model is
created) on the same inputs; the GPU test runs whole pipelines."""
import copy
import importlib
import json
import math
import os
import types

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_HF_BOOKKEEPING = {"_name_or_path", "architectures", "chunk_size_feed_forward", "dtype", "id2label", "label2id", "is_encoder_decoder",
                   "output_attentions", "output_hidden_states", "return_dict", "rope_parameters", "transformers_version", "model_type"}


def _ns(obj):
    if isinstance(obj, dict):
        return types.SimpleNamespace(**{k: _ns(v) for k, v in obj.items()})
    return obj


def _cases():
    with open(os.path.join(HERE, "golden", "pipeline_config.json")) as fh:
        return json.load(fh)


def _cfg(case, shape):
    """the reference-shaped tree from the fixture's INPUTS: attribute tree, plain dicts or this repo's dataclasses"""
    model, training = copy.deepcopy(case["model_nested"]), copy.deepcopy(case["training"])
    if shape == "namespace":
        return types.SimpleNamespace(tokenization=None, model=_ns(model), training=_ns(training), generation=None)
    if shape == "dict":
        return {"tokenization": None, "model": model, "training": training, "generation": None}
    CF = importlib.import_module("graph-gpt_amd.conf")
    def dc(cls, d):
        names = {f.name for f in __import__("dataclasses").fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names and not isinstance(v, dict)})
    tc = dc(CF.TrainingConfig, training)
    tc.schedule, tc.optimizer = dc(CF.ScheduleConfig, training["schedule"]), dc(CF.OptimizerConfig, training["optimizer"])
    tc.finetune, tc.distributed = dc(CF.FinetuneTrainConfig, training["finetune"]), CF.DistConfig()
    return CF.Config(model=_ns(model), training=tc)


def _config_phases(cfg, mode, world, monkeypatch):
    """pipeline.run() up to (not including) model creation - no GPU needed"""
    T = importlib.import_module("graph-gpt_amd.training")
    monkeypatch.setattr(T, "set_dist_env", lambda backend=None: (0, 0, world))
    p = T.TrainingPipeline(cfg, mode)
    assert p.reference_cfg
    p._extract_config()
    p.mode.update_config(p)
    p._setup_deepspeed_flag()
    p._setup_distributed()
    p.mode.prepare_data(p)
    return p


@pytest.mark.parametrize("shape", ["namespace", "dict", "dataclass"])
def test_pretrain_config_phases_match_reference(shape, monkeypatch):
    T = importlib.import_module("graph-gpt_amd.training")
    CF = importlib.import_module("graph-gpt_amd.conf")
    case = _cases()["pretrain_ds"]
    tok = case["tokenizer"]
    mode = T.PretrainMode(batches=[], tokens_per_sample=case["tokens_per_sample"], samples_per_gpu=case["samples_per_gpu"], **tok)
    p = _config_phases(_cfg(case, shape), mode, case["world_size"], monkeypatch)
    g = CF._get
    assert p.use_deepspeed and g(p.train_cfg, "use_deepspeed") is True
    assert (g(p.sched_cfg, "total_num_steps"), g(p.sched_cfg, "warmup_num_steps"), g(p.sched_cfg, "epochs")) == \
        (case["total_num_steps"], case["warmup_num_steps"], case["epochs"])
    assert g(p.sched_cfg, "steps_per_saving") == 1000000 // (8 * 256)                                    # pretrain_mode.py:114-116
    assert g(p.optim_cfg, "min_lr") == pytest.approx(0.1 * 3e-4)                                          # pretrain_mode.py:108
    assert (g(g(p.train_cfg, "distributed"), "world_size"), g(g(p.train_cfg, "distributed"), "rank")) == (8, 0)
    got = p.config.to_dict()
    for k, v in case["flat"].items():
        if k not in _HF_BOOKKEEPING:
            assert got[k] == v, f"{k}: {got[k]!r} != reference {v!r}"
    assert (p.config.vocab_size, p.config.next_n_token, p.config.causal_attention, p.config.bos_token_id) == (756, 13, False, 19)
    # optimizer: what the reference hands DeepSpeed (conf_utils.py:49-103) - Adam block and the WarmupDecayLR parameters; the
    # schedule has warmup_min_lr == warmup_max_lr, i.e. lr(step) is the constant lr whatever gamma(step) is
    o = CF.optim_from_training(p.train_cfg, p.use_deepspeed, finetune=False)
    ds_o, ds_s = case["ds_optimizer"]["params"], case["ds_scheduler"]
    assert ds_s["type"] == "WarmupDecayLR" and ds_s["params"]["warmup_min_lr"] == ds_s["params"]["warmup_max_lr"] == ds_o["lr"]
    assert (o.lr, list(o.betas), o.eps, o.weight_decay, o.max_grad_norm) == \
        (ds_o["lr"], ds_o["betas"], ds_o["eps"], ds_o["weight_decay"], case["ds_gradient_clipping"])
    assert (o.schedule, o.min_lr, o.warmup_num_steps, o.total_num_steps) == \
        ("warmup_decay", ds_s["params"]["warmup_min_lr"], ds_s["params"]["warmup_num_steps"], ds_s["params"]["total_num_steps"])
    for step in (0, 1, 100, 2183, 40000, 87310):
        assert o.lr_at(step) == pytest.approx(3e-4, rel=1e-12)


def test_pretrain_ddp_schedule_matches_reference_onecycle(monkeypatch):
    """No DeepSpeed JSON -> the reference's DDP path: AdamW + OneCycleLR over total_num_steps + 1 steps (opt_utils.py:18-33)."""
    CF = importlib.import_module("graph-gpt_amd.conf")
    case = _cases()["pretrain_ddp"]
    tc = _ns(copy.deepcopy(case["training"]))
    o = CF.optim_from_training(tc, use_deepspeed=False, finetune=False)
    assert (o.schedule, o.onecycle_extra_step, o.total_num_steps, o.warmup_num_steps) == ("onecycle", 1, 2000, 150)
    lrs, stride = case["lr_by_step"], case["lr_stride"]
    steps = list(range(0, 2000, stride)) + [1997, 1998, 1999]
    assert len(steps) == len(lrs)
    for s, want in zip(steps, lrs):
        assert o.lr_at(s) == pytest.approx(want, rel=1e-9, abs=1e-18), s


@pytest.mark.parametrize("shape", ["namespace", "dict"])
def test_finetune_config_phases_match_reference(shape, monkeypatch):
    T = importlib.import_module("graph-gpt_amd.training")
    CF = importlib.import_module("graph-gpt_amd.conf")
    case = _cases()["finetune_ds"]
    mode = T.FinetuneMode(batches=[], samples_per_gpu=case["samples_per_gpu"], **case["tokenizer"])
    p = _config_phases(_cfg(case, shape), mode, case["world_size"], monkeypatch)
    g = CF._get
    assert (g(p.sched_cfg, "total_num_steps"), g(p.sched_cfg, "warmup_num_steps")) == (case["total_num_steps"], case["warmup_num_steps"])
    got = p.config.to_dict()
    for k, v in case["flat"].items():
        if k not in _HF_BOOKKEEPING:
            assert got[k] == v, f"{k}: {got[k]!r} != reference {v!r}"
    assert (p.config.next_n_token, p.config.num_labels, p.config.path_pdrop, p.config.layer_scale_init_value) == (1, 2, 0.2, 1.0)
    CF.set_finetune_cfg(g(p.train_cfg, "finetune"))
    assert {k: g(g(p.train_cfg, "finetune"), k) for k in ("aux_ratio", "use_aux")} == {k: case["finetune"][k] for k in ("aux_ratio", "use_aux")}
    # DeepSpeed + torch OneCycleLR (conf_utils.py:106-131): total_steps = total_num_steps (no extra step)
    o = CF.optim_from_training(p.train_cfg, p.use_deepspeed, finetune=True)
    sp = case["scheduler_conf"]["scheduler"]["params"]
    assert (o.schedule, o.onecycle_extra_step, o.total_num_steps) == ("onecycle", 0, sp["total_steps"])
    assert o.warmup_num_steps / o.total_num_steps == pytest.approx(sp["pct_start"])
    ds_o = case["ds_optimizer"]["params"]
    assert (o.lr, list(o.betas), o.eps, o.weight_decay) == (ds_o["lr"], ds_o["betas"], ds_o["eps"], ds_o["weight_decay"])
    lrs, stride = case["lr_by_step"], case["lr_stride"]
    n = case["total_num_steps"] - 1
    steps = list(range(0, n, stride)) + [n - 3, n - 2, n - 1]
    assert len(steps) == len(lrs)
    for s, want in zip(steps, lrs):
        assert o.lr_at(s) == pytest.approx(want, rel=1e-9, abs=1e-18), s


def test_pretrain_mode_smtp_inside_and_euler_rules(monkeypatch):
    """pretrain_mode.py:121-128: `pt_head.smtp_inside` follows the tokenizer's masking method whatever the model config said (here: the
    mode's `mask_inside_model`, the tokenizer lives on the host), and with it the task type becomes pretrain-smtp; :191-195: the
    pretrain-euler task counts HALF the tokens per sample in its schedule."""
    T = importlib.import_module("graph-gpt_amd.training")
    CF = importlib.import_module("graph-gpt_amd.conf")
    case = _cases()["pretrain_ds"]
    tok = case["tokenizer"]
    g = CF._get

    def phases(mask_inside, task=None, smtp_cfg=True):
        cfg = _cfg(case, "namespace")
        cfg.model.pt_head.smtp_inside = smtp_cfg
        if task:
            cfg.training.task_type = task
        mode = T.PretrainMode(batches=[], tokens_per_sample=case["tokens_per_sample"], samples_per_gpu=case["samples_per_gpu"],
                              mask_inside_model=mask_inside, **tok)
        return _config_phases(cfg, mode, case["world_size"], monkeypatch)
    p = phases(False)
    assert g(g(p.model_cfg, "pt_head"), "smtp_inside") is False and p.config.smtp_inside is False      # reset although the config said True
    assert g(p.sched_cfg, "total_num_steps") == case["total_num_steps"]
    p = phases(True, smtp_cfg=False)
    assert g(g(p.model_cfg, "pt_head"), "smtp_inside") is True and g(p.train_cfg, "task_type") == "pretrain-smtp" and p.config.smtp_inside is True
    p = phases(False, task="pretrain-euler")
    half = int(case["tokens_per_sample"]) // 2
    want = int(math.ceil(case["training"]["schedule"]["total_tokens"] / (half * case["training"]["batch_size"] * case["world_size"])))
    assert g(p.sched_cfg, "total_num_steps") == want and want > case["total_num_steps"]


def test_lean_config_still_recognised():
    T = importlib.import_module("graph-gpt_amd.training")
    p = T.TrainingPipeline({"model": dict(vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                                          hidden_act="gelu", causal_attention=False), "optim": {"lr": 1e-3}, "batches": [], "max_steps": 3},
                           T.PretrainMode())
    assert not p.reference_cfg and p.config.hidden_size == 128 and p.optim.lr == 1e-3 and p.max_steps == 3
    from src.conf import Config
    assert not T.TrainingPipeline(Config(model=p.config, optim=None, batches=[]), T.PretrainMode()).reference_cfg


def test_missing_schedule_inputs_fail_loudly(monkeypatch):
    T = importlib.import_module("graph-gpt_amd.training")
    case = _cases()["pretrain_ds"]
    with pytest.raises(ValueError, match="tokens_per_sample"):
        _config_phases(_cfg(case, "namespace"), T.PretrainMode(batches=[]), 8, monkeypatch)
    with pytest.raises(ValueError, match="samples_per_gpu"):
        _config_phases(_cfg(_cases()["finetune_ds"], "namespace"), T.FinetuneMode(batches=[]), 4, monkeypatch)


# ---------------------------------------------------------------------------------------------- whole pipelines on the GPU
def _tiny_reference_cfg(tmp_path, kind):
    case = _cases()["pretrain_ds" if kind == "pt" else "finetune_ds"]
    cfg = _cfg(case, "namespace")
    m = cfg.model
    m.hidden_size, m.num_hidden_layers, m.intermediate_size, m.num_attention_heads, m.head_dim = 128, 2, 512, 2, 64
    m.num_key_value_heads, m.max_position_embeddings = 2, 64
    m.dropout_settings.attention_dropout = m.dropout_settings.path_dropout = 0.0
    m.layer_scale_init_value = 0.0
    t = cfg.training
    t.output_dir, t.batch_size = str(tmp_path / "out"), 8
    return cfg, case


@pytest.mark.gpu
def test_pretrain_pipeline_runs_from_a_reference_shaped_config(tmp_path):
    """3 optimizer steps through TrainingPipeline(cfg, PretrainMode(...)).run() with the reference's Config shape: the schedule comes
    from the token budget, the optimizer from training.optimizer (+ the DeepSpeed JSON's scheduler type), the model from the nested
    model config; the losses equal the same 3 steps driven by hand through initialize / batch_training with the same settings."""
    import numpy as np
    import torch
    T = importlib.import_module("graph-gpt_amd.training")
    M = importlib.import_module("graph-gpt_amd.modeling")
    synth = importlib.import_module("graph-gpt_amd.synth")
    cfg, case = _tiny_reference_cfg(tmp_path, "pt")
    cfg.training.deepspeed_conf_file = "ds_config2_pt.json"      # (absent here: the stage's default scheduler type, WarmupDecayLR)
    cfg.training.schedule.total_tokens, cfg.training.schedule.warmup_tokens = 3 * 8 * 20.0, 8 * 20.0
    batches = [{k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=50 + i).items()} for i in range(5)]
    mode = T.PretrainMode(batches=batches, tokens_per_sample=20.0, samples_per_gpu=1000, vocab_size=756, bos_token_id=19, eos_token_id=20)
    p = T.TrainingPipeline(cfg, mode).run()
    assert p.reference_cfg and p.use_deepspeed and p.max_steps == 3 and p.engine.global_steps == 3
    assert (p.optim.schedule, p.optim.lr, p.optim.min_lr, p.optim.eps) == ("warmup_decay", 3e-4, 3e-4, 1e-8)
    assert os.path.isfile(os.path.join(cfg.training.output_dir, "config.json")) and os.path.isfile(os.path.join(cfg.training.output_dir, "model.pt"))
    saved = M.GraphGPTConfig.from_pretrained(cfg.training.output_dir)
    assert (saved.vocab_size, saved.stacked_feat, saved.next_n_token, saved.hidden_size) == (756, 13, 13, 128)
    last = float(p.last_loss)
    # by hand, same seed -> same initial weights
    model = M.GraphGPTPretrainBase(p.config)
    model.cuda()
    eng = T.initialize(model, T.OptimConfig(lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0))
    for b in batches[:3]:
        want = float(T.batch_training(b, eng))
    assert np.isfinite(last) and abs(last - want) <= 1e-6 * abs(want), (last, want)
    # resume: the run wrote log.csv next to its checkpoint (the reference's save_all, misc_utils.py:150-154); its presence in the output
    # directory makes the next run continue from that checkpoint (pipeline.py:127-133, :178-202) and do the REMAINDER of its schedule
    assert os.path.isfile(os.path.join(cfg.training.output_dir, "log.csv"))
    cfg2, _ = _tiny_reference_cfg(tmp_path, "pt")
    cfg2.training.deepspeed_conf_file = "ds_config2_pt.json"
    cfg2.training.schedule.total_tokens, cfg2.training.schedule.warmup_tokens = 4 * 8 * 20.0, 8 * 20.0       # a 4-step budget, 3 done
    p2 = T.TrainingPipeline(cfg2, T.PretrainMode(batches=batches[3:], tokens_per_sample=20.0, vocab_size=756, bos_token_id=19, eos_token_id=20))
    p2.run()
    assert p2.pretrain_cpt == cfg2.training.output_dir and p2.max_steps == 4 and p2.engine.global_steps == 4      # one step, not four more
    want4 = float(T.batch_training(batches[3], eng))
    assert abs(float(p2.last_loss) - want4) <= 2e-5 * abs(want4), (float(p2.last_loss), want4)
    # ... and a third run with the same budget has nothing left to do
    cfg3, _ = _tiny_reference_cfg(tmp_path, "pt")
    cfg3.training.deepspeed_conf_file = "ds_config2_pt.json"
    cfg3.training.schedule.total_tokens, cfg3.training.schedule.warmup_tokens = 4 * 8 * 20.0, 8 * 20.0
    p3 = T.TrainingPipeline(cfg3, T.PretrainMode(batches=batches, tokens_per_sample=20.0, vocab_size=756, bos_token_id=19, eos_token_id=20)).run()
    assert p3.engine.global_steps == 4 and p3.last_loss is None


@pytest.mark.gpu
def test_pipeline_ddp_branch_skips_a_non_finite_step(tmp_path):
    """a reference-shaped config with an EMPTY deepspeed_conf_file selects the reference's DDP / AMP branch
    (training_utils.py:46-86), whose optimizer step is GradScaler.step - a step with an inf / NaN gradient is skipped.  The modes drive
    the lean `batch_training(batch, engine)` form; the rule must still apply: a poisoned gradient inside a pipeline run leaves the weights
    finite, counts one skipped step and lets the LR schedule advance."""
    import torch
    T = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    cfg, _ = _tiny_reference_cfg(tmp_path, "pt")
    cfg.training.deepspeed_conf_file = ""
    cfg.training.schedule.total_tokens, cfg.training.schedule.warmup_tokens = 4 * 8 * 20.0, 8 * 20.0
    batches = [{k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=90 + i).items()} for i in range(5)]

    class Poisoning(T.PretrainMode):
        calls = 0

        def train_step(self, engine, batch):
            Poisoning.calls += 1
            if Poisoning.calls != 2:
                return super().train_step(engine, batch)
            out = engine(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
            engine.backward(out.head1_loss)
            engine.module._engine.grad_bf16[4321] = float("nan")
            engine.step()
            return out.head1_loss
    p = T.TrainingPipeline(cfg, Poisoning(batches=batches, tokens_per_sample=20.0, vocab_size=756, bos_token_id=19, eos_token_id=20)).run()
    e = p.model._engine
    torch.cuda.synchronize()
    assert not p.use_deepspeed and p.engine.skip_nonfinite
    assert p.engine.global_steps == 4 and p.engine.skipped_steps == 1 and e.step_count == 3       # the schedule moved on, Adam's count did not
    assert bool(torch.isfinite(e.master).all()) and bool(torch.isfinite(e.adam_m).all())
    import numpy as np
    assert np.isfinite(float(p.last_loss))


@pytest.mark.gpu
def test_gradient_accumulation_steps_on_the_deepspeed_branch(tmp_path):
    """`training.optimizer.gradient_accumulation_steps = k` on the DeepSpeed branch (conf_utils.py:59-66 hands it to the DS engine, whose
    step() only fires at the boundary): k micro-batches per optimizer step, gradients summed, the update made with their mean.  With the
    SAME micro-batch twice and k = 2 the mean gradient is that batch's gradient exactly (the fp32 sum of two equal bf16 values rounds
    back to twice the value), so the weights must equal one plain step on it, bit for bit; with two different micro-batches the update
    lies between the two single-batch updates' directions and one optimizer step is counted."""
    import torch
    T = importlib.import_module("graph-gpt_amd.training")
    M = importlib.import_module("graph-gpt_amd.modeling")
    CF = importlib.import_module("graph-gpt_amd.conf")
    synth = importlib.import_module("graph-gpt_amd.synth")
    cfg = dict(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=64, causal_attention=False, stacked_feat=13, next_n_token=13)
    A, Bb = ({k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=s_).items() if k != "lengths"} for s_ in (3, 4))

    def run(batches, k):
        model = M.GraphGPTPretrainBase(M.GraphGPTConfig(**cfg), seed=2).cuda().eval()
        eng = T.initialize(model, T.OptimConfig(lr=1e-3, max_grad_norm=1.0, gradient_accumulation_steps=k))
        for b in batches:
            T.batch_training(b, eng)
        torch.cuda.synchronize()
        return model._engine.master.clone(), eng, model._engine
    w1, e1, _ = run([A], 1)
    w2, e2, eng2 = run([A, A], 2)
    assert e2.global_steps == 1 and eng2.step_count == 1 and e2.micro_steps == 2
    assert torch.equal(w1, w2), "k = 2 on the same micro-batch twice must be the plain step"
    w3, e3, _ = run([A, Bb], 2)
    assert e3.global_steps == 1 and not torch.equal(w3, w1)
    w4, e4, _ = run([A, Bb, A], 2)            # the third call is a micro-step of the NEXT update: nothing applied yet
    assert e4.global_steps == 1 and torch.equal(w4, w3)
    # the reference-shaped config carries it on the DeepSpeed branch only (the DDP branch asserts k == 1, training_utils.py:47-49)
    case = _cases()["pretrain_ds"]
    tc = _ns(copy.deepcopy(case["training"]))
    tc.optimizer.gradient_accumulation_steps = 4
    tc.schedule.total_num_steps, tc.schedule.warmup_num_steps = 100, 10
    assert CF.optim_from_training(tc, use_deepspeed=True, finetune=False).gradient_accumulation_steps == 4
    assert CF.optim_from_training(tc, use_deepspeed=False, finetune=False).gradient_accumulation_steps == 1


@pytest.mark.gpu
def test_finetune_pipeline_runs_from_a_reference_shaped_config(tmp_path):
    import numpy as np
    import torch
    T = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    cfg, case = _tiny_reference_cfg(tmp_path, "ft")
    cfg.training.deepspeed_conf_file = ""                         # the DDP-style schedule: OneCycleLR over total + 1 steps
    cfg.training.schedule.epochs, cfg.training.schedule.warmup_epochs = 2, 0.5
    batches = [{k: torch.from_numpy(v) for k, v in synth.make_task_batch(B=8, S=32, F=4, V=41245, seed=70 + i).items()} for i in range(6)]
    mode = T.FinetuneMode(batches=batches, samples_per_gpu=16, vocab_size=41245, bos_token_id=1, eos_token_id=2)
    p = T.TrainingPipeline(cfg, mode).run()
    assert (p.max_steps, p.engine.global_steps) == (4, 4)         # 2 epochs x (16 // 8) steps
    assert (p.optim.schedule, p.optim.onecycle_extra_step, p.optim.warmup_num_steps) == ("onecycle", 1, 1)
    assert p.engine.last_lr == pytest.approx(p.optim.lr_at(3)) and np.isfinite(float(p.last_loss))
    assert p.config.next_n_token == 1 and p.config.num_labels == 2
