#!/usr/bin/env python3
"""Step-time survey of the BASELINE.json configs other than the headline one (they are parity/coverage cases, not bench
lines): C2 base24 pre-train, C3 ogbl-ppa-like fine-tune (S=256, V=41245, LayerScale + DropPath), C4 long-sequence fine-tune."""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec_mod = importlib.import_module("graph-gpt_amd.spec")
synth = importlib.import_module("graph-gpt_amd.synth")
modeling = importlib.import_module("graph-gpt_amd.modeling")
training = importlib.import_module("graph-gpt_amd.training")

def run(name, kind, size, B, S, F, V, steps=6, warmup=2, **cfgkw):
    sz = spec_mod.MODEL_SIZES[size]
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=sz["hidden_size"], intermediate_size=4 * sz["hidden_size"],
                                  num_hidden_layers=sz["num_layers"], num_attention_heads=sz["hidden_size"] // 64,
                                  max_position_embeddings=max(1024, S), causal_attention=False, stacked_feat=F,
                                  next_n_token=F if kind == "pt" else 1, attention_dropout=0.1, **cfgkw)
    model = (modeling.GraphGPTPretrainBase if kind == "pt" else modeling.GraphGPTTaskModel)(cfg, seed=0)
    model._ensure_engine(B, S)
    eng = training.initialize(model, training.OptimConfig(lr=3e-4))
    if kind == "pt":
        batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234)
        fn = lambda d: training.batch_training(d, eng)
    else:
        batch = synth.make_task_batch(B=B, S=S, F=F, V=V, seed=1234, lengths="uniform", min_len=S // 4)
        fn = lambda d: training.ft_batch_training(d, eng)[0]
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    for _ in range(warmup): loss = fn(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): loss = fn(dev)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    real = int(batch["attention_mask"].sum())
    print(json.dumps({"config": name, "ms_per_step": dt * 1e3, "real_tokens_per_s": real / dt, "padded_tokens_per_s": B * S / dt,
                      "loss": float(loss), "workspace_GB": model._engine.workspace_bytes / 2**30}), flush=True)
    del eng, model
    torch.cuda.empty_cache()

which = sys.argv[1:] or ["c2", "c3", "c4"]
if "c2" in which: run("C2 base24 pre-train B256 S32 F13 V756", "pt", "base24", 256, 32, 13, 756)
if "c3" in which: run("C3 ppa-like fine-tune base B256 S256 F4 V41245 (LayerScale 1, DropPath 0.2)", "ft", "base", 256, 256, 4, 41245,
                      layer_scale_init_value=1.0, path_pdrop=0.2, num_labels=2, problem_type="single_label_classification")
if "c4" in which: run("C4 long-seq fine-tune base B16 S2048 F4 V41245", "ft", "base", 16, 2048, 4, 41245, steps=3, warmup=1,
                      num_labels=2, problem_type="single_label_classification")
