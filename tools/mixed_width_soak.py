#!/usr/bin/env python3
"""Real epochs change the padded width from batch to batch (the collator pads to each batch's longest graph).  N steps of the base model over
batches whose width cycles through 24 .. 64 (every launch plan, the per-sample kernels with and without 33 .. 64-row samples, the three-launch
forward) fed through DevicePrefetcher: the loss must stay finite and fall, no deferred guard may fire, the step time per width is printed.
    python tools/mixed_width_soak.py [steps=600]
"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("graph-gpt_amd.modeling"); tr = importlib.import_module("graph-gpt_amd.training"); synth = importlib.import_module("graph-gpt_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
widths = (32, 40, 24, 56, 32, 48, 64, 40)
cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                       max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13, attention_dropout=0.1)
model = M.GraphGPTPretrainBase(cfg, seed=0); model._ensure_engine(256, 64)
eng = tr.initialize(model, tr.OptimConfig(lr=3e-4, betas=(0.9, 0.95), weight_decay=0.1, max_grad_norm=1.0))
host = [{k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=256, S=S_, F=13, V=756, seed=500 + i).items() if k != "lengths"} for i, S_ in enumerate(widths * 2)]
def feed():
    for i in range(steps): yield host[i % len(host)]
losses, t_by = [], {}
torch.cuda.synchronize(); t0 = time.perf_counter(); last = t0
for i, data in enumerate(tr.DevicePrefetcher(feed(), model.device)):
    losses.append(tr.batch_training(data, eng))
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize(); now = time.perf_counter()
        l = [float(x) for x in losses[-100:]]
        print(f"step {i + 1}: loss mean of last 100 = {np.mean(l):.4f} (min {min(l):.4f} max {max(l):.4f}), {(now - last) / 100 * 1e3:.3f} ms/step over the mixed widths", flush=True)
        assert all(np.isfinite(l)), "non-finite loss"
        last = now
        model.check_deferred()
torch.cuda.synchronize()
l = [float(x) for x in losses]
print(f"{steps} steps over widths {widths}: first-100 mean loss {np.mean(l[:100]):.4f} -> last-100 {np.mean(l[-100:]):.4f}; total {(time.perf_counter() - t0):.1f} s; "
      f"skipped steps {eng.skipped_steps}; engine rows/varlen of the last step {model._engine.varlen_status()}")
assert np.mean(l[-100:]) < np.mean(l[:100]), "the loss did not fall"
