#!/usr/bin/env python3
"""Where does a step spend HOST time when the batches come through DevicePrefetcher?  (round 6, debugging aid)"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("graph-gpt_amd.modeling"); tr = importlib.import_module("graph-gpt_amd.training"); synth = importlib.import_module("graph-gpt_amd.synth")
cfg = M.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                       max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13, attention_dropout=0.1)
model = M.GraphGPTPretrainBase(cfg, seed=0); model._ensure_engine(256, 32)
eng = tr.initialize(model, tr.OptimConfig(lr=3e-4))
host = [{k: torch.from_numpy(v) for k, v in synth.make_pretrain_batch(B=256, S=32, F=13, V=756, seed=1234 + i).items() if k != "lengths"} for i in range(4)]
def feed(n):
    for i in range(n): yield host[i % 4]
for mode in ("prefetch", "sync", "resident"):
    N = 30
    torch.cuda.synchronize()
    if mode == "prefetch":
        pf = tr.DevicePrefetcher(feed(N + 3), model.device)
        orig, orig2 = pf._stage, pf._to_device
        tstage = []
        def timed(data, slot, orig=orig):
            t = time.perf_counter(); r = orig(data, slot); tstage.append(time.perf_counter() - t); return r
        def timed2(st, orig2=orig2):
            t = time.perf_counter(); r = orig2(st); tdev.append(time.perf_counter() - t); return r
        tdev = []
        pf._stage, pf._to_device = timed, timed2
        it = iter(pf); tstep = []
        for _ in range(3): tr.batch_training(next(it), eng)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for d in it:
            t = time.perf_counter(); tr.batch_training(d, eng); tstep.append(time.perf_counter() - t)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(mode, f"{dt / N * 1e3:.3f} ms/step; host: stage avg {sum(tstage[4:]) / len(tstage[4:]) * 1e3:.3f} ms (max {max(tstage[4:]) * 1e3:.3f}), step enqueue avg {sum(tstep) / len(tstep) * 1e3:.3f} ms, device-copy enqueue avg {sum(tdev[4:]) / len(tdev[4:]) * 1e3:.3f} ms (max {max(tdev[4:]) * 1e3:.3f})")
    else:
        dev = [{k: v.cuda() for k, v in b.items()} for b in host]
        for i in range(3): tr.batch_training(dev[i % 4], eng)
        torch.cuda.synchronize(); t0 = time.perf_counter(); tstep = []
        for i in range(N):
            t = time.perf_counter()
            d = {k: v.cuda() for k, v in host[i % 4].items()} if mode == "sync" else dev[i % 4]
            tr.batch_training(d, eng); tstep.append(time.perf_counter() - t)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(mode, f"{dt / N * 1e3:.3f} ms/step; host step enqueue avg {sum(tstep) / len(tstep) * 1e3:.3f} ms")
