#!/usr/bin/env python3
"""Timing experiment (results are NOT a valid training run): the clip + AdamW launch of step n on a side stream while forward and
backward of step n+1 run on the compute stream - what a bucketed optimizer overlapped with the next forward could return at most.
Prints ms/step of the sequential step and of the overlapped arrangement, alternated."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec_mod = importlib.import_module("graph-gpt_amd.spec")
synth = importlib.import_module("graph-gpt_amd.synth")
weights = importlib.import_module("graph-gpt_amd.weights")
eng_mod = importlib.import_module("graph-gpt_amd.engine")

B, S, F, V = 256, 32, 13, 756
spec = spec_mod.spec_from_size("base", kind=spec_mod.KIND_PRETRAIN, vocab_size=V, stacked_feat=F, next_n_token=F, causal=False, max_position=1024)
e = eng_mod.Engine(spec, B * S, B)
e.load_state_dict(weights.make_state_dict(spec, seed=0))
batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234)
n_tok = int(batch["attention_mask"].sum())
dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
side = torch.cuda.Stream()


def fwd_bwd():
    e.set_dropout(0.1, 0.0, 7)
    e.forward_pretrain(dev["input_ids"], dev["attention_mask"], dev["labels"], num_tokens=n_tok)
    e.backward()


def seq(n):
    for _ in range(n):
        fwd_bwd()
        e.adamw_step(3e-4)


def ovl(n):
    main = torch.cuda.current_stream()
    for _ in range(n):
        fwd_bwd()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            e.adamw_step(3e-4)
    main.wait_stream(side)


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


seq(3); ovl(3)
for r in range(3):
    print(f"round {r}: sequential {timed(seq):.3f} ms/step   adamw on a side stream under the next fwd+bwd {timed(ovl):.3f} ms/step", flush=True)
