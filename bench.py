#!/usr/bin/env python3
"""Headline benchmark: SMTP pre-training steps of the Graph Eulerian Transformer on synthetic
Eulerian-token batches (BASELINE.json: graph-tokens/sec + SMTP loss, PCQM4M-v2 base model).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward (+ bucketed RCCL all-reduce overlapped on a side stream when N>1) + fused
clip/AdamW over one batch of B=256 x S=32 x F=13 tokens per GPU, inputs resident in HBM.  Prints ONE JSON
line on rank 0.  `value` counts un-padded graph tokens (reference metric definition,
src/conf/stats_configs.py:69-76) summed over all ranks.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (model size, B per GPU, S, F, V)   -- SURVEY.md 8d C1/C2, toy for quick checks
    "pcqm4m-v2-pretrain-base": ("base", 256, 32, 13, 756),
    "pcqm4m-v2-pretrain-base24": ("base24", 256, 32, 13, 756),
    "toy-tiny": ("tiny", 128, 64, 1, 300),
}


def flops_per_step(spec, B, S, M, Lm):
    """Algorithmic FLOPs of one training step: F_step = 3*F_fwd, no recompute credit,
    full SxS attention, head terms with the measured M / Lm of the batch."""
    T, d, ff, L, F, V = B * S, spec.hidden_size, spec.intermediate_size, spec.num_layers, spec.next_n_token, spec.vocab_size
    fwd = T * L * (8 * d * d + 6 * d * ff) + 4 * L * B * S * S * d + M * 2 * d * (F * d if F > 1 else 0) + Lm * 2 * d * V
    return 3.0 * fwd


def cpu_baseline(spec, state, F, V, seed, threads):
    """Oracle (CPU port of the reference path, parity-pinned) timed on the host cores: fwd + bwd + AdamW, fp32,
    on a bounded sample of the same workload (smaller batch, same S/F/V/model)."""
    from oracle import gget_oracle as O
    synth = importlib.import_module("graph-gpt_amd.synth")
    torch.set_num_threads(threads)
    Bc, Sc = 16, 32
    b = synth.make_pretrain_batch(B=Bc, S=Sc, F=F, V=V, seed=seed)
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    p = O.to_params(state, torch.float32)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    fn = lambda q: O.pretrain_forward(spec, q, tb["input_ids"], tb["attention_mask"], tb["labels"])
    times = []
    t_begin = time.time()
    for it in range(6):
        t0 = time.time()
        _, grads = O.loss_and_grads(fn, p, "head1_loss")
        with torch.no_grad():
            O.adamw_step(p, grads, m, v, it + 1, 3e-4, 0.9, 0.95, 1e-8, 0.1, 1.0)
        times.append(time.time() - t0)
        if time.time() - t_begin > 25.0:   # bounded: the default bench run must finish within minutes
            break
    dt = float(np.median(times[1:])) if len(times) > 1 else float(times[0])
    real = int(b["attention_mask"].sum())
    return {"value": real / dt, "unit": "graph-tokens/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 fwd+bwd+AdamW, base model, B={Bc} S={Sc}, {len(times)} steps ({dt:.2f} s/step)"}


def time_dominant_kernel(spec, T, iters=20):
    """Live HIP-event timing of the dominant kernel: the FFN gate|up GEMM [T,d]x[d,2ff] (NT bf16 MFMA)."""
    L = importlib.import_module("graph-gpt_amd._lib")
    import ctypes as C
    lib = L.load()
    d, ff = spec.hidden_size, spec.intermediate_size
    A = torch.randn(T, d, device="cuda").to(torch.bfloat16)
    W = (torch.randn(2 * ff, d, device="cuda") * 0.02).to(torch.bfloat16)
    Cm = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (L.GEMM_NT, L.EPI_NONE, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(Cm.data_ptr()), None,
            T, 2 * ff, d, d, d, 2 * ff, 1, st)
    for _ in range(3):
        L.check(lib.gget_op_gemm(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.check(lib.gget_op_gemm(*args))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * T * d * 2 * ff
    return fl / (ms * 1e-3) / 1e12, ms


def pmc_traffic():
    """L2<->fabric bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r01_gu_gemm_pmc.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc runs); None when
    the summary is absent.  PMC counters cannot be collected from inside the process being timed."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gu_gemm_pmc.json")
    try:
        with open(p) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="pcqm4m-v2-pretrain-base", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    # GGET_BENCH_BACKEND=gloo: plumbing check of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices;
    # RCCL refuses that).  The driver's runs use the default: one rank per GPU over RCCL.
    backend = os.environ.get("GGET_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, init_method="env://")

    spec_mod = importlib.import_module("graph-gpt_amd.spec")
    weights = importlib.import_module("graph-gpt_amd.weights")
    synth = importlib.import_module("graph-gpt_amd.synth")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    training = importlib.import_module("graph-gpt_amd.training")

    size, B, S, F, V = WORKLOADS[a.workload]
    sz = spec_mod.MODEL_SIZES[size]
    cfg = modeling.GraphGPTConfig(vocab_size=V, hidden_size=sz["hidden_size"], intermediate_size=4 * sz["hidden_size"],
                                  num_hidden_layers=sz["num_layers"], num_attention_heads=sz["hidden_size"] // 64,
                                  max_position_embeddings=1024, causal_attention=False, stacked_feat=F, next_n_token=F,
                                  attention_dropout=0.1)   # every reference pre-train script trains with 0.1
    model = modeling.GraphGPTPretrainBase(cfg, seed=0)   # same random-init weights on every rank (DP replicas)
    spec = model.spec
    model._ensure_engine(B, S)
    engine = training.initialize(model, training.OptimConfig(lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1,
                                                             max_grad_norm=1.0))
    batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234 + rank)     # distinct data per rank
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    real_tokens = synth.real_tokens(batch)

    def step():
        return training.batch_training(dev, engine)

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, float(real_tokens), float(loss.item())], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, tot_real, mean_loss = float(mx[0]), float(sm[1]), float(sm[2]) / world
    else:
        tot_real, mean_loss = float(real_tokens), float(loss.item())

    if rank == 0:
        M, Lm = model._engine.head_counts()
        fstep = flops_per_step(spec, B, S, M, Lm)
        ms = dt / a.steps * 1e3
        step_tflops = fstep / (ms * 1e-3) / 1e12
        k_tflops, k_ms = time_dominant_kernel(spec, B * S)
        out = {
            "metric": "graph-tokens/sec (un-padded Eulerian tokens, whole job) + SMTP loss, PCQM4M-v2 base pre-train",
            "value": tot_real * a.steps / dt, "unit": "graph-tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": a.workload, "model": f"{size} d{spec.hidden_size}/L{spec.num_layers}/H{spec.num_heads} "
                       f"({spec.num_params() / 1e6:.1f}M params)", "per_gpu_batch": B, "global_batch": B * world,
                       "seq_len": S, "stacked_feat": F, "vocab": V, "parallelism": f"dp{world}",
                       "step": "fwd+bwd+allreduce+clip+AdamW", "attention_dropout": cfg.attention_dropout},
            "smtp_loss": mean_loss,
            "padded_tokens_per_s": B * S * world * a.steps / dt,
            "tokens_per_s_per_gpu": tot_real * a.steps / dt / world,
            "step_mfma": {"flops_per_step": fstep, "achieved_tflops_per_gpu": step_tflops,
                          "frac_of_peak": step_tflops / PEAK_BF16_TFLOPS},
            "roofline": {"bound": "mfma", "kernel": "gemm_persist_kernel<256,256,64,NT> FFN gate|up [T,d]x[d,2ff] (26% of step FLOPs)",
                         "achieved": k_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": k_tflops / PEAK_BF16_TFLOPS, "traffic": pmc_traffic() if B * S == 8192 and spec.hidden_size == 768 else None,
                         "avg_launch_ms": k_ms},
        }
        if world == 1 and not a.no_cpu_baseline:
            state = weights.make_state_dict(spec, seed=0)
            out["cpu_baseline"] = cpu_baseline(spec, state, F, V, 1234, min(os.cpu_count() or 1, 32))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 is still timing the dominant kernel: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
