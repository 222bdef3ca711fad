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
    # name: (kind, model size, B per GPU, S, F, V)   -- SURVEY.md 8d C1..C4; toy for quick checks
    "pcqm4m-v2-pretrain-base": ("pt", "base", 256, 32, 13, 756),          # C1: the configuration the metric is quoted on
    "pcqm4m-v2-pretrain-base24": ("pt", "base24", 256, 32, 13, 756),      # C2 (per-GPU shape of the 8-GPU run)
    # the reference's pack_tokens option (src/data/tokenizer.py:359-415): the same 8192 token slots per step filled with
    # whole graphs back to back (block-diagonal attention) instead of one padded graph per row.  Reported NEXT TO the
    # headline line, never instead of it (every reference script runs pack_tokens=0).
    "pcqm4m-v2-pretrain-base-packed": ("pt-packed", "base", 64, 128, 13, 756),
    "ogbl-ppa-finetune-base": ("ft", "base", 256, 256, 4, 41245),         # C3: LayerScale + DropPath, 2-class edge task
    "longseq-finetune-base": ("ft-long", "base", 16, 2048, 4, 41245),     # C4 (per-GPU shape)
    "toy-tiny": ("pt", "tiny", 128, 64, 1, 300),
}


def flops_per_step(spec, B, S, M, Lm, kind="pt", rows=None, lengths=None):
    """Algorithmic FLOPs of one training step: F_step = 3*F_fwd, no recompute credit,
    full SxS attention, head terms with the measured M / Lm of the batch (fine-tune: the pooled score head).
    rows / lengths: the EXECUTED count of a var-len step - token-wise terms over the rows the engine ran on, attention over
    each sample's own len x len block."""
    T, d, ff, L, F, V = B * S, spec.hidden_size, spec.intermediate_size, spec.num_layers, spec.next_n_token, spec.vocab_size
    att = B * S * S if lengths is None else float((np.asarray(lengths, np.float64) ** 2).sum())
    fwd = (rows if rows is not None else T) * L * (8 * d * d + 6 * d * ff) + 4 * L * att * d
    if kind.startswith("pt"):
        fwd += M * 2 * d * (F * d if F > 1 else 0) + Lm * 2 * d * V
    else:
        fwd += B * 2 * d * max(int(getattr(spec, "num_labels", 2) or 2), 1)
    return 3.0 * fwd


def cpu_baseline(spec, state, B, S, F, V, seed, threads, budget_s=30.0, tail=None):
    """Oracle (CPU port of the reference path, parity-pinned) timed on the host cores: fwd + bwd + AdamW, fp32, on the
    SAME batch shape as the GPU run (B x S x F of the workload, same generator and seed), bounded in time: one warm-up step
    and as many timed steps as fit in ~budget_s seconds (at least one)."""
    from oracle import gget_oracle as O
    synth = importlib.import_module("graph-gpt_amd.synth")
    torch.set_num_threads(threads)
    b = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=seed, **(tail or {}))
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    # parity leg (the oracle as CHECKER, not timed): SMTP loss of the oracle's forward on the bench batch, weights rounded to bf16 first
    # (the cast point of the engine's compute copy) - compared with the engine's eval-mode forward in `loss_parity`
    with torch.no_grad():
        p_bf = O.to_params({k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}, torch.float32,
                           requires_grad=False)
        oracle_loss = float(O.pretrain_forward(spec, p_bf, tb["input_ids"], tb["attention_mask"], tb["labels"])["head1_loss"])
    p = O.to_params(state, torch.float32)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    fn = lambda q: O.pretrain_forward(spec, q, tb["input_ids"], tb["attention_mask"], tb["labels"])
    times = []
    t_begin = time.time()
    for it in range(8):
        t0 = time.time()
        _, grads = O.loss_and_grads(fn, p, "head1_loss")
        with torch.no_grad():
            O.adamw_step(p, grads, m, v, it + 1, 3e-4, 0.9, 0.95, 1e-8, 0.1, 1.0)
        times.append(time.time() - t0)
        if it >= 1 and time.time() - t_begin + times[-1] > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    dt = float(np.median(timed))
    real = int(b["attention_mask"].sum())
    return {"value": real / dt, "unit": "graph-tokens/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 fwd+bwd+AdamW, same batch shape as the GPU step (B={B} S={S} F={F}), 1 warm-up + {len(timed)} timed "
                      f"step(s), {dt:.2f} s/step"}, oracle_loss


def _event_time(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_kernels(spec, T, iters=20):
    """Live HIP-event timing (torch's current stream = the stream the kernels are launched on) of the dominant kernel - the
    FFN gate|up GEMM [T,d]x[d,2ff] (NT bf16 MFMA) with the gated-GELU product in its epilogue - and of the two next heaviest
    launches of a decoder layer: the dgrad into the residual stream through W_gu (NN, K = 2ff) and the grouped weight-gradient
    launch (TN, K = T, four problems)."""
    L = importlib.import_module("graph-gpt_amd._lib")
    import ctypes as C
    lib = L.load()
    d, ff = spec.hidden_size, spec.intermediate_size
    P = lambda t: C.c_void_p(t.data_ptr())
    bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x, wgu = bf(T, d), bf(2 * ff, d, sc=0.02)
    gu = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
    h = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
    out = {}
    ms = _event_time(lambda: L.check(lib.gget_op_gateup_geglu(P(x), P(wgu), P(gu), P(h), T, d, ff, st)), iters)
    out["gateup"] = (2.0 * T * d * 2 * ff, ms)
    dgu, dxn = bf(T, 2 * ff), torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
    ms = _event_time(lambda: L.check(lib.gget_op_gemm(L.GEMM_NN, L.EPI_NONE, P(dgu), P(wgu), P(dxn), None, T, d, 2 * ff, 2 * ff, d, d, 1, st)), iters)
    out["dgrad_gu"] = (2.0 * T * d * 2 * ff, ms)
    dy, dqkv, xn, attn = bf(T, d), bf(T, 3 * d), bf(T, d), bf(T, d)
    gw = [torch.empty(2 * ff, d, dtype=torch.bfloat16, device="cuda"), torch.empty(d, ff, dtype=torch.bfloat16, device="cuda"),
          torch.empty(3 * d, d, dtype=torch.bfloat16, device="cuda"), torch.empty(d, d, dtype=torch.bfloat16, device="cuda")]
    probs = [(dgu, xn, gw[0], 2 * ff, d, T, 2 * ff, d, d), (dy, h, gw[1], d, ff, T, d, ff, ff),
             (dqkv, xn, gw[2], 3 * d, d, T, 3 * d, d, d), (dy, attn, gw[3], d, d, T, d, d, d)]
    ms = _event_time(lambda: L.check(L.gemm_grouped(lib, L.GEMM_TN, probs, st)), iters)
    out["wgrad_layer"] = (2.0 * T * (2 * ff * d + d * ff + 4 * d * d), ms)
    return out


def time_per_sample_kernels(spec, B, S, lengths, p_drop, iters=20):
    """S <= 32: the per-sample launches of a decoder layer (attention + o projection + residual + RMSNorm forward; RMSNorm backward + o
    dgrad + attention backward) stand-alone through their operator entries, on the batch's own sample lengths (var-len rows), HIP-event
    timed.  What bounds them is the stream of the fragment-major o weight through every sample's vector cache (B x d x d x 2 bytes of
    L2 -> CU traffic per launch), reported next to their algorithmic HBM bytes."""
    L = importlib.import_module("graph-gpt_amd._lib")
    import ctypes as C
    import numpy as np
    lib = L.load()
    H, d = spec.num_heads, spec.hidden_size
    if S > 64 or H not in (2, 4, 8, 12) or lengths is None:
        return None
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lens = torch.from_numpy(np.asarray(lengths, dtype=np.int32))
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = (int(cu[-1]) + 63) // 64 * 64
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, device="cuda") * sc).to(torch.bfloat16)
    emp = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device="cuda")
    qkv, wo, x, nw = bf(T, 3 * d), bf(d, d, sc=0.03), bf(T, d), bf(d)
    wo_f, wo_b = emp(d, d), emp(d, d)
    L.check(lib.gget_op_pack_wo(P(wo), 0, P(wo_f), P(wo_b), d, 1, st))
    attn, xmid, xn, dxmid, dqkv = emp(T, d), emp(T, d), emp(T, d), emp(T, d), emp(T, 3 * d)
    dxn, dres = bf(T, d, sc=0.5), bf(T, d, sc=0.5)
    dattn_long = emp(T, d)
    lse = torch.empty(B * H * S, dtype=torch.float32, device="cuda")
    rstd = torch.empty(T, dtype=torch.float32, device="cuda")
    dw = torch.zeros(16 * 1024, dtype=torch.float32, device="cuda")
    lens_d, rb = lens.cuda(), cu[:B].contiguous().cuda()
    taken = C.c_int32(0)
    fwd = lambda: L.check(lib.gget_op_attn_oproj_fwd(P(qkv), P(lens_d), P(rb), P(attn), P(lse), P(wo_f), P(x), P(xmid), P(nw), P(xn), P(rstd), B, S, H, 0,
                                                     1e-6, p_drop, 7, st, C.byref(taken)))
    bwd = lambda: L.check(lib.gget_op_attn_oproj_bwd(P(dxn), P(xmid), P(nw), P(rstd), P(dres), P(dxmid), P(dw), 16, 1024, P(wo_b), P(qkv), P(lse), P(lens_d),
                                                     P(rb), P(dqkv), B, S, H, 0, None, None, None, p_drop, 7, T, st, C.byref(taken), P(dattn_long)))
    fwd()
    if not taken.value:
        return None
    f_ms, b_ms = _event_time(fwd, iters), _event_time(bwd, iters)
    stream = float(B) * d * d * 2
    hbm_f = 2.0 * T * (3 * d + 3 * d + d)          # qkv, x_in read; attn_out, x_mid, xn written
    hbm_b = 2.0 * T * (3 * d + d + 3 * d + 3 * d)  # dxn, x_mid, dres, qkv read; dx_mid, dqkv written
    return {"what": "S <= 32 (var-len layout: S <= 64, every sample by its own row count): one workgroup per sample - attention of all heads + o projection + residual + RMSNorm (forward), RMSNorm backward + o dgrad + "
                    "attention backward; each replaces three launches of a decoder layer (attention.hip: attn_oproj_fwd_kernel / attn_oproj_bwd_kernel)",
            "rows": T, "samples": B, "timed": "stand-alone loop, HIP events",
            "forward": {"avg_launch_ms": f_ms, "weight_stream_TBps": stream / (f_ms * 1e-3) / 1e12, "algorithmic_hbm_bytes": hbm_f,
                        "hbm_GBps": hbm_f / (f_ms * 1e-3) / 1e9},
            "backward": {"avg_launch_ms": b_ms, "weight_stream_TBps": stream / (b_ms * 1e-3) / 1e12, "algorithmic_hbm_bytes": hbm_b,
                         "hbm_GBps": hbm_b / (b_ms * 1e-3) / 1e9},
            "bound": "L2 -> CU stream of the fragment-major o weight (B x d x d x 2 bytes per launch, every byte an L2 hit; the vector caches' 64 B/clk x "
                     "256 CUs = 39 TB/s is the ceiling, ~10 TB/s measured with all workgroups streaming the same lines)"}


def _source_digest():
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "common.h", "gemm.h"):
        with open(os.path.join(ROOT, "graph-gpt_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(which="wgrad_layer", rows=8192):
    """L2<->fabric bytes per launch of a timed kernel from the committed PMC passes (FETCH_SIZE x2 (gfx950 correction) +
    WRITE_SIZE, separate --pmc runs: tools/pmc_wgrad.sh -> profiles/r0N_wgrad_gemm_pmc*.json, tools/pmc_gu.sh ->
    profiles/r0N_gu_geglu_gemm_pmc*.json).  PMC counters cannot be collected from inside the process being timed, so the figure
    is a recorded one: it carries the digest of the kernel sources and the row count (T, = K of the weight-gradient launch) it was
    measured on and is reported as null when no record matches both."""
    import glob
    pat = {"wgrad_layer": "r0*_wgrad_gemm_pmc*.json", "gateup": "r0*_gu_geglu_gemm_pmc*.json", "dgrad_gu": "r0*_dxn2_gemm_pmc*.json"}.get(which)
    if pat is None:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get("source_digest") == _source_digest() and int(rec.get("rows", 8192)) == int(rows):
                return rec["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


KERNEL_NAMES = {
    "wgrad_layer": "gemm_ks_kernel<192,192,TN> (in-block K split): grouped weight gradients of one decoder layer - dW = dY^T X for "
                   "gate|up, down, q|k|v, o in ONE launch, K = T, 256 tiles = one per CU",
    "gateup": "gemm_persist_kernel<192 or 256,256,64,NT,EPI_GEGLU_FWD> (192-row tiles at T = 5696): FFN gate|up [T,d]x[d,2ff] with the gated-GELU product in its epilogue",
    "dgrad_gu": "gemm_ks_kernel<64, 96 or 128,192,NN> (in-block K split; 96-row tiles at T = 5696): dgrad dxn2 = dgu W_gu [T,2ff]x[2ff,d]",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="pcqm4m-v2-pretrain-base", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=4,
                    help="distinct synthetic batches the steps rotate through (pre-train workloads; every batch has its own lengths, so the "
                         "var-len row count - and with it the launch shapes - changes from step to step like in a real epoch; 1 = the "
                         "single repeated batch of rounds 1-3)")
    ap.add_argument("--seq-len", type=int, default=0,
                    help="padded width S of the pre-train workloads (SURVEY 8d: the C1 sweep S = 24 / 32 / 40 / 56).  The collator pads a batch to "
                         "8 * ceil(max_len / 8) of its longest graph (reference src/data/collator.py:70-111), so real PCQM4M-v2 batches of 256 graphs "
                         "are rarely S <= 32; the lengths keep the workload's distribution (clipped N(22, 6), one graph of the batch at the full width). "
                         "0 = the workload's own S (the headline line)")
    ap.add_argument("--per-gpu-batch", type=int, default=0,
                    help="graphs per GPU and step instead of the workload's own B (256 for the headline: the reference's batch size); a shape knob "
                         "for sweeps like --seq-len / --mean-len, never the headline line")
    ap.add_argument("--mean-len", type=float, default=22.0,
                    help="mean of the graphs' token count (clipped N(mean, 6); 22 = the workload's own): moves the var-len row count of the "
                         "batches, the shape every GEMM launch plan is chosen for (tools/rows_sweep.py)")
    ap.add_argument("--long-tail", type=float, default=0.0,
                    help="fraction of the graphs redrawn uniformly from (32, S] (a heavier long tail than the clipped normal's; needs --seq-len > 32)")
    ap.add_argument("--layout", default="varlen", choices=["varlen", "varlen-count", "padded"],
                    help="token layout of the timed steps.  varlen (default): the call of the reference's own step - device-resident "
                         "tensors, no token count passed - the engine counts the mask on the device and runs the padding-free layout; "
                         "varlen-count: the same layout with the count handed over by the caller (data['num_tokens']: no device->host "
                         "read); padded: every row of the padded [B,S] grid")
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
        # (before the communicator exists: NCCL_MAX_NCHANNELS = the CUs the GEMM launches leave free, graph-gpt_amd/training.py dp_env_defaults)
        importlib.import_module("graph-gpt_amd.training").dp_env_defaults()
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, init_method="env://")

    spec_mod = importlib.import_module("graph-gpt_amd.spec")
    weights = importlib.import_module("graph-gpt_amd.weights")
    synth = importlib.import_module("graph-gpt_amd.synth")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    training = importlib.import_module("graph-gpt_amd.training")

    kind, size, B, S, F, V = WORKLOADS[a.workload]
    if a.seq_len:
        assert kind == "pt", "--seq-len: pre-train workloads"
        S = a.seq_len
    if a.per_gpu_batch:
        B = a.per_gpu_batch
    tail = dict(long_tail=a.long_tail) if a.long_tail > 0 else {}
    if a.mean_len != 22.0:
        tail["mean_len"] = a.mean_len
    sz = spec_mod.MODEL_SIZES[size]
    pt = kind.startswith("pt")
    extra = {}
    if kind == "ft":      # ogbl-ppa scripts: LayerScale + stochastic depth (examples/graph_lvl/ppa_*.sh), 2-class edge task
        extra = dict(layer_scale_init_value=1.0, path_pdrop=0.2, num_labels=2, problem_type="single_label_classification")
    elif kind == "ft-long":
        extra = dict(num_labels=2, problem_type="single_label_classification")
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=sz["hidden_size"], intermediate_size=4 * sz["hidden_size"],
                                  num_hidden_layers=sz["num_layers"], num_attention_heads=sz["hidden_size"] // 64,
                                  max_position_embeddings=max(1024, S), causal_attention=False, stacked_feat=F,
                                  next_n_token=F if pt else 1,
                                  attention_dropout=0.1, **extra)   # every reference training script runs attention dropout 0.1
    model = (modeling.GraphGPTPretrainBase if pt else modeling.GraphGPTTaskModel)(cfg, seed=0)   # same init on every rank
    spec = model.spec
    model._ensure_engine(B, S)
    engine = training.initialize(model, training.OptimConfig(lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1,
                                                             max_grad_norm=1.0))
    extra_batches = []
    if kind == "pt":
        batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234 + rank, **tail)     # distinct data per rank
        real_tokens = synth.real_tokens(batch)
        extra_batches = [synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234 + rank + 1000 * i, **tail) for i in range(1, max(1, a.batches))]
    elif kind == "pt-packed":
        batch = synth.make_packed_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234 + rank, mean_len=22, min_len=6)
        real_tokens = int(batch["lengths"].sum())
    else:
        mk_task = lambda sd: synth.make_task_batch(B=B, S=S, F=F, V=V, seed=sd, lengths="uniform" if kind == "ft" else "full", min_len=S // 4)
        batch = mk_task(1234 + rank)
        real_tokens = synth.real_tokens(batch)
        # (a rotation here as well since round 6: on ONE repeated batch the fine-tune model memorises it within 25 steps - loss 2.8e-6 /
        #  6.0e-8 in profiles/r05_other_workloads.json - harmless for the timing, useless as a loss sanity figure)
        extra_batches = [mk_task(1234 + rank + 1000 * i) for i in range(1, max(1, a.batches))]
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k not in ("lengths", "segments")}
    # the rotation: batch 0 (seed 1234 + rank: the batch of the probes, the parity leg and the CPU baseline) and the extra ones
    host_batches = [batch] + extra_batches
    devs = [dev] + [{k: torch.from_numpy(v).cuda() for k, v in b_.items() if k not in ("lengths", "segments")} for b_ in extra_batches]
    reals = [int(real_tokens)] + [int(synth.real_tokens(b_)) for b_ in extra_batches]
    nb = len(devs)
    # Token layout of the step.  "varlen" (default): the batch carries its real-token count (a host int the collator knows:
    # sum(attention_mask)), and the engine runs embedding, layer stack and backward on the compacted real tokens instead of the padded
    # [B,S] grid (include/gget.h: gget_set_token_count) - same batch, same loss, same gradients.  "padded": every row of the grid,
    # as the reference computes it.  Packed batches (3-D masks) have no padding to skip.
    layout = a.layout if kind != "pt-packed" else "padded"

    def make_step(lay):
        datas = [dict(d_, num_tokens=r_) if lay == "varlen-count" else d_ for d_, r_ in zip(devs, reals)]

        def one(i=0):
            data = datas[i % nb]
            if lay == "padded":
                os.environ["GGET_VARLEN"] = "0"
            try:
                return training.batch_training(data, engine) if pt else training.ft_batch_training(data, engine)[0]
            finally:
                if lay == "padded":
                    del os.environ["GGET_VARLEN"]
        return one

    step = make_step(layout)

    # eval-mode forward of the bench batch on the initial (seed 0) weights: the loss the oracle leg is compared with (`loss_parity`)
    gpu_eval_loss = None
    if pt and kind == "pt":
        model.eval()
        with torch.no_grad():
            gpu_eval_loss = float(model(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"]).head1_loss)
        model.train()

    # N > 1: no scaling curve of this engine exists - the job times the exchange arrangements on THIS machine first
    # (10 steps each: collectives beside the backward / behind it / beside it with 32 CUs left free by the GEMM plans) and runs the
    # reported steps on the fastest; GGET_DP_PROBE=0 or an explicit GGET_DP_OVERLAP / GGET_DP_RESERVE_CUS keeps what the environment says
    menu_probe = None
    if world > 1 and bool(int(os.environ.get("GGET_DP_PROBE", "1"))) and "GGET_DP_OVERLAP" not in os.environ and "GGET_DP_RESERVE_CUS" not in os.environ:
        probe_i = [0]

        def probe_step():
            probe_i[0] += 1
            return step(probe_i[0])
        try:
            menu_probe = engine.probe_dp_menu(probe_step, steps=int(os.environ.get("GGET_DP_PROBE_STEPS", "10")))
        except Exception as ex:      # (the probe must never cost the run its line: defaults then, and the reason in the line)
            engine.set_dp_menu(overlap=True, reserve_cus=0)
            menu_probe = {"menus": [], "chosen": {"overlap": True, "reserve_cus": 0}, "probed": False, "error": repr(ex)[:300]}
    for i in range(a.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the other token layouts, same batch, same K steps each (outside the reported time; single process only)
    layouts = None
    if world == 1 and kind != "pt-packed":
        layouts = {layout: dt / a.steps * 1e3}
        for lay in ("varlen", "varlen-count", "padded"):
            if lay in layouts:
                continue
            other = make_step(lay)
            for i in range(2):
                other(a.warmup - 2 + i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(a.steps):
                other(a.warmup + i)      # (the same rotation as the reported steps)
            torch.cuda.synchronize()
            layouts[lay] = (time.perf_counter() - t1) / a.steps * 1e3
        step(0)     # (the probes below run on the reported layout again, on batch 0)
        torch.cuda.synchronize()
    # The batch hand-over inside the number: the same K steps once more, outside the reported time, fed with HOST
    # batches - nothing pre-staged - (a) through the product's DevicePrefetcher (pinned staging of batch t + 1 during step t, copy kernels in
    # the compute queue: what TrainingPipeline's loop does) and (b) the reference's way, every tensor `.to(device)` at the top of the step
    # (training_utils.py:17-26).  `value` stays the device-resident figure the contract asks for; these two sit next to it.
    h2d = None
    if world == 1 and kind != "pt-packed":
        host_t = [{k: torch.from_numpy(v) for k, v in b_.items() if k not in ("lengths", "segments")} for b_ in host_batches]
        run_on = (lambda d_: training.batch_training(d_, engine)) if pt else (lambda d_: training.ft_batch_training(d_, engine)[0])

        def feed(n, start):
            for i in range(n):
                yield host_t[(start + i) % nb]
        it = iter(training.DevicePrefetcher(feed(2 + a.steps, a.warmup - 2), torch.device("cuda", local)))
        for _ in range(2):
            run_on(next(it))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for d_ in it:
            run_on(d_)
        torch.cuda.synchronize()
        pre_ms = (time.perf_counter() - t1) / a.steps * 1e3
        for i in range(2):
            run_on({k: v.cuda() for k, v in host_t[(a.warmup - 2 + i) % nb].items()})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.steps):
            run_on({k: v.cuda() for k, v in host_t[(a.warmup + i) % nb].items()})
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t1) / a.steps * 1e3
        nbytes = sum(v.numel() * v.element_size() for v in host_t[0].values())
        h2d = {"h2d_inclusive_ms_per_step": pre_ms, "synchronous_to_device_ms_per_step": sync_ms, "batch_bytes": nbytes,
               "what": "host batches in, nothing pre-staged, same K steps and rotation: h2d_inclusive = graph-gpt_amd.training.DevicePrefetcher (pinned "
                       "staging one batch ahead, one copy kernel per tensor in the compute queue in front of the step; the host-side mask sum "
                       "rides along as num_tokens); synchronous = every tensor .cuda() from pageable memory at the top of the step, the "
                       "reference's batch_training"}
        step(0)
        torch.cuda.synchronize()
    # N > 1: how much of the step is gradient exchange the backward does not hide - the same K steps once more (outside the reported
    # time) with the collectives switched off (GgetEngine.exchange = False: identical kernels on the compute stream, nothing on the
    # side stream), max over ranks; exposed = reported step - that.  The replicas drift apart in these steps: they come last.
    dp_info = None
    replicas_identical = None
    if world > 1:
        # replicas must still be bit-identical after the timed steps (same init, averaged gradients, same update on every rank): a
        # 64-bit checksum of the fp32 master weights, min- and max-reduced over the ranks
        chk = model._engine.master.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        replicas_identical = bool(int(lo_[0]) == int(hi_[0]))
        engine.exchange = False
        for i in range(2):
            step(a.warmup - 2 + i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)       # (the same rotation as the reported steps)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt_nox = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt_nox, op=dist.ReduceOp.MAX)
        engine.exchange = True
        dp_info = engine.describe_dp()
        dp_info["menu_probe"] = menu_probe
        dp_info["ms_per_step_without_exchange"] = float(dt_nox[0]) / a.steps * 1e3
        dp_info["replicas_bit_identical"] = replicas_identical
        # which device every rank ran on (the driver's SCALE run must see N distinct GPUs): name, index, PCI bus id / uuid
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        me = {"rank": rank, "device_index": torch.cuda.current_device(), "name": pr.name,
              "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")) or None}
        allr = [None] * world
        dist.all_gather_object(allr, me)
        dp_info["rank_devices"] = allr
        dp_info["distinct_devices"] = len({(d_["device_index"], d_["pci_bus_id"], d_["uuid"]) for d_ in allr})
    # the dominant kernels' launch durations INSIDE a step: three more (untimed) steps with HIP events around those launches, on the
    # stream they run on (single process only - the extra steps would otherwise need every rank)
    in_step = {}
    gemm_info = None
    if world == 1:
        eng_ = model._engine
        eng_.debug_probe(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        w_ms, g_ms = eng_.debug_probe(False)
        in_step = {k: v for k, v in (("wgrad_layer", w_ms), ("gateup", g_ms)) if v > 0}
        # every GEMM launch of three more steps between events: the TIME-WEIGHTED rate of the step's GEMMs (sum of 2 M N K over the
        # sum of the launch durations; the SMTP head's device-sized launches are counted but left out of both sums)
        import ctypes as C_
        L = importlib.import_module("graph-gpt_amd._lib")
        lib_ = L.load()
        L.check(lib_.gget_debug_gemm_probe(1, None, None, None, None))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        fl_, ms_, n_, sk_ = C_.c_double(), C_.c_double(), C_.c_int32(), C_.c_int32()
        L.check(lib_.gget_debug_gemm_probe(0, C_.byref(fl_), C_.byref(ms_), C_.byref(n_), C_.byref(sk_)))
        if ms_.value > 0:
            gemm_info = {"launches_per_step": n_.value // 3, "device_sized_launches_left_out_per_step": sk_.value // 3,
                         "ms_per_step": ms_.value / 3, "tflops_time_weighted": fl_.value / (ms_.value * 1e-3) / 1e12,
                         "frac_of_peak_time_weighted": fl_.value / (ms_.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "flops_per_step": fl_.value / 3,
                         "note": "executed rows (var-len layout: real tokens); HIP events around every launch, untimed extra steps"}
    timed = [(a.warmup + i) % nb for i in range(a.steps)]        # batch index of every timed step
    real_per_step = sum(reals[j] for j in timed) / a.steps           # real tokens of this rank per timed step (mean over the rotation)
    stats = torch.tensor([dt, float(real_per_step), float(loss.item())], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, tot_real, mean_loss = float(mx[0]), float(sm[1]), float(sm[2]) / world
    else:
        tot_real, mean_loss = float(real_per_step), float(loss.item())

    if rank == 0:
        M, Lm = model._engine.head_counts() if pt else (0, 0)       # (of the last forward)
        ran_varlen, t_rows, mismatch = model._engine.varlen_status()
        assert not mismatch, "the batch's real-token count handed to the engine disagrees with its attention mask"
        if kind == "pt" and world == 1:                             # (single process: the last forward was a probe step on batch 0)
            lab0 = host_batches[0]["labels"] != -100
            assert (M, Lm) == (int(lab0.any(-1).sum()), int(lab0.sum())), "head counts of batch 0 differ from its labels"
        if ran_varlen:
            t_rows = (reals[0] + 63) // 64 * 64                     # rows of batch 0: the shape the kernel timings below are taken on

        def batch_flops(j, executed):
            """FLOPs of one step on batch j (M / Lm from its labels; executed: the var-len rows and len x len attention blocks)"""
            b_ = host_batches[j]
            if kind == "pt":
                lab = b_["labels"] != -100
                m_, lm_ = int(lab.any(-1).sum()), int(lab.sum())
            else:
                m_, lm_ = M, Lm
            if not executed or not ran_varlen:
                return flops_per_step(spec, B, S, m_, lm_, kind)
            return flops_per_step(spec, B, S, m_, lm_, kind, rows=(reals[j] + 63) // 64 * 64, lengths=b_.get("lengths"))
        fstep = sum(batch_flops(j, False) for j in timed) / a.steps      # the reference's computation, padded grid
        fexec = sum(batch_flops(j, True) for j in timed) / a.steps       # what the engine executed (mean over the timed steps)
        ms = dt / a.steps * 1e3
        step_tflops = fstep / (ms * 1e-3) / 1e12
        kt = time_kernels(spec, t_rows)
        L_ = spec.num_layers
        share = lambda fl_per_layer: round(100.0 * fl_per_layer * L_ / fstep, 1)
        # the dominant kernel = the launch that takes the largest share of the step's TIME (each of the three runs once per
        # layer): with the GEGLU fused into the gate|up GEMM that is the grouped weight-gradient launch (~20 % of a C1 step)
        c1_shape = B * S == 8192 and spec.hidden_size == 768

        def entry(k):
            fl, alone = kt[k]
            kms = in_step.get(k, alone)      # inside the step when the probe covers the kernel, else the stand-alone loop
            tf = fl / (kms * 1e-3) / 1e12
            return {"kernel": KERNEL_NAMES[k], "step_flops_pct": share(fl), "step_time_pct": round(100.0 * kms * L_ / ms, 1),
                    "avg_launch_ms": kms, "timed": "HIP events around the launch inside the step" if k in in_step else "stand-alone loop",
                    "avg_launch_ms_standalone": alone, "achieved": tf, "frac": tf / PEAK_BF16_TFLOPS,
                    "traffic": pmc_traffic(k, t_rows) if c1_shape else None}
        order = sorted(kt, key=lambda k: -in_step.get(k, kt[k][1]))
        dom = entry(order[0])
        roofline = {"bound": "mfma", "kernel": dom["kernel"] + f" ({dom['step_time_pct']}% of the step's time, {dom['step_flops_pct']}% of its FLOPs)",
                    "achieved": dom["achieved"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                    "traffic": dom["traffic"], "avg_launch_ms": dom["avg_launch_ms"], "timed": dom["timed"],
                    "avg_launch_ms_standalone": dom["avg_launch_ms_standalone"],
                    "other_kernels": [entry(k) for k in order[1:]]}
        names = {"pt": "SMTP loss", "pt-packed": "SMTP loss", "ft": "task loss", "ft-long": "task loss"}
        out = {
            "metric": f"graph-tokens/sec (un-padded Eulerian tokens, whole job) + {names[kind]}, "
                      + ("PCQM4M-v2 base pre-train" if a.workload == "pcqm4m-v2-pretrain-base" else a.workload),
            "value": tot_real * a.steps / dt, "unit": "graph-tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" + (f" ({nb} batches in rotation)" if nb > 1 else ""),
            "config": {"workload": a.workload, "model": f"{size} d{spec.hidden_size}/L{spec.num_layers}/H{spec.num_heads} "
                       f"({spec.num_params() / 1e6:.1f}M params)", "per_gpu_batch": B, "global_batch": B * world,
                       "seq_len": S, "stacked_feat": F, "vocab": V, "parallelism": f"dp{world}",
                       "step": "fwd+bwd+allreduce+clip+AdamW", "attention_dropout": cfg.attention_dropout,
                       "token_layout": ("varlen (real tokens compacted on the device; same batch, loss and gradients as the padded grid); "
                                        + ("the step is called like the reference's - device tensors, no token count: the engine counts the mask"
                                           if layout == "varlen" else "token count passed by the caller")) if ran_varlen else "padded",
                       **({"path_pdrop": 0.2, "layer_scale_init_value": 1.0} if kind == "ft" else {}),
                       **({"packing": "whole graphs back to back, block-diagonal attention"} if kind == "pt-packed" else {})},
            ("smtp_loss" if pt else "task_loss"): mean_loss,
            "padded_tokens_per_s": B * S * world * a.steps / dt,
            "tokens_per_s_per_gpu": tot_real * a.steps / dt / world,
            # frac_of_peak: what the engine EXECUTED (var-len layout: the real-token rows, len x len attention blocks) over the step
            # time and the dense bf16 peak - the matrix pipes' duty.  reference_grid: SURVEY 8(d)'s accounting - the FLOPs the reference
            # spends on this batch (every row of the padded [B,S] grid, full S x S attention) over the same time: a throughput
            # equivalent (pad-row work is never executed here), NOT an MFMA utilisation.
            "step_mfma": {"flops_per_step": fexec, "achieved_tflops_per_gpu": fexec / (ms * 1e-3) / 1e12,
                          "frac_of_peak": fexec / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                          "token_layout": "varlen" if ran_varlen else "padded", "rows": t_rows, "padded_rows": B * S,
                          "rows_per_batch": [(r_ + 63) // 64 * 64 if ran_varlen else B * S for r_ in reals],
                          "reference_grid": {"flops_per_step": fstep, "equivalent_tflops_per_gpu": step_tflops,
                                             "equivalent_frac_of_peak": step_tflops / PEAK_BF16_TFLOPS}},
            "roofline": roofline,
        }
        if gemm_info is not None:
            out["step_mfma"]["gemms"] = gemm_info
        if world == 1 and kind == "pt":
            psk = time_per_sample_kernels(spec, B, S, host_batches[0].get("lengths"), cfg.attention_dropout)
            if psk is not None:
                out["per_sample_kernels"] = psk
        if dp_info is not None:
            dp_info["exposed_comm_ms"] = ms - dp_info["ms_per_step_without_exchange"]
            out["dp"] = dp_info
        if h2d is not None:
            h2d["vs_device_resident"] = h2d["h2d_inclusive_ms_per_step"] / ms
            out["h2d_inclusive_ms_per_step"] = h2d["h2d_inclusive_ms_per_step"]
            out["batch_handover"] = h2d
        if layouts is not None:
            out["layouts"] = {"ms_per_step": layouts, "reported": layout,
                              "note": "varlen = reference-shaped call (device tensors only; the engine sums the mask and reads 4 bytes back), "
                                      "varlen-count = caller passes data['num_tokens'], padded = every row of the [B,S] grid; same batch, same "
                                      "K steps each, outside the reported time"}
        if world == 1 and not a.no_cpu_baseline and kind == "pt":
            state = weights.make_state_dict(spec, seed=0)
            out["cpu_baseline"], oracle_loss = cpu_baseline(spec, state, B, S, F, V, 1234, min(os.cpu_count() or 1, 32), tail=tail)
            if gpu_eval_loss is not None:
                out["loss_parity"] = {"gpu_eval_loss": gpu_eval_loss, "oracle_loss": oracle_loss,
                                      "rel": abs(gpu_eval_loss - oracle_loss) / abs(oracle_loss),
                                      "what": "SMTP loss of the bench batch (seed 1234) on the initial weights (seed 0, rounded to bf16), "
                                              "eval mode: HIP engine forward vs the CPU oracle's forward"}
        elif world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = None   # the CPU port is timed on the headline workload only
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 is still timing the dominant kernel: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
