"""Same-box A/B of engine variants INSIDE the C1 training step (one process, variants alternated round-robin so clock / thermal drift hits
all of them alike).  A variant is a list of gget_debug_set (key, value) pairs applied before its block of steps, e.g.

    python tools/step_ab.py --variants "base:" "mfma16:1=64" --rounds 5 --steps 10 [--workload pcqm4m-v2-pretrain-base] [--check]

prints ms/step per variant and round, then the medians.  --check: also runs the grouped weight-gradient launch of one layer under every
variant on the same random operands and prints the largest element-wise difference to the first variant (bit-equal summation orders give 0)."""
import argparse, ctypes as C, importlib, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

L = importlib.import_module("graph-gpt_amd._lib")


def parse(v):
    """name:key=value,...  - integer keys are gget_debug_set keys, `env.NAME=value` sets an environment variable for the variant's steps"""
    name, _, rest = v.partition(":")
    pairs = []
    for kv in rest.split(","):
        if not kv:
            continue
        k, _, val = kv.partition("=")
        pairs.append((k, val) if k.startswith("env.") else (int(k), int(val)))
    return name, pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", default=["base:", "v64:1=64"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--workload", default="pcqm4m-v2-pretrain-base")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    lib = L.load()
    spec_mod = importlib.import_module("graph-gpt_amd.spec")
    synth = importlib.import_module("graph-gpt_amd.synth")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    training = importlib.import_module("graph-gpt_amd.training")
    kind, size, Bn, S, F, V = B.WORKLOADS[a.workload]
    sz = spec_mod.MODEL_SIZES[size]
    pt = kind.startswith("pt")
    extra = dict(layer_scale_init_value=1.0, path_pdrop=0.2, num_labels=2, problem_type="single_label_classification") if kind == "ft" else \
        (dict(num_labels=2, problem_type="single_label_classification") if kind == "ft-long" else {})
    cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=sz["hidden_size"], intermediate_size=4 * sz["hidden_size"],
                                  num_hidden_layers=sz["num_layers"], num_attention_heads=sz["hidden_size"] // 64,
                                  max_position_embeddings=max(1024, S), causal_attention=False, stacked_feat=F, next_n_token=F if pt else 1,
                                  attention_dropout=0.1, **extra)
    model = (modeling.GraphGPTPretrainBase if pt else modeling.GraphGPTTaskModel)(cfg, seed=0)
    model._ensure_engine(Bn, S)
    engine = training.initialize(model, training.OptimConfig(lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0))
    if kind == "pt":
        batch = synth.make_pretrain_batch(B=Bn, S=S, F=F, V=V, seed=1234)
    else:
        batch = synth.make_task_batch(B=Bn, S=S, F=F, V=V, seed=1234, lengths="uniform" if kind == "ft" else "full", min_len=S // 4)
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k not in ("lengths", "segments")}
    dev["num_tokens"] = synth.real_tokens(batch)
    step = (lambda: training.batch_training(dev, engine)) if pt else (lambda: training.ft_batch_training(dev, engine)[0])
    variants = [parse(v) for v in a.variants]
    keys = sorted({k for _, ps in variants for k, _ in ps if isinstance(k, int)})
    envs = sorted({k[4:] for _, ps in variants for k, _ in ps if not isinstance(k, int)})

    def apply(pairs):
        for k in keys:
            L.check(lib.gget_debug_set(k, 0 if k != 2 else 1))      # defaults (key 2 = LDS headroom: 1)
        for k in envs:
            os.environ.pop(k, None)
        for k, v in pairs:
            if isinstance(k, int):
                L.check(lib.gget_debug_set(k, v))
            else:
                os.environ[k[4:]] = v

    for _, ps in variants:          # warm every variant's kernels up
        apply(ps)
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    times = {n: [] for n, _ in variants}
    for r in range(a.rounds):
        for n, ps in variants:
            apply(ps)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            times[n].append((time.perf_counter() - t0) / a.steps * 1e3)
        print("round", r, " ".join(f"{n} {times[n][-1]:.3f}" for n, _ in variants), flush=True)
    for n, _ in variants:
        print(f"median {n}: {statistics.median(times[n]):.3f} ms/step  (min {min(times[n]):.3f})")
    if a.check:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        T, d, ff = 5696, 768, 3072
        g = torch.Generator(device="cuda").manual_seed(1)
        bf = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
        dgu, dy, dqkv, xn, h, attn = bf(T, 2 * ff), bf(T, d), bf(T, 3 * d), bf(T, d), bf(T, ff), bf(T, d)
        outs = []
        for n, ps in variants:
            apply(ps)
            gw = [torch.zeros(2 * ff, d, dtype=torch.bfloat16, device="cuda"), torch.zeros(d, ff, dtype=torch.bfloat16, device="cuda"),
                  torch.zeros(3 * d, d, dtype=torch.bfloat16, device="cuda"), torch.zeros(d, d, dtype=torch.bfloat16, device="cuda")]
            probs = [(dgu, xn, gw[0], 2 * ff, d, T, 2 * ff, d, d), (dy, h, gw[1], d, ff, T, d, ff, ff),
                     (dqkv, xn, gw[2], 3 * d, d, T, 3 * d, d, d), (dy, attn, gw[3], d, d, T, d, d, d)]
            L.check(L.gemm_grouped(lib, L.GEMM_TN, probs, st))
            torch.cuda.synchronize()
            outs.append(torch.cat([w.float().flatten() for w in gw]))
        ref = (dgu.float().T @ xn.float()).flatten()
        print("wgrad gate|up rel-L2 vs fp32 matmul:", [float((o[: ref.numel()] - ref).norm() / ref.norm()) for o in outs])
        print("wgrad max |diff| to first variant:", [float((o - outs[0]).abs().max()) for o in outs])


if __name__ == "__main__":
    main()
