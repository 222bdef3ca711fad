#!/usr/bin/env python3
"""Does a data-parallel rank's launch menu survive a collective's workgroups?  ONE GPU, the C1 training step; while the BACKWARD runs on the
compute stream, a stand-in for RCCL's kernel sits on a side stream: N workgroups of 256 threads with the footprint read from librccl's
gfx950 code object (rcclGenericKernel: 264 registers per lane - gget_debug_set(16, 1) - and 19.7 KiB of LDS), resident for the length of the
backward the way the bucketed all-reduce is.  Menus (gget_debug_set):

  single   the single-GPU menu                                   (what a rank would run if nothing were done)
  r4       rounds 2-4: LDS headroom on every CU        (2, 2)
  r5       round 5: 16 CUs left to the collective      (15, 16), (13, 0), (2, 1)
  r5_32    ... 32 CUs left                              (15, 32)
  r5_64    ... 64 CUs left                              (15, 64)

prints ms/step (forward + backward + clip + AdamW, HIP events) without the stand-in and with N = 16 / 32 workgroups of it.
DP_STANDIN_SCHEDULE=300,150,60: instead, the bucketed exchange SCHEDULE of a rank - the staged backward with a stand-in behind every
bucket that lasts as long as a ring all-reduce of that bucket at the given bus bandwidth (GB/s) over 8 ranks."""
import ctypes as C, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
modeling = importlib.import_module("graph-gpt_amd.modeling")
training = importlib.import_module("graph-gpt_amd.training")
synth = importlib.import_module("graph-gpt_amd.synth")
spec_mod = importlib.import_module("graph-gpt_amd.spec")
lib = L.load()
B, S, F, V = 256, 32, 13, 756
sz = spec_mod.MODEL_SIZES["base"]
cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=sz["hidden_size"], intermediate_size=4 * sz["hidden_size"],
                              num_hidden_layers=sz["num_layers"], num_attention_heads=sz["hidden_size"] // 64,
                              max_position_embeddings=1024, causal_attention=False, stacked_feat=F, next_n_token=F, attention_dropout=0.1)
model = modeling.GraphGPTPretrainBase(cfg, seed=0)
model._ensure_engine(B, S)
eng = training.initialize(model, training.OptimConfig(lr=3e-4))
batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
n_tok = int(batch["attention_mask"].sum())
side = torch.cuda.Stream()
scratch = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
RCCL_LDS = 19744
BWD_US = 4300          # the stand-in stays for about the length of the backward
MENUS = {"single": [(15, 0), (13, 1), (2, 1)], "r4": [(15, 0), (13, 1), (2, 2)], "r5": [(15, 16), (13, 0), (2, 1)],
         "r5_32": [(15, 32), (13, 0), (2, 1)], "r5_64": [(15, 64), (13, 0), (2, 1)]}


def run(menu, blocks, steps=12, warm=4):
    for k, v in MENUS[menu]:
        L.check(lib.gget_debug_set(k, v))
    L.check(lib.gget_debug_set(16, 1))

    def step():
        out = eng(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"], num_tokens=n_tok)
        if blocks:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)           # the "collective" starts when the backward does
            L.check(lib.gget_debug_occupy(C.c_void_p(scratch.data_ptr()), scratch.numel(), blocks, RCCL_LDS, BWD_US, C.c_void_p(side.cuda_stream)))
        eng.backward(out.head1_loss)
        if blocks:
            torch.cuda.current_stream().wait_stream(side)     # (the step waits for its collectives before AdamW)
        eng.step()
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def run_schedule(menu, blocks, gbps, world=8, steps=12, warm=4):
    """The exchange SCHEDULE of a data-parallel rank: the staged backward (head, layer L-1 ... 0, embeddings), behind every bucket a stand-in
    that stays as long as a ring all-reduce of the bucket would at `gbps` GB/s of bus bandwidth (+ 30 us of latency) over `world` ranks."""
    for k, v in MENUS[menu]:
        L.check(lib.gget_debug_set(k, v))
    L.check(lib.gget_debug_set(16, 1))
    e = model._engine
    nl = e.spec.num_layers
    main = torch.cuda.current_stream()

    eng.bucket_mb = float(os.environ.get("DP_STANDIN_BUCKET_MB", "0"))      # (GGET_DP_BUCKET_MB: consecutive buckets in ONE collective)
    eng._groups = None
    groups = eng.exchange_groups(e)

    def bucket(b):
        if not blocks or b not in groups:
            return
        us = int(30 + 2.0 * (world - 1) / world * groups[b][1] * 2 / (gbps * 1e3))
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        L.check(lib.gget_debug_occupy(C.c_void_p(scratch.data_ptr()), scratch.numel(), blocks, RCCL_LDS, us, C.c_void_p(side.cuda_stream)))

    def step():
        eng(input_ids=dev["input_ids"], attention_mask=dev["attention_mask"], labels=dev["labels"], num_tokens=n_tok)
        e.backward_begin()
        bucket(0)
        for i in range(nl - 1, -1, -1):
            e.backward_layer(i)
            bucket(nl - i)
        e.backward_end()
        bucket(nl + 1)
        if blocks:
            main.wait_stream(side)
        eng.step()
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


rows = []
if os.environ.get("DP_STANDIN_SCHEDULE"):      # "gbps[,gbps...]": the bucketed schedule instead of one stand-in for the whole backward
    mb = [round(c * 2 / 2 ** 20, 1) for _, c in model._engine.buckets]
    print("buckets (MiB, completion order):", mb)
    for gbps in [float(x) for x in os.environ["DP_STANDIN_SCHEDULE"].split(",")]:
        for menu in os.environ.get("DP_STANDIN_MENUS", "single,r4,r5_32").split(","):
            for blocks in (0, 16):
                ms = [run_schedule(menu, blocks, gbps) for _ in range(2)]
                rows.append({"schedule_gbps": gbps, "menu": menu, "standin_workgroups": blocks, "ms_per_step": ms})
                print(f"schedule at {gbps:.0f} GB/s bus bandwidth, 8 ranks: menu {menu:6s} stand-in workgroups {blocks:3d}: {ms[0]:.3f} / {ms[1]:.3f} ms/step", flush=True)
    for k, v in MENUS["single"] + [(16, 0)]:
        L.check(lib.gget_debug_set(k, v))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dp_standin_schedule.json")
    json.dump({"rows": rows, "buckets_mib": mb}, open(out, "w"), indent=1)
    sys.exit(0)
ONLY = os.environ.get("DP_STANDIN_ONLY")      # "menu:blocks" - one configuration (for a kernel trace)
for rnd in range(2):
    for menu in ("single", "r4", "r5", "r5_32", "r5_64"):
        for blocks in (0, 16, 32):
            if ONLY and ONLY != f"{menu}:{blocks}":
                continue
            ms = run(menu, blocks)
            rows.append({"round": rnd, "menu": menu, "standin_workgroups": blocks, "ms_per_step": ms})
            print(f"round {rnd} menu {menu:6s} stand-in workgroups {blocks:3d}: {ms:.3f} ms/step", flush=True)
for k, v in MENUS["single"] + [(16, 0)]:
    L.check(lib.gget_debug_set(k, v))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dp_standin.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({"standin": {"threads": 256, "registers_per_lane": 264, "lds_bytes": RCCL_LDS, "resident_us": BWD_US,
                       "from": "librccl.so gfx950 code object: rcclGenericKernel 261-280 VGPRs, 19744 B LDS, 256 threads"}, "rows": rows},
          open(out, "w"), indent=1)
