#!/usr/bin/env python3
"""A/B timing of the two fused GEGLU GEMMs (and plain gate|up / dh / q|k|v) under gget_debug_set variants, interleaved."""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
variants = [int(x, 0) for x in sys.argv[1:]] or [0]
T, d, ff = int(os.environ.get("T", 8192)), 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
x, wgu, wdown, dy, wqkv = rn(T, d), rn(2 * ff, d, sc=0.05), rn(d, ff, sc=0.05), rn(T, d), rn(3 * d, d, sc=0.05)
gu, h, dgu = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda"), torch.empty(T, ff, dtype=torch.bfloat16, device="cuda"), torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
qkv = torch.empty(T, 3 * d, dtype=torch.bfloat16, device="cuda")
ops = {
    "gateup+geglu": lambda: lib.gget_op_gateup_geglu(P(x), P(wgu), P(gu), P(h), T, d, ff, st),
    "down_dgrad+geglu": lambda: lib.gget_op_down_dgrad_geglu(P(dy), P(wdown), P(gu), P(dgu), None, T, d, ff, st),
    "plain gate|up": lambda: lib.gget_op_gemm(L.GEMM_NT, 0, P(x), P(wgu), P(gu), None, T, 2 * ff, d, d, d, 2 * ff, 1, st),
    "plain dh": lambda: lib.gget_op_gemm(L.GEMM_NN, 0, P(dy), P(wdown), P(h), None, T, ff, d, d, ff, ff, 1, st),
    "plain qkv": lambda: lib.gget_op_gemm(L.GEMM_NT, 0, P(x), P(wqkv), P(qkv), None, T, 3 * d, d, d, d, 3 * d, 1, st),
}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in ops.items():
    times = {v: [] for v in variants}
    for r in range(5):
        for v in variants:
            L.check(lib.gget_debug_set(1, v))
            L.check(fn())
            e0.record()
            for _ in range(10): L.check(fn())
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 100)
    print(f"{name:18s} " + " | ".join(f"v={v:#x} {statistics.median(times[v]):6.1f} us" for v in variants), flush=True)
L.check(lib.gget_debug_set(1, 0))
