#!/usr/bin/env python3
"""q|k|v projection with and without the RoPE epilogue (HIP events), T env (default 8192), S = 32."""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, S = int(os.environ.get("T", 8192)), 768, 32
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
x, w = rn(T, d), rn(3 * d, d, sc=0.05)
qkv = torch.empty(T, 3 * d, dtype=torch.bfloat16, device="cuda")
cos = torch.rand(1024, 32, device="cuda"); sin = torch.rand(1024, 32, device="cuda")
ops = {"plain q|k|v": lambda: lib.gget_op_gemm(L.GEMM_NT, 0, P(x), P(w), P(qkv), None, T, 3 * d, d, d, d, 3 * d, 1, st),
       "q|k|v + RoPE": lambda: lib.gget_op_qkv_rope(P(x), P(w), P(qkv), P(cos), P(sin), None, T, S, d, st)}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(3):
    for name, fn in ops.items():
        L.check(fn())
        e0.record()
        for _ in range(20): L.check(fn())
        e1.record(); torch.cuda.synchronize()
        print(f"{name:14s} T={T}: {e0.elapsed_time(e1) * 50:7.1f} us", flush=True)
