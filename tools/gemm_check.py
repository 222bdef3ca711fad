#!/usr/bin/env python3
"""Correctness + timing of one GEMM shape through gget_op_gemm against torch (fp32 reference of the same bf16 inputs);
used to A/B kernel variants selected by environment knobs.  usage: gemm_check.py [M N K]"""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = int(os.environ.get("GEMM_ITERS", "30"))
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (8192, 6144, 768)
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
args = (L.GEMM_NT, 0, P(A), P(B), P(Cm), None, M, N, K, K, K, N, 1, st)
L.check(lib.gget_op_gemm(*args))
ref = A.float() @ B.float().t()
err = float((Cm.float() - ref).norm() / ref.norm())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): L.check(lib.gget_op_gemm(*args))
e0.record()
for _ in range(ITERS): L.check(lib.gget_op_gemm(*args))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / ITERS * 1e3
print(f"M={M} N={N} K={K} rel_err={err:.2e} {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TFLOP/s")
assert err < 5e-3 or os.environ.get("GGET_GEMM_ABLATE")
