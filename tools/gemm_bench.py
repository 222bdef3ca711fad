#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel through the C ABI on the shapes of the C1 workload (HIP-event timed)."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def bench(mode, M, N, K, iters=20, epi=0):
    g = torch.Generator(device="cuda").manual_seed(0)
    if mode == L.GEMM_NT: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g); lda, ldb = K, K
    elif mode == L.GEMM_NN: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g); lda, ldb = K, N
    else: A = torch.randn(K, M, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g); lda, ldb = M, N
    A = A.to(torch.bfloat16); B = (B * 0.05).to(torch.bfloat16)
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    R = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 1 else None
    args = (mode, epi, P(A), P(B), P(Cm), P(R) if R is not None else None, M, N, K, lda, ldb, N, 1, st)
    for _ in range(3): L.check(lib.gget_op_gemm(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.check(lib.gget_op_gemm(*args))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12

T, d, ff = int(os.environ.get('GGET_T', '8192')), 768, 3072   # GGET_T: rows
shapes = [("NT qkv", L.GEMM_NT, T, 3 * d, d, 0), ("NT gu", L.GEMM_NT, T, 2 * ff, d, 0), ("NT o+res", L.GEMM_NT, T, d, d, 1),
          ("NT down+res", L.GEMM_NT, T, d, ff, 1), ("NN dh", L.GEMM_NN, T, ff, d, 0), ("NN dxn2", L.GEMM_NN, T, d, 2 * ff, 0),
          ("NN dattn", L.GEMM_NN, T, d, d, 0), ("NN dxn1", L.GEMM_NN, T, d, 3 * d, 0), ("TN dWgu", L.GEMM_TN, 2 * ff, d, T, 0),
          ("TN dWdown", L.GEMM_TN, d, ff, T, 0), ("TN dWqkv", L.GEMM_TN, 3 * d, d, T, 0), ("TN dWo", L.GEMM_TN, d, d, T, 0)]
extra = [("NT gu K=%d" % k, L.GEMM_NT, T, 2 * ff, k, 0) for k in (192, 384, 1536, 3072, 6144)]
print("ablate =", os.environ.get("GGET_GEMM_ABLATE", "0"))
for name, mode, M, N, K, epi in shapes + (extra if "--ksweep" in sys.argv else []):
    us, tf = bench(mode, M, N, K, epi=epi)
    print(f"{name:16s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {tf:7.1f} TFLOP/s")
