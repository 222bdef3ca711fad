#!/usr/bin/env python3
"""Vendor-library yardstick for the layer-stack GEMM shapes.  tools/ only: nothing here is linked
into or imported by the product path.  For every GEMM shape of one decoder layer (forward NT, dgrad NN, wgrad TN) at the
row counts of C1 (T = 5696) and C3 (T = 41088) it times torch.matmul (hipBLASLt / rocBLAS behind ATen) and this repo's
kernel through gget_op_gemm on the same bf16 operands, HIP-event timed, alternated in one process, operands rotated over
several copies so that neither side runs on an L2-warm problem.  The vendor figure has no fused epilogue (no RoPE, no
GEGLU, no residual): it is the plain-GEMM yardstick the fused launches are compared with, not a like-for-like replacement.
usage: gemm_yardstick.py [T ...]"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ROUNDS, ITERS, COPIES = int(os.environ.get("AB_ROUNDS", "5")), int(os.environ.get("AB_ITERS", "8")), int(os.environ.get("AB_COPIES", "4"))
PEAK = 2.5e15


def operands(mode, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if mode == L.GEMM_NT: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g)
    elif mode == L.GEMM_NN: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g)
    else: A = torch.randn(K, M, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g)
    return A.to(torch.bfloat16), (B * 0.05).to(torch.bfloat16)


def vendor(mode, A, B, out):
    if mode == L.GEMM_NT: torch.matmul(A, B.t(), out=out)
    elif mode == L.GEMM_NN: torch.matmul(A, B, out=out)
    else: torch.matmul(A.t(), B, out=out)


def ours(mode, A, B, out, M, N, K):
    lda, ldb = (K, K) if mode == L.GEMM_NT else (K, N) if mode == L.GEMM_NN else (M, N)
    L.check(lib.gget_op_gemm(mode, 0, P(A), P(B), P(out), None, M, N, K, lda, ldb, N, 1, st))


def run(name, mode, M, N, K):
    ops = [operands(mode, M, N, K, 17 * i + M + N + K) for i in range(COPIES)]
    outs = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(COPIES)]
    vendor(mode, *ops[0], outs[0]); want = outs[0].float().clone()
    outs[1].zero_(); ours(mode, *ops[0], outs[1], M, N, K)
    err = float((outs[1].float() - want).norm() / want.norm())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tv, to = [], []
    for r in range(ROUNDS + 1):
        for which, acc in (("v", tv), ("o", to)):
            e0.record()
            for i in range(ITERS):
                A, B = ops[i % COPIES]
                if which == "v": vendor(mode, A, B, outs[i % COPIES])
                else: ours(mode, A, B, outs[i % COPIES], M, N, K)
            e1.record(); torch.cuda.synchronize()
            if r: acc.append(e0.elapsed_time(e1) / ITERS * 1e3)
    fl = 2.0 * M * N * K
    mv, mo = statistics.median(tv), statistics.median(to)
    print(f"{name:12s} {M:6d}x{N:5d}x{K:6d} | vendor {mv:8.1f} us {fl / mv / 1e6:6.0f} TF {fl / mv * 1e6 / PEAK:5.3f} | "
          f"gget {mo:8.1f} us {fl / mo / 1e6:6.0f} TF {fl / mo * 1e6 / PEAK:5.3f} | gget/vendor time {mo / mv:5.2f} | rel diff {err:.1e}", flush=True)
    return mv, mo


d, ff = 768, 3072
print(f"torch {torch.__version__}  blas {torch.backends.cuda.preferred_blas_library()}  device {torch.cuda.get_device_name(0)}")
print(f"rounds {ROUNDS} x iters {ITERS}, {COPIES} operand copies in rotation; peak for the fraction = 2.5 PFLOP/s dense bf16")
for T in [int(x) for x in sys.argv[1:]] or [5696, 41088]:
    shapes = [("NT qkv", L.GEMM_NT, T, 3 * d, d), ("NT o", L.GEMM_NT, T, d, d), ("NT gate|up", L.GEMM_NT, T, 2 * ff, d),
              ("NT down", L.GEMM_NT, T, d, ff), ("NN dh", L.GEMM_NN, T, ff, d), ("NN dxn2", L.GEMM_NN, T, d, 2 * ff),
              ("NN dattn", L.GEMM_NN, T, d, d), ("NN dxn1", L.GEMM_NN, T, d, 3 * d), ("TN dWgu", L.GEMM_TN, 2 * ff, d, T),
              ("TN dWdown", L.GEMM_TN, d, ff, T), ("TN dWqkv", L.GEMM_TN, 3 * d, d, T), ("TN dWo", L.GEMM_TN, d, d, T)]
    print(f"--- T = {T}")
    sv = so = 0.0
    for s in shapes:
        v, o = run(*s)
        sv += v; so += o
    lf = 2.0 * T * d * (3 * d + d + 2 * ff + ff) * 3
    print(f"layer total (12 plain GEMMs): vendor {sv:8.1f} us = {lf / sv * 1e6 / PEAK:5.3f} of peak | gget {so:8.1f} us = {lf / so * 1e6 / PEAK:5.3f} of peak")
