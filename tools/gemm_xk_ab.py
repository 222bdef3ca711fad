#!/usr/bin/env python3
"""A/B of the cross-block K-half form of the in-block K-split GEMM (gemm_ks_kernel<XK>, g_gemm_variant bit 11) on the N = d launches of the
C1 step: dxn2 (NN, K = 6144), dxn1 (NN, K = 2304), down + residual (NT, K = 3072).  tools/ only.
The variant is NOT in the tree: apply profiles/r05_cross_block_k_halves_experiment.diff first (without it both columns time the shipped kernel).
usage: gemm_xk_ab.py [T]"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.gget_op_gemm_streamk_bytes.restype = C.c_uint64
ws = torch.zeros(int(lib.gget_op_gemm_streamk_bytes()), dtype=torch.uint8, device="cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5696
d, ff = 768, 3072
COPIES, ITERS, ROUNDS = 4, 10, 5


def run(name, mode, epi, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    ops = []
    for _ in range(COPIES):
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        B = (torch.randn((N, K) if mode == L.GEMM_NT else (K, N), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        R = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi else None
        ops.append((A, B, R, torch.empty(M, N, dtype=torch.bfloat16, device="cuda")))

    def launch(i):
        A, B, R, out = ops[i % COPIES]
        lda, ldb = (K, K) if mode == L.GEMM_NT else (K, N)
        L.check(lib.gget_op_gemm_streamk(mode, epi, P(A), P(B), P(out), P(R), M, N, K, lda, ldb, N, P(ws), st))
        return out
    res, times = {}, {}
    for var in (0, 2048):
        L.check(lib.gget_debug_set(1, var))
        res[var] = launch(0).float().clone()
    A, B, R, _ = ops[0]
    want = A.float() @ (B.float().t() if mode == L.GEMM_NT else B.float()) + (R.float() if R is not None else 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(ROUNDS + 1):
        for var in (0, 2048):
            L.check(lib.gget_debug_set(1, var))
            e0.record()
            for i in range(ITERS):
                launch(i)
            e1.record(); torch.cuda.synchronize()
            if r: times.setdefault(var, []).append(e0.elapsed_time(e1) / ITERS * 1e3)
    L.check(lib.gget_debug_set(1, 0))
    rel = lambda x: float((x - want).norm() / want.norm())
    print(f"{name:10s} {M}x{N}x{K}: shipped {statistics.median(times[0]):7.1f} us  K halves on two blocks {statistics.median(times[2048]):7.1f} us | "
          f"rel-L2 vs fp32 {rel(res[0]):.2e} / {rel(res[2048]):.2e}  max |diff| between the two {float((res[0] - res[2048]).abs().max()):.3g}", flush=True)


run("dxn2 NN", L.GEMM_NN, 0, T, d, 2 * ff)
run("dxn1 NN", L.GEMM_NN, 0, T, d, 3 * d)
run("down NT+R", L.GEMM_NT, 1, T, d, ff)
