#!/usr/bin/env python3
"""Interleaved A/B timing of GEMM kernel variants in ONE process (cdna guide rule 24): for every C1 shape the variants
selected through gget_debug_set(1, mask) are run in alternating rounds on the same random operands, HIP-event timed,
median and min reported, each checked once against torch (fp32 matmul of the same bf16 inputs).
usage: gemm_ab.py [mask ...]   (default masks: 0 and 15)"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
masks = [int(x) for x in sys.argv[1:] if x.lstrip("-").isdigit()] or [0, 15]
ROUNDS, ITERS = int(os.environ.get("AB_ROUNDS", "7")), int(os.environ.get("AB_ITERS", "10"))
# a cache-cold pass between timed launches: the step's GEMM operands come from HBM / Infinity Cache, not a warm L2
FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if os.environ.get("AB_FLUSH", "0") != "0" else None


def operands(mode, M, N, K, epi):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    if mode == L.GEMM_NT: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g); lda, ldb = K, K
    elif mode == L.GEMM_NN: A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g); lda, ldb = K, N
    else: A = torch.randn(K, M, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g); lda, ldb = M, N
    A = A.to(torch.bfloat16); B = (B * 0.05).to(torch.bfloat16)
    fill = os.environ.get("AB_FILL", "rand")   # operand data changes the clock the chip sustains (cdna guide rule 25)
    if fill == "zero": A.zero_(); B.zero_()
    elif fill == "const": A.fill_(0.5); B.fill_(0.25)
    R = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 1 else None
    return A, B, R, lda, ldb


def ref(mode, A, B, R):
    a, b = A.float(), B.float()
    r = a @ b.t() if mode == L.GEMM_NT else a @ b if mode == L.GEMM_NN else a.t() @ b
    return r + R.float() if R is not None else r


def run(name, mode, M, N, K, epi):
    A, B, R, lda, ldb = operands(mode, M, N, K, epi)
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    if os.environ.get("AB_LD0"):   # every operand row aliases row 0: the whole problem is L1/L2-hot (isolates the memory system)
        lda = ldb = 0
    args = (mode, epi, P(A), P(B), P(Cm), P(R), M, N, K, lda, ldb, N, 1, st)
    want = ref(mode, A, B, R)
    errs, times = {}, {m: [] for m in masks}
    for m in masks:
        L.check(lib.gget_debug_set(1, m))
        Cm.zero_()
        L.check(lib.gget_op_gemm(*args))
        errs[m] = float((Cm.float() - want).norm() / want.norm())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(ROUNDS):
        for m in masks:
            L.check(lib.gget_debug_set(1, m))
            L.check(lib.gget_op_gemm(*args))
            if FLUSH is not None:
                tot = 0.0
                for _ in range(ITERS):
                    FLUSH.add_(1)
                    e0.record(); L.check(lib.gget_op_gemm(*args)); e1.record(); torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1)
                times[m].append(tot / ITERS * 1e3)
            else:
                e0.record()
                for _ in range(ITERS): L.check(lib.gget_op_gemm(*args))
                e1.record(); torch.cuda.synchronize()
                times[m].append(e0.elapsed_time(e1) / ITERS * 1e3)
    fl = 2.0 * M * N * K
    cells = []
    for m in masks:
        med, mn = statistics.median(times[m]), min(times[m])
        cells.append(f"pp={m:<2d} {med:7.1f} us (min {mn:6.1f}) {fl / med / 1e6:6.0f} TF err {errs[m]:.1e}")
    print(f"{name:12s} {M:5d}x{N:5d}x{K:5d} | " + " | ".join(cells), flush=True)
    bad = [m for m in masks if not errs[m] < 5e-3 and fill_ok]
    return bad


fill_ok = os.environ.get("AB_FILL", "rand") == "rand" and not os.environ.get("GGET_GEMM_ABLATE") and not os.environ.get("AB_LD0")
T, d, ff = 8192, 768, 3072
shapes = [("NT qkv", L.GEMM_NT, T, 3 * d, d, 0), ("NT gu", L.GEMM_NT, T, 2 * ff, d, 0), ("NT o+res", L.GEMM_NT, T, d, d, 1),
          ("NT down+res", L.GEMM_NT, T, d, ff, 1), ("NN dh", L.GEMM_NN, T, ff, d, 0), ("NN dxn2", L.GEMM_NN, T, d, 2 * ff, 0),
          ("NN dattn", L.GEMM_NN, T, d, d, 0), ("NN dxn1", L.GEMM_NN, T, d, 3 * d, 0), ("TN dWgu", L.GEMM_TN, 2 * ff, d, T, 0),
          ("TN dWdown", L.GEMM_TN, d, ff, T, 0)]
if "--big" in sys.argv:
    shapes += [("NT 4k", L.GEMM_NT, 4096, 4096, 4096, 0), ("NT 8k", L.GEMM_NT, 8192, 8192, 8192, 0)]
print(f"rounds {ROUNDS} x iters {ITERS}, flush {'on' if FLUSH is not None else 'off'}")
bad = []
for s in shapes:
    bad += [(s[0], m) for m in run(*s)]
L.check(lib.gget_debug_set(1, 0))
if bad:
    print("WRONG RESULTS:", bad)
    sys.exit(1)
