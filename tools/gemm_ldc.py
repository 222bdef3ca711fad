"""Does the row pitch of C matter (L2 channel striding)?  gate|up and dh GEMMs with padded leading dimensions."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(mode, M, N, K, ldc, lda=None, iters=20):
    if mode == L.GEMM_NT: A = torch.randn(M, lda or K, device="cuda"); B = torch.randn(N, K, device="cuda"); la, lb = lda or K, K
    else: A = torch.randn(M, lda or K, device="cuda"); B = torch.randn(K, N, device="cuda"); la, lb = lda or K, N
    A = A.to(torch.bfloat16); B = (B * 0.05).to(torch.bfloat16)
    Cm = torch.empty(M, ldc, dtype=torch.bfloat16, device="cuda")
    args = (mode, 0, P(A), P(B), P(Cm), None, M, N, K, la, lb, ldc, 1, st)
    for _ in range(3): L.check(lib.gget_op_gemm(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.check(lib.gget_op_gemm(*args))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
T, d, ff = 8192, 768, 3072
for rep in range(2):
    for pad in (0, 64, 128, 192, 320):
        print(f"gu  ldc={2*ff+pad:5d}: {bench(L.GEMM_NT, T, 2*ff, d, 2*ff+pad):7.1f} us   dh ldc={ff+pad:5d}: {bench(L.GEMM_NN, T, ff, d, ff+pad):7.1f} us   "
              f"qkv ldc={3*d+pad:5d}: {bench(L.GEMM_NT, T, 3*d, d, 3*d+pad):7.1f} us   o ldc={d+pad:5d}: {bench(L.GEMM_NT, T, d, d, d+pad):7.1f} us")
