import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, ff = 8192, 768, 3072
def run(mode, M, N, K):
    if mode == L.GEMM_NT: A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); lda, ldb = K, K
    elif mode == L.GEMM_NN: A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = K, N
    else: A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = M, N
    A = A.to(torch.bfloat16); B = B.to(torch.bfloat16); Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): L.check(lib.gget_op_gemm(mode, 0, P(A), P(B), P(Cm), None, M, N, K, lda, ldb, N, 1, st))
    torch.cuda.synchronize()
run(L.GEMM_NT, T, 2 * ff, d); run(L.GEMM_NN, T, ff, d); run(L.GEMM_TN, 2 * ff, d, T)
