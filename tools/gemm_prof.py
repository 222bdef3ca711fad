"""In-kernel s_memtime stamps of the persistent GEMM K-loop (GGET_GEMM_ABLATE=128): per K-step cycles spent in
wait / barrier / DMA issue / ds_read+MFMA issue for wave 0 and wave 5 of one block."""
import ctypes as C, importlib, os, sys, torch
os.environ["GGET_GEMM_ABLATE"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(name, mode, M, N, K):
    if mode == L.GEMM_NT: A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); lda, ldb = K, K
    elif mode == L.GEMM_NN: A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = K, N
    else: A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = M, N
    A = A.to(torch.bfloat16); B = (B * .05).to(torch.bfloat16); Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    dbg = torch.zeros(128 * 6, dtype=torch.int64, device="cuda")
    for _ in range(3): L.check(lib.gget_op_gemm(mode, 0, P(A), P(B), P(Cm), P(dbg), M, N, K, lda, ldb, N, 1, st))
    torch.cuda.synchronize()
    d = dbg.cpu().view(2, 64, 6)
    print(name)
    for w in range(2):
        t = d[w]
        n = int((t[:, 0] != 0).sum())
        print(f"  wave {'0' if w == 0 else '4'}: steps {n}")
        for k in range(min(n, 26)):
            prev_end = int(t[k - 1, 5]) if k else int(t[k, 0])
            print(f"   k={k:2d} t0 {int(t[k,0]-d[0,0,0]):7d} gap {int(t[k,0])-prev_end:5d} dma {int(t[k,1]-t[k,0]):5d} ds_read+wait {int(t[k,2]-t[k,1]):5d} barrier1 {int(t[k,3]-t[k,2]):5d} mfma {int(t[k,4]-t[k,3]):5d} tail+barrier2 {int(t[k,5]-t[k,4]):5d}  step {int(t[k,5])-prev_end:5d}")
T, d, ff = 8192, 768, 3072
os.environ.setdefault("GGET_GEMM_NO_256", "1")
run("NT dh-like 256x128x64 (M=8192,N=3072,K=768)", L.GEMM_NT, T, ff, d)
