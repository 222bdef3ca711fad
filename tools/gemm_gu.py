"""Launch the dominant kernel of the C1 step a few times (for rocprofv3 --pmc): the FFN gate|up GEMM [8192,768]x[768,6144]
with the gated-GELU product in its epilogue (GU_PLAIN=1: the plain GEMM of round 1)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, ff = int(__import__("os").environ.get("GGET_T", "8192")), 768, 3072   # GGET_T: rows of the launch
A = torch.randn(T, d, device="cuda").to(torch.bfloat16); B = (torch.randn(2 * ff, d, device="cuda") * 0.02).to(torch.bfloat16)
Cm = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda"); H = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    if os.environ.get("GU_PLAIN"): L.check(lib.gget_op_gemm(L.GEMM_NT, 0, P(A), P(B), P(Cm), None, T, 2 * ff, d, d, d, 2 * ff, 1, st))
    else: L.check(lib.gget_op_gateup_geglu(P(A), P(B), P(Cm), P(H), T, d, ff, st))
torch.cuda.synchronize()
