"""Launch the dgrad of the gate|up projection of the C1 step a few times (for rocprofv3 --pmc; tools/pmc_dxn2.sh):
dxn2 = dgu W_gu, [T,6144] x [6144,768] (NN, in-block K-split kernel).  GGET_T = rows."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, ff = int(os.environ.get("GGET_T", "5696")), 768, 3072
dgu = torch.randn(T, 2 * ff, device="cuda").to(torch.bfloat16); w = (torch.randn(2 * ff, d, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    L.check(lib.gget_op_gemm(L.GEMM_NN, 0, P(dgu), P(w), P(out), None, T, d, 2 * ff, 2 * ff, d, d, 1, st))
torch.cuda.synchronize()
