"""Launch the dh + GEGLU' GEMM of the C1 step a few times (for rocprofv3 --pmc; tools/pmc_dh.sh): dgu = GEGLU'(dy W_down, gu),
[T,768] x [768,3072] with the gated-GELU backward in the epilogue.  GGET_T = rows."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, ff = int(os.environ.get("GGET_T", "5696")), 768, 3072
dy = torch.randn(T, d, device="cuda").to(torch.bfloat16); w = (torch.randn(d, ff, device="cuda") * 0.02).to(torch.bfloat16)
gu = torch.randn(T, 2 * ff, device="cuda").to(torch.bfloat16); dgu = torch.empty_like(gu)
dh = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    L.check(lib.gget_op_down_dgrad_geglu(P(dy), P(w), P(gu), P(dgu), P(dh), T, d, ff, st))
torch.cuda.synchronize()
