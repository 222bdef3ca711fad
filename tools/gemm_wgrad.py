"""Launch the heaviest kernel of the C1 step a few times (for rocprofv3 --pmc): the grouped weight-gradient launch of one decoder
layer - dW = dY^T X for gate|up, down, q|k|v and o in ONE launch of 192x192 tiles (TN, K = T = 8192; 256 tiles = one per CU)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d, ff = int(os.environ.get("GGET_T", "8192")), 768, 3072     # GGET_T: rows = K of the launch (var-len layout: a batch's real tokens)
bf = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
dgu, dy, dqkv, xn, h, attn = bf(T, 2 * ff), bf(T, d), bf(T, 3 * d), bf(T, d), bf(T, ff), bf(T, d)
gw = [torch.empty(2 * ff, d, dtype=torch.bfloat16, device="cuda"), torch.empty(d, ff, dtype=torch.bfloat16, device="cuda"),
      torch.empty(3 * d, d, dtype=torch.bfloat16, device="cuda"), torch.empty(d, d, dtype=torch.bfloat16, device="cuda")]
probs = [(dgu, xn, gw[0], 2 * ff, d, T, 2 * ff, d, d), (dy, h, gw[1], d, ff, T, d, ff, ff),
         (dqkv, xn, gw[2], 3 * d, d, T, 3 * d, d, d), (dy, attn, gw[3], d, d, T, d, d, d)]
for _ in range(5):
    L.check(L.gemm_grouped(lib, L.GEMM_TN, probs, st))
torch.cuda.synchronize()
