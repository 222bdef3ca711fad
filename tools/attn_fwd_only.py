"""Launch the long-sequence attention forward (B=16, S=2048, H=12, dropout from argv) a few times (for rocprofv3 --pmc)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H = 16, 2048, 12
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
d = H * 64; T = B * S
qkv = torch.randn(T, 3 * d, device="cuda").to(torch.bfloat16); out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
dout = torch.randn(T, d, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
lse = torch.empty(B * H * S, device="cuda"); delta = torch.empty(B * H * S, device="cuda")
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
lo = torch.zeros(B, S, dtype=torch.int32, device="cuda"); hi_ = torch.full((B, S), S - 1, dtype=torch.int32, device="cuda")
for _ in range(5):
    if os.environ.get("RANGES"): L.check(lib.gget_op_attn_fwd_ranges(P(qkv), P(lo), P(hi_), P(out), P(lse), B, S, H, 0, p, 7, st))
    else: L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, 0, None, None, None, p, 7, st))
    if which == "bwd":
        L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None, None, p, 7, st))
torch.cuda.synchronize()
