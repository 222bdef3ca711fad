"""Launch the long-sequence attention backward a few times (for rocprofv3 --kernel-trace / --pmc). env P = dropout."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H = int(os.environ.get("B", 16)), int(os.environ.get("S", 2048)), 12
p = float(os.environ.get("P", 0.0))
d = H * 64; T = B * S
qkv = torch.randn(T, 3 * d, device="cuda").to(torch.bfloat16); out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
dout = torch.randn(T, d, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
lse = torch.empty(B * H * S, device="cuda"); delta = torch.empty(B * H * S, device="cuda")
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, 0, None, None, None, p, 7, st))
for _ in range(5):
    L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None, None, p, 7, st))
torch.cuda.synchronize()
