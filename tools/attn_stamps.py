"""Phase times of the dK/dV attention kernel from its s_memtime stamps (measurement build, tools/attn_stamps.sh)."""
import ctypes as C, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = C.CDLL(os.environ["GGET_LIB_PATH"])
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H = 16, 2048, 12
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
d = H * 64; T = B * S
qkv = torch.randn(T, 3 * d, device="cuda").to(torch.bfloat16); out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
dout = torch.randn(T, d, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
lse = torch.empty(B * H * S, device="cuda"); delta = torch.empty(B * H * S, device="cuda")
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
lib.gget_op_attn_fwd.restype = C.c_int; lib.gget_op_attn_bwd.restype = C.c_int
fa = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_float, C.c_uint32, C.c_void_p]
lib.gget_op_attn_fwd.argtypes = fa
lib.gget_op_attn_bwd.argtypes = [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_float, C.c_uint32, C.c_void_p]
assert lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, 0, None, None, None, p, 7, st) == 0
for _ in range(3):
    assert lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None, None, p, 7, st) == 0
buf = np.zeros((8, 40, 16), np.uint64)
lib.gget_debug_attn_stamps.argtypes = [C.c_void_p]
assert lib.gget_debug_attn_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.astype(np.int64)
names = ["wait+barrier", "DMA issue", "j0: frags + S/dP MFMA issue", "j0: exp (incl. MFMA results)", "j0: dS", "j0: pack", "j0: dV/dK MFMA issue",
         "j1: frags + S/dP MFMA issue", "j1: exp", "j1: dS", "j1: pack", "j1: dV/dK MFMA issue"]
print(f"dK/dV kernel, block (0,0,0), B={B} S={S} H={H} dropout {p}: cycles per phase, median over stages 4..31, per wave")
for w in range(8):
    seg = np.diff(s[w, 4:32, :13], axis=1)            # [stage, 12 phases]
    loop = s[w, 5:32, 0] - s[w, 4:31, 0]
    med = np.median(seg, axis=0)
    vmw = np.median(s[w, 4:32, 13] - s[w, 4:32, 0])
    print(f"wave {w}: stage {np.median(loop):6.0f} | (vmcnt wait {int(vmw)}) " + " | ".join(f"{int(m)}" for m in med))
print("phases: " + " | ".join(names))
