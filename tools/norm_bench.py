"""RMSNorm fwd/bwd micro-benchmark (HIP events) at the C1 shape; env GGET_RMS_ROWS = rows per wave of the bwd kernel."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d = int(os.environ.get("T", 8192)), 768
x = torch.randn(T, d, device="cuda").to(torch.bfloat16); w = torch.ones(d, device="cuda").to(torch.bfloat16)
dy = torch.randn(T, d, device="cuda").to(torch.bfloat16); dres = torch.randn(T, d, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x); dx = torch.empty_like(x); rstd = torch.empty(T, device="cuda"); dw = torch.zeros(d, device="cuda")
def t(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("fwd us", t(lambda: L.check(lib.gget_op_rmsnorm_fwd(P(x), P(w), P(y), P(rstd), T, d, 1e-6, st))))
print("bwd us", t(lambda: L.check(lib.gget_op_rmsnorm_bwd(P(dy), P(x), P(w), P(rstd), P(dres), P(dx), P(dw), T, d, st))))
