"""Can an HBM-bound elementwise kernel hide under the persistent MFMA GEMMs (two streams)?"""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
T, d, ff = 8192, 768, 3072
A = torch.randn(T, d, device="cuda").to(torch.bfloat16); B = (torch.randn(2 * ff, d, device="cuda") * 0.02).to(torch.bfloat16)
Cm = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
x = torch.randn(400_000_000, device="cuda")   # 1.6 GB fp32: x.mul_(1.0001) reads + writes 3.2 GB
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemms(n=24):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(n): L.check(lib.gget_op_gemm(L.GEMM_NT, 0, P(A), P(B), P(Cm), None, T, 2 * ff, d, d, d, 2 * ff, 1, st))
def elem(): x.mul_(1.0001)
def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    if fa:
        with torch.cuda.stream(s1): fa()
    if fb:
        with torch.cuda.stream(s2): fb()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(2): timed(gemms, elem)
print("gemm x24 alone   %.3f ms" % timed(gemms, None))
print("elementwise alone %.3f ms" % timed(None, elem))
print("both, two streams %.3f ms" % timed(gemms, elem))
