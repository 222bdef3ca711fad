#!/usr/bin/env python3
"""Do the epilogue store bursts of the K = 768 GEMMs overlap with MFMA work when two launches share the chip?  The fused gate|up +
GEGLU launch and the dh + GEGLU' launch of the C1 step (T rows) once as ONE launch over all CUs, and as TWO launches of T/2 rows each on
two streams (run with GGET_GEMM_NUM_CU=128 so that each takes half the CUs).  Prints us per T rows of work."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
T, d, ff = int(os.environ.get("T", "5760")), 768, 3072
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
x, wgu, wdn = bf(T, d), bf(2 * ff, d, sc=0.02), bf(d, ff, sc=0.02)
gu = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda"); h = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
dy = bf(T, d); dgu = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
two = os.environ.get("GGET_GEMM_NUM_CU") is not None
H = T // 2

def gateup(rows, off, st):
    L.check(lib.gget_op_gateup_geglu(C.c_void_p(x.data_ptr() + off * d * 2), P(wgu), C.c_void_p(gu.data_ptr() + off * 2 * ff * 2),
                                     C.c_void_p(h.data_ptr() + off * ff * 2), rows, d, ff, C.c_void_p(st.cuda_stream)))

def dh(rows, off, st):
    L.check(lib.gget_op_down_dgrad_geglu(C.c_void_p(dy.data_ptr() + off * d * 2), P(wdn), C.c_void_p(gu.data_ptr() + off * 2 * ff * 2),
                                         C.c_void_p(dgu.data_ptr() + off * 2 * ff * 2), None, rows, d, ff, C.c_void_p(st.cuda_stream)))

def run(fn, n=40):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if two:
            fn(H, 0, s1); fn(T - H, H, s2)
        else:
            fn(T, 0, s1)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, fn in (("gate|up + GEGLU", gateup), ("dh + GEGLU'", dh)):
    run(fn, 5)
    print(f"T={T} {'two half launches on two streams, 128 CUs each' if two else 'one launch, all CUs'}: {name} {run(fn):.1f} us", flush=True)
