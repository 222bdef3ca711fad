"""Time the gate|up + GEGLU launch, A/B over gget_debug_set(1, bits) (bit 2 (4) = no 192-row tiles).  GGET_T = rows list."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
d, ff = 768, 3072
for T in [int(x) for x in os.environ.get("GGET_T", "5696,8192").split(",")]:
    x = torch.randn(T, d, device="cuda").to(torch.bfloat16); w = (torch.randn(2 * ff, d, device="cuda") * 0.02).to(torch.bfloat16)
    gu = torch.empty(T, 2 * ff, dtype=torch.bfloat16, device="cuda"); h = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
    flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for rnd in range(3):
        for bits in [int(v) for v in os.environ.get("GGET_BITS", "0,4").split(",")]:
            L.check(lib.gget_debug_set(1, bits))
            ts = []
            for it in range(12):
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.gget_op_gateup_geglu(P(x), P(w), P(gu), P(h), T, d, ff, st))
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            print(f"T={T} variant={bits:2d}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  ({2.0*T*2*ff*d/ts[len(ts)//2]/1e6:.0f} TF)", flush=True)
    L.check(lib.gget_debug_set(1, 0))
