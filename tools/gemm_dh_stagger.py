"""dh + GEGLU' launch (two workgroups per CU, gemm_persist_kernel MODE 2) against the start delay of a CU's second workgroup
(gget_debug_set key 5, 100 MHz ticks): cold cache (flushed before every launch, like the step sees it) - median / min us, and the
output compared with the un-staggered launch (must be bit-equal).  GGET_T rows (default 5696), GGET_TICKS comma list."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
d, ff = 768, 3072
ticks = [int(x) for x in os.environ.get("GGET_TICKS", "0,200,400,600,800,1000,1300,1600,2000").split(",")]
for T in [int(x) for x in os.environ.get("GGET_T", "5696").split(",")]:
    dy = torch.randn(T, d, device="cuda").to(torch.bfloat16); w = (torch.randn(d, ff, device="cuda") * 0.02).to(torch.bfloat16)
    gu = torch.randn(T, 2 * ff, device="cuda").to(torch.bfloat16); dgu = torch.empty_like(gu)
    dh = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
    flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ref = None
    for rnd in range(2):
        for tk in ticks:
            L.check(lib.gget_debug_set(5, tk))
            ts = []
            for it in range(12):
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.gget_op_down_dgrad_geglu(P(dy), P(w), P(gu), P(dgu), P(dh), T, d, ff, st))
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            if ref is None:
                ref = dgu.clone()
            same = bool(torch.equal(ref, dgu))
            print(f"T={T} stagger={tk * 0.01:5.1f} us: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  bit-equal {same}", flush=True)
    L.check(lib.gget_debug_set(5, 0))
