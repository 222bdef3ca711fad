"""A/B of the RMSNorm backward at the C1 shape (HIP events, operands rotated over several copies so that nothing is served from the caches of
the previous call):

  RMSNorm backward   T = 5696 x 768:  gget_debug_set(13, 0) = 4-wave blocks, 4 rows per wave | (13, 1) = one 16-wave block per CU

prints us per launch and checks that the variants agree (dx bit-equal, dw to fp32 rounding).  (The cross-entropy launch's per-block loss
partials - gget_debug_set(14, .) - need the engine's workspace: measured in the step, tools/step_ab.py "base:" "old:13=0,14=0".)"""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr())
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
NCOPY = 6


def timed(f, n=60):
    for i in range(6):
        f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def norm(T, d=768):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: [torch.randn(T, d, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NCOPY)]
    x, dy, dres = mk(), mk(), mk()
    w = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).to(torch.bfloat16)
    rstd = [torch.rsqrt((xi.float() ** 2).mean(-1) + 1e-6) for xi in x]
    dx = [torch.empty(T, d, dtype=torch.bfloat16, device="cuda") for _ in range(NCOPY)]
    dw = torch.zeros(d, device="cuda")
    out = {}
    for v in (0, 1, 0, 1):
        lib.gget_debug_set(13, v)
        f = lambda i: L.check(lib.gget_op_rmsnorm_bwd(P(dy[i % NCOPY]), P(x[i % NCOPY]), P(w), P(rstd[i % NCOPY]), P(dres[i % NCOPY]),
                                                      P(dx[i % NCOPY]), P(dw), T, d, ST()))
        us = timed(f)
        dw.zero_()
        f(0)
        torch.cuda.synchronize()
        out.setdefault(v, []).append(us)
        out[("dx", v)] = dx[0].clone()
        out[("dw", v)] = dw.clone()
    byt = 4 * T * d * 2
    print(f"rmsnorm_bwd T={T}: 4-wave blocks {out[0]} us | 16-wave blocks {out[1]} us  ({byt / 1e6:.1f} MB: {byt / min(out[1]) / 1e6:.2f} TB/s)")
    print("   dx bit-equal:", bool(torch.equal(out[("dx", 0)], out[("dx", 1)])),
          " dw max rel diff:", float(((out[("dw", 0)] - out[("dw", 1)]).abs().max() / out[("dw", 0)].abs().max())))
    lib.gget_debug_set(13, 1)


if __name__ == "__main__":
    for T in (5696, 5504, 8192, 2048):
        norm(T)
