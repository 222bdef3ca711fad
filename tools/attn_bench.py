"""Attention fwd/bwd micro-benchmark through the C ABI (HIP events): (B,S,H) list, dropout 0 / 0.1."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib"); lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, S, H) in [(256, 32, 12), (256, 256, 12), (32, 1024, 12), (16, 2048, 12)]:
    d = H * 64; T = B * S
    qkv = torch.randn(T, 3 * d, device="cuda").to(torch.bfloat16); out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
    dout = torch.randn(T, d, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
    lse = torch.empty(B * H * S, device="cuda"); delta = torch.empty(B * H * S, device="cuda")
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    fl = 4.0 * B * H * S * S * 64
    lo = torch.zeros(B, S, dtype=torch.int32, device="cuda"); hi = torch.full((B, S), S - 1, dtype=torch.int32, device="cuda")
    for p in (0.0, 0.1):
        if os.environ.get("RANGES"):    # the packed-row entry points with ranges that cover every key (same work)
            f = t(lambda: L.check(lib.gget_op_attn_fwd_ranges(P(qkv), P(lo), P(hi), P(out), P(lse), B, S, H, 0, p, 7, st)))
            b = t(lambda: L.check(lib.gget_op_attn_bwd_ranges(P(qkv), P(out), P(dout), P(lse), P(lo), P(hi), P(dqkv), P(delta), B, S, H, 0, p, 7, st)))
        else:
            f = t(lambda: L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, 0, None, None, None, p, 7, st)))
            b = t(lambda: L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None, None, p, 7, st)))
        if S >= 256 and not os.environ.get("RANGES"):   # the one-pass backward against the two-kernel form, alternated (same process, same clocks)
            acc = torch.empty((S + 255) // 256, T, d, dtype=torch.bfloat16, device="cuda")
            two = lambda: L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None, None, p, 7, st))
            one = lambda: L.check(lib.gget_op_attn_bwd_fused(P(qkv), P(out), P(dout), P(lse), P(lens), None, None, P(dqkv), P(delta), P(acc), B, S, H, 0, p, 7, st))
            r2, r1 = [], []
            for _ in range(5):
                r2.append(t(two, 5)); r1.append(t(one, 5))
            print(f"B={B} S={S} H={H} p={p}: bwd two-kernel " + " ".join(f"{x:.0f}" for x in r2) + "  | one-pass " + " ".join(f"{x:.0f}" for x in r1) +
                  f"  | medians {sorted(r2)[2]:.1f} / {sorted(r1)[2]:.1f} us ({2.5*fl/sorted(r1)[2]/1e6:.0f} TF)")
        print(f"B={B} S={S} H={H} p={p}: fwd {f:8.1f} us ({fl/f/1e6:6.1f} TF)  bwd {b:8.1f} us ({2.5*fl/b/1e6:6.1f} TF)")
