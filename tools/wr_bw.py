import torch, time
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (32, 100, 400, 2000):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    s = t(lambda: x.zero_()); print(f"fill {mb} MB: {mb/1024/s/1e3*1.048576:.2f} TB/s ({s*1e6:.1f} us)")
    s = t(lambda: y.copy_(x)); print(f"copy {mb} MB: r+w {2*mb/1024/s/1e3*1.048576:.2f} TB/s ({s*1e6:.1f} us)")
    s = t(lambda: x.sum()); print(f"read {mb} MB: {mb/1024/s/1e3*1.048576:.2f} TB/s ({s*1e6:.1f} us)")
