#!/usr/bin/env python3
"""S <= 32 decoder-layer front half at the headline shape (B 256, S 32, H 12, PCQM4M-v2 length distribution, var-len rows): the fused
launch (attention + o projection + residual + RMSNorm, one workgroup per sample) against the three launches it replaces, HIP-event timed,
alternated in one process."""
import ctypes as C, importlib, os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("graph-gpt_amd._lib")
synth = importlib.import_module("graph-gpt_amd.synth")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, S, H, p = int(os.environ.get("B", "256")), int(os.environ.get("S", "32")), 12, 0.1      # S=40 / 56: every sample by its own row count (round 6)
d = H * 64
lens = torch.from_numpy(synth.make_pretrain_batch(B=B, S=S, F=13, V=756, seed=1234)["attention_mask"].sum(1).astype(np.int32))
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = torch.cumsum(lens, 0)
T = (int(cu[-1]) + 63) // 64 * 64
print(f"B {B} S {S} H {H}: {int(cu[-1])} real tokens -> {T} rows")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
NB = 6      # operand sets in rotation (a layer's tensors are cold in L2 when it runs)
R = B * S    # rows allocated (the padded-grid attention op reads B * S rows; the var-len launches use the first T)
sets = [dict(qkv=mk(R, 3 * d), wo=mk(d, d, sc=0.03), x=mk(R, d), nw=mk(d), attn=torch.empty(R, d, dtype=torch.bfloat16, device="cuda"),
             xmid=torch.empty(R, d, dtype=torch.bfloat16, device="cuda"), xn=torch.empty(R, d, dtype=torch.bfloat16, device="cuda"),
             lse=torch.empty(B * H * S, dtype=torch.float32, device="cuda"), rstd=torch.empty(R, dtype=torch.float32, device="cuda")) for _ in range(NB)]
lens_d, rb = lens.cuda(), cu[:B].contiguous().cuda()
taken = C.c_int32(0)
for o in sets:     # fragment-major copies of the o weight (the engine rebuilds them once per forward for all layers)
    o["wo_f"], o["wo_b"] = torch.empty_like(o["wo"]), torch.empty_like(o["wo"])
    L.check(lib.gget_op_pack_wo(P(o["wo"]), 0, P(o["wo_f"]), P(o["wo_b"]), d, 1, st))
    o["dxn"], o["dres"], o["dxmid"] = mk(R, d, sc=0.5), mk(R, d, sc=0.5), torch.empty(R, d, dtype=torch.bfloat16, device="cuda")
    o["dattn"], o["dqkv"] = torch.empty(R, d, dtype=torch.bfloat16, device="cuda"), torch.empty(R, 3 * d, dtype=torch.bfloat16, device="cuda")
    o["dattn_long"] = torch.empty(R, d, dtype=torch.bfloat16, device="cuda")
    o["dw"] = torch.zeros(16 * 1024, dtype=torch.float32, device="cuda")
    o["delta"] = torch.empty(B * H * S, dtype=torch.float32, device="cuda")


def fused(o):
    L.check(lib.gget_op_attn_oproj_fwd(P(o["qkv"]), P(lens_d), P(rb), P(o["attn"]), P(o["lse"]), P(o["wo_f"]), P(o["x"]), P(o["xmid"]), P(o["nw"]),
                                       P(o["xn"]), P(o["rstd"]), B, S, H, 0, 1e-6, p, 7, st, C.byref(taken)))


def attn_only(o):     # (the var-len entry: one-tile kernel + the launch of the 33 .. 64-row samples when S > 32)
    L.check(lib.gget_op_attn_fwd_varlen(P(o["qkv"]), P(lens_d), P(rb), P(o["attn"]), P(o["lse"]), B, S, H, 0, None, None, None, 1, p, 7, st))


def gemm(o):
    L.check(lib.gget_op_gemm(L.GEMM_NT, 1, P(o["attn"]), P(o["wo"]), P(o["xmid"]), P(o["x"]), T, d, d, d, d, d, 1, st))


def norm(o):
    L.check(lib.gget_op_rmsnorm_fwd(P(o["xmid"]), P(o["nw"]), P(o["xn"]), P(o["rstd"]), T, d, 1e-6, st))


def separate(o):
    attn_only(o); gemm(o); norm(o)


def fused_bwd(o):
    L.check(lib.gget_op_attn_oproj_bwd(P(o["dxn"]), P(o["xmid"]), P(o["nw"]), P(o["rstd"]), P(o["dres"]), P(o["dxmid"]), P(o["dw"]), 16, 1024, P(o["wo_b"]),
                                       P(o["qkv"]), P(o["lse"]), P(lens_d), P(rb), P(o["dqkv"]), B, S, H, 0, None, None, None, p, 7, T, st, C.byref(taken),
                                       P(o["dattn_long"])))


def norm_bwd(o):
    L.check(lib.gget_op_rmsnorm_bwd(P(o["dxn"]), P(o["xmid"]), P(o["nw"]), P(o["rstd"]), P(o["dres"]), P(o["dxmid"]), P(o["dw"]), T, d, st))


def gemm_bwd(o):
    L.check(lib.gget_op_gemm(L.GEMM_NN, 0, P(o["dxmid"]), P(o["wo"]), P(o["dattn"]), None, T, d, d, d, d, d, 1, st))


def attn_bwd(o):
    L.check(lib.gget_op_attn_bwd_varlen(P(o["qkv"]), P(o["attn"]), P(o["dattn"]), P(o["lse"]), P(lens_d), P(rb), P(o["dqkv"]), P(o["delta"]), B, S, H, 0,
                                        None, None, None, 1, p, 7, st))


def separate_bwd(o):
    norm_bwd(o); gemm_bwd(o); attn_bwd(o)


def pack_all(o):     # 12 layers' worth
    for _ in range(1):
        L.check(lib.gget_op_pack_wo(P(big_w), d * d, P(big_f), P(big_b), d, 12, st))


big_w = mk(12 * d * d); big_f = torch.empty_like(big_w); big_b = torch.empty_like(big_w)


def timeit(fn, iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3): fn(sets[i % NB])
    e0.record()
    for i in range(iters): fn(sets[i % NB])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


fused(sets[0]); fwd_taken = taken.value == 1      # (S > 32: the forward keeps its three launches)
for o in sets: separate(o)      # (lse, rstd, x_mid of every set: the backward's inputs)
fused_bwd(sets[0]); assert taken.value == 1
cases = ((("fused", fused),) if fwd_taken else ()) + (("separate", separate), ("attn", attn_only), ("gemm", gemm), ("norm", norm), ("fused_bwd", fused_bwd),
         ("separate_bwd", separate_bwd), ("norm_bwd", norm_bwd), ("gemm_bwd", gemm_bwd), ("attn_bwd", attn_bwd), ("pack_12_layers", pack_all))
res = {k: [] for k, _ in cases}
for r in range(5):
    for k, fn in cases:
        res[k].append(timeit(fn))
for k, v in res.items():
    print(f"{k:9s} median {statistics.median(v):7.2f} us   min {min(v):7.2f} us")
wbytes = d * d * 2
if fwd_taken:
    print(f"weight stream per sample {wbytes / 1e6:.2f} MB; fused launch = {B * wbytes / statistics.median(res['fused']) / 1e6:.2f} TB/s of L2 -> CU traffic over all CUs")
