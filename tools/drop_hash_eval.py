"""Statistics of the attention-dropout hash (csrc/attention.hip drop_word): drop rates and mask correlations of the 24-bit-multiply mix
against the 32-bit multiply it replaced.  CPU only: python tools/drop_hash_eval.py"""
import numpy as np
M = np.uint64(0xffffffff)
def base(seed, bh, q, kh):
    x = (np.uint64(seed) ^ ((bh.astype(np.uint64) * np.uint64(0x9E3779B1)) & M))
    x = (x + q.astype(np.uint64) * np.uint64(0x85EBCA77) + kh.astype(np.uint64) * np.uint64(0xC2B2AE3D)) & M
    return x
def old(x):
    x = x ^ (x >> np.uint64(16)); x = (x * np.uint64(0x7FEB352D)) & M; x = x ^ (x >> np.uint64(15)); return x
def new(x, C):
    x = x ^ (x >> np.uint64(16)); x = ((x & np.uint64(0xffffff)) * np.uint64(C)) & M; x = x ^ (x >> np.uint64(15)); return x
def evaluate(f, name):
    out = []
    for seed in (1234, 0xdeadbeef, 7):
        BH, S = 24, 512
        bh, q, kh = np.meshgrid(np.arange(BH), np.arange(S), np.arange(S // 2), indexing="ij")
        w = f(base(seed, bh, q, kh))
        lo = (w & np.uint64(0xffff)).astype(np.int64); hi = (w >> np.uint64(16)).astype(np.int64)
        res = {}
        for p in (0.1, 0.3):
            t = int(p * 65536)
            dl = (lo < t).astype(np.float64); dh = (hi < t).astype(np.float64)
            full = np.stack([dl, dh], -1).reshape(BH, S, S)   # [bh, q, k]
            c = lambda a, b: float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
            res[p] = dict(rate_lo=dl.mean(), rate_hi=dh.mean(), pair=c(dl, dh), k1=c(full[:, :, :-1], full[:, :, 1:]), k2=c(full[:, :, :-2], full[:, :, 2:]),
                          q1=c(full[:, :-1], full[:, 1:]), q2=c(full[:, :-2], full[:, 2:]), h1=c(full[:-1], full[1:]),
                          diag=c(full[:, :-1, :-1], full[:, 1:, 1:]), anti=c(full[:, :-1, 1:], full[:, 1:, :-1]),
                          row_rate_std=full.mean(2).std() / np.sqrt(p * (1 - p) / S), col_rate_std=full.mean(1).std() / np.sqrt(p * (1 - p) / S))
        out.append(res)
    print(name)
    for p in (0.1, 0.3):
        keys = out[0][p].keys()
        print("  p=%.1f " % p + " ".join("%s=%s" % (k, "/".join("%.4f" % o[p][k] for o in out)) for k in keys))
evaluate(old, "old (mul 32)")
for C in (0x9E3779, 0xB5ED51, 0xD2B74D, 0xC2B2AF):
    evaluate(lambda x, C=C: new(x, C), "new mul_u24 C=%06x" % C)
