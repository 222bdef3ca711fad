"""Kernel-level parity: every HIP op, called through the C ABI (include/gget.h, gget_op_*), against a
plain fp32 PyTorch statement of the same arithmetic on the same bf16 inputs (tolerances written per test)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from _util import rel_l2

pytestmark = pytest.mark.gpu

L = importlib.import_module("graph-gpt_amd._lib")


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ST():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def lib():
    return L.load()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("mode", [L.GEMM_NT, L.GEMM_NN, L.GEMM_TN])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (200, 136, 72), (1000, 756, 128), (64, 2304, 768),
                                   (333, 768, 3072), (4096, 2304, 128), (2048, 768, 512),
                                   (4096, 4096, 192), (8192, 2304, 64)])  # the last two take the 256x256x32 tile
def test_gemm_modes(lib, mode, M, N, K):
    # asymmetric operands (catches transposed outputs); sizes include ragged M/N/K tails
    if mode == L.GEMM_NT:
        A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
        ref = A.float() @ B.float().T
        lda, ldb = K, K
    elif mode == L.GEMM_NN:
        if N % 8:
            pytest.skip("N-contiguous operand needs N % 8 == 0")
        A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
        ref = A.float() @ B.float()
        lda, ldb = K, N
    else:
        if N % 8 or M % 8:
            pytest.skip("M/N-contiguous operands need M,N % 8 == 0")
        A, B = rnd(K, M, seed=1), rnd(K, N, seed=2)
        ref = A.float().T @ B.float()
        lda, ldb = M, N
    ldc = (N + 7) // 8 * 8
    Cm = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(mode, L.EPI_NONE, P(A), P(B), P(Cm), None, M, N, K, lda, ldb, ldc, 1, ST()))
    torch.cuda.synchronize()
    got = Cm[:, :N].float()
    err = rel_l2(got.cpu().numpy(), ref.cpu().numpy())
    assert err < 4e-3, f"mode {mode} {M}x{N}x{K}: rel-L2 {err}"  # bf16 output rounding ~ 2^-9
    if ldc > N:
        assert torch.all(Cm[:, N:] == 7.0), "wrote outside the N bound"


def test_gemm_residual_and_atomic(lib):
    M, N, K = 300, 256, 320
    A, B, R = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(M, N, seed=5)
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(L.GEMM_NT, L.EPI_RESIDUAL, P(A), P(B), P(Cm), P(R), M, N, K, K, K, N, 1, ST()))
    ref = A.float() @ B.float().T + R.float()
    assert rel_l2(Cm.float().cpu().numpy(), ref.cpu().numpy()) < 4e-3
    Cf = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_gemm(L.GEMM_NT, L.EPI_ATOMIC_F32, P(A), P(B), P(Cf), None, M, N, K, K, K, N, 3, ST()))
    ref2 = A.float() @ B.float().T
    assert rel_l2(Cf.cpu().numpy(), ref2.cpu().numpy()) < 1e-5  # fp32 accumulate, no output rounding
    # split-K into per-slice fp32 slabs (what the wgrad of the attention projections and of lm_head use)
    Cs = torch.zeros(4, M, N, dtype=torch.float32, device="cuda")  # slices past the end of K write nothing
    L.check(lib.gget_op_gemm(L.GEMM_NT, L.EPI_SLAB_F32, P(A), P(B), P(Cs), None, M, N, K, K, K, N, 4, ST()))
    assert rel_l2(Cs.sum(0).cpu().numpy(), ref2.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("mode", [L.GEMM_NT, L.GEMM_NN])
@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (1000, 768, 256), (40000, 384, 128), (41472, 768, 768), (41413, 768, 2304), (33000, 1536, 320)])
def test_gemm_tile_128x192(lib, mode, M, N, K):
    # N = d outputs take the 128x192 persistent tile (N % 192 == 0); residual epilogue, ragged M, several tiles per block.  The last three
    # shapes (ogbl-ppa-sized row counts) take 256x256 tiles by the "within 10 % of the default tile's rounds x area" rule, one with a ragged
    # last row tile, one with K no multiple of 64 x ring depth
    A, R = rnd(M, K, seed=11), rnd(M, N, seed=13)
    B = rnd(N, K, seed=12) if mode == L.GEMM_NT else rnd(K, N, seed=12)
    ref = A.float() @ (B.float().T if mode == L.GEMM_NT else B.float()) + R.float()
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(mode, L.EPI_RESIDUAL, P(A), P(B), P(Cm), P(R), M, N, K, K, K if mode == L.GEMM_NT else N, N, 1, ST()))
    assert rel_l2(Cm.float().cpu().numpy(), ref.cpu().numpy()) < 4e-3


@pytest.mark.parametrize("mode", [L.GEMM_NT, L.GEMM_NN])
@pytest.mark.parametrize("M,N,K,res", [(5696, 768, 3072, True), (5696, 768, 2304, False), (5632, 768, 6144, False), (5700, 768, 3072, True),
                                       (41000, 768, 3072, False), (3000, 768, 2048, True), (9000, 768, 1536, False), (10900, 768, 3072, True),
                                       (12000, 768, 2304, False), (1000, 768, 1536, True), (8192, 768, 3072, True), (6100, 384, 1536, False)])
def test_gemm_stream_k(lib, mode, M, N, K, res):  # (name kept: the split of the last round grew out of a stream-K schedule)
    """Row counts that are no multiple of the tile grid - the shapes the var-len token layout produces (M = a batch's real tokens).
    (a) plain entry point: one-round N = d launches take the K-split kernel with 64 / 96 / 128 rows per tile, whichever fills the CUs
    best (csrc/gemm.hip launch_t), ragged last tile included; (b) gget_op_gemm_streamk with the split of the last round switched on (gget_debug_set key 3): the last,
    partial round splits its tiles' K range among the idle workgroups (fp32 partial tiles through the workspace, agent-scope
    release / acquire) - the result must be the plain kernel's up to fp32 summation order, repeatedly (the workspace is reused launch
    after launch: the flag epochs keep the launches apart), and exact against an fp32 matmul within the bf16 output rounding."""
    A = rnd(M, K, seed=51)
    B = rnd(N, K, seed=52) if mode == L.GEMM_NT else rnd(K, N, seed=52)
    R = rnd(M, N, seed=53) if res else None
    ref = A.float() @ (B.float().T if mode == L.GEMM_NT else B.float()) + (R.float() if res else 0)
    epi = L.EPI_RESIDUAL if res else L.EPI_NONE
    ldb = K if mode == L.GEMM_NT else N
    plain = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(mode, epi, P(A), P(B), P(plain), P(R) if res else None, M, N, K, K, ldb, N, 1, ST()))
    assert rel_l2(plain.float().cpu().numpy(), ref.cpu().numpy()) < 4e-3
    ws = torch.zeros(int(lib.gget_op_gemm_streamk_bytes()), dtype=torch.uint8, device="cuda")
    outs = []
    L.check(lib.gget_debug_set(3, 1))            # split the last round (off by default)
    L.check(lib.gget_debug_set(1, 1))            # ... on the plain 128-row persistent tile (the K-split kernel takes one-round launches otherwise)
    try:
        for it in range(3):
            Cm = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            L.check(lib.gget_op_gemm_streamk(mode, epi, P(A), P(B), P(Cm), P(R) if res else None, M, N, K, K, ldb, N, P(ws), ST()))
            outs.append(Cm)
        torch.cuda.synchronize()
    finally:
        L.check(lib.gget_debug_set(3, 0))
        L.check(lib.gget_debug_set(1, 0))
    err_flag = int(ws[512 * 4: 512 * 4 + 4].view(torch.int32)[0])
    assert err_flag == 0, "a stream-K owner gave up waiting for its contributor"
    for Cm in outs:
        assert rel_l2(Cm.float().cpu().numpy(), ref.cpu().numpy()) < 4e-3
        d = (Cm.float() - plain.float()).abs()
        # same products, another fp32 summation order in the split tiles: isolated one-ulp bf16 differences at most
        assert float(d.max()) <= 2.0 ** -6 * float(plain.float().abs().max()) and float((d > 0).float().mean()) < 0.05
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])        # deterministic: fixed split, fixed order


@pytest.mark.parametrize("M,N,K", [(1536, 768, 2048), (768, 3072, 1024), (768, 768, 8192), (6144, 768, 512)])
def test_gemm_tn_tile_192x192(lib, M, N, K):
    # weight-gradient shapes (M, N multiples of 192) take the 192x192 persistent TN tile when it fills the chip better
    A, B = rnd(K, M, seed=31), rnd(K, N, seed=32)
    ref = A.float().T @ B.float()
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(L.GEMM_TN, L.EPI_NONE, P(A), P(B), P(Cm), None, M, N, K, M, N, N, 1, ST()))
    assert rel_l2(Cm.float().cpu().numpy(), ref.cpu().numpy()) < 4e-3


@pytest.mark.parametrize("M,N,K", [(300, 211, 128), (5000, 41245 // 8 + 1, 64)])
def test_gemm_nt_odd_width(lib, M, N, K):
    # vocabulary-sized outputs need not be multiples of 4 (ogbl-ppa: 41 245): C is written up to the next multiple of 4
    # inside ldc, columns beyond stay untouched
    A, B = rnd(M, K, seed=41), rnd(N, K, seed=42)
    ldc = (N + 63) // 64 * 64
    Cm = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(L.GEMM_NT, L.EPI_NONE, P(A), P(B), P(Cm), None, M, N, K, K, K, ldc, 1, ST()))
    ref = A.float() @ B.float().T
    assert rel_l2(Cm[:, :N].float().cpu().numpy(), ref.cpu().numpy()) < 4e-3
    assert torch.all(Cm[:, (N + 3) // 4 * 4:] == 7.0)


def test_gemm_identity_layout(lib):
    # A = I with an asymmetric B: the output must be B^T exactly (bit-exact, catches row/col swaps)
    n = 128
    A = torch.eye(n, dtype=torch.bfloat16, device="cuda")
    B = rnd(n, n, seed=6)
    Cm = torch.zeros(n, n, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gemm(L.GEMM_NT, L.EPI_NONE, P(A), P(B), P(Cm), None, n, n, n, n, n, n, 1, ST()))
    assert torch.equal(Cm, B.T.contiguous())
    L.check(lib.gget_op_gemm(L.GEMM_NN, L.EPI_NONE, P(A), P(B), P(Cm), None, n, n, n, n, n, n, 1, ST()))
    assert torch.equal(Cm, B)
    L.check(lib.gget_op_gemm(L.GEMM_TN, L.EPI_NONE, P(B), P(A), P(Cm), None, n, n, n, n, n, n, 1, ST()))
    assert torch.equal(Cm, B.T.contiguous())


# ------------------------------------------------------------------------------------------ RMSNorm
@pytest.mark.parametrize("T,d", [(37, 128), (512, 768), (100, 1024)])
def test_rmsnorm(lib, T, d):
    x, w, dy, dres = rnd(T, d, seed=1), (rnd(d, seed=2) * 0.1 + 1).to(torch.bfloat16), rnd(T, d, seed=3), rnd(T, d, seed=4)
    y = torch.empty_like(x)
    rstd = torch.empty(T, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_rmsnorm_fwd(P(x), P(w), P(y), P(rstd), T, d, 1e-6, ST()))
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    var = xf.pow(2).mean(-1, keepdim=True)
    ref = wf * (xf * torch.rsqrt(var + 1e-6))
    assert rel_l2(y.float().cpu().numpy(), ref.detach().cpu().numpy()) < 6e-3  # two bf16 roundings (hf :62-67)
    np.testing.assert_allclose(rstd.cpu().numpy(), torch.rsqrt(var + 1e-6).squeeze(-1).detach().cpu().numpy(), rtol=1e-5)
    ref.backward(dy.float())
    dx = torch.empty_like(x)
    dw = torch.zeros(d, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_rmsnorm_bwd(P(dy), P(x), P(w), P(rstd), P(dres), P(dx), P(dw), T, d, ST()))
    assert rel_l2(dx.float().cpu().numpy(), (xf.grad + dres.float()).cpu().numpy()) < 4e-3
    assert rel_l2(dw.cpu().numpy(), wf.grad.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("T,d", [(5696, 768), (8320, 768), (777, 512), (4097, 1024), (23, 768)])
def test_rmsnorm_bwd_short_launch_form_is_bit_equal(lib, T, d):
    """Short launches (<= 64 rows per CU) run one 16-wave block per CU with the rows dealt in contiguous ranges (rmsnorm_bwd_wide_kernel);
    gget_debug_set(13, 0) selects the 4-wave blocks every other launch uses.  Same expressions: dx bit-equal, the weight gradient equal up to
    the order of its fp32 partial sums."""
    x, w, dy, dres = rnd(T, d, seed=1), (rnd(d, seed=2) * 0.1 + 1).to(torch.bfloat16), rnd(T, d, seed=3), rnd(T, d, seed=4)
    rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6)
    out = []
    for form in (0, 1):
        lib.gget_debug_set(13, form)
        dx = torch.full((T, d), 5.0, dtype=torch.bfloat16, device="cuda")
        dw = torch.zeros(d, dtype=torch.float32, device="cuda")
        L.check(lib.gget_op_rmsnorm_bwd(P(dy), P(x), P(w), P(rstd), P(dres), P(dx), P(dw), T, d, ST()))
        torch.cuda.synchronize()
        out.append((dx, dw))
    lib.gget_debug_set(13, 1)
    assert torch.equal(out[0][0], out[1][0]), "dx of the 16-wave form differs from the 4-wave form"
    assert rel_l2(out[1][1].cpu().numpy(), out[0][1].cpu().numpy()) < 2e-6
    # without a residual gradient
    dx0 = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
    dx1 = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(d, dtype=torch.float32, device="cuda")
    lib.gget_debug_set(13, 0)
    L.check(lib.gget_op_rmsnorm_bwd(P(dy), P(x), P(w), P(rstd), None, P(dx0), P(dw), T, d, ST()))
    lib.gget_debug_set(13, 1)
    L.check(lib.gget_op_rmsnorm_bwd(P(dy), P(x), P(w), P(rstd), None, P(dx1), P(dw), T, d, ST()))
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)


# ------------------------------------------------------------------------------------------ embedding
@pytest.mark.parametrize("gated,V", [(False, 97), (True, 97), (False, 9001), (False, 756)])
def test_embed(lib, gated, V):
    T, F, d = (300, 13, 128) if V < 9000 else (5000, 4, 128)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (T, F), generator=g)
    ids[::3, ::2] = 1   # hot <mask> id
    ids[5:9] = 0        # pad rows
    ids = ids.cuda()
    emb = rnd(V, d, seed=1)
    emb[0] = 0
    gate = rnd(F, d, seed=2) if gated else None
    out = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_embed_fwd(P(ids), P(emb), P(gate), P(out), T, F, F, d, ST()))
    ef = emb.float().requires_grad_(True)
    gf = gate.float().requires_grad_(True) if gated else None
    e = ef[ids]
    ref = torch.einsum("tfd,fd->td", e, gf) if gated else e.sum(1)
    assert rel_l2(out.float().cpu().numpy(), ref.detach().cpu().numpy()) < 4e-3
    dx = rnd(T, d, seed=3)
    ref.backward(dx.float())
    demb = torch.zeros(V, d, dtype=torch.float32, device="cuda")
    dgate = torch.zeros(F, d, dtype=torch.float32, device="cuda") if gated else None
    L.check(lib.gget_op_embed_bwd(P(ids), P(dx), P(emb), P(gate), P(demb), P(dgate), T, F, F, d, V, 0, ST()))
    want = ef.grad.clone()
    want[0] = 0  # padding_idx row receives no gradient
    assert rel_l2(demb.cpu().numpy(), want.cpu().numpy()) < 1e-4
    if gated:
        assert rel_l2(dgate.cpu().numpy(), gf.grad.cpu().numpy()) < 1e-4


def test_embed_bwd_dense_headline_shape(lib, monkeypatch):
    """T = 8192, F = 13, V = 756, d = 768: the count-matrix GEMM form (split-K, fp32 atomics) against the sorted scatter-add
    form (GGET_EMBED_SORTED is read per call by the op entry's helper only once, so the comparison is against torch)."""
    T, F, d, V = 8192, 13, 768, 756
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, V, (T, F), generator=g)
    ids[torch.rand(T, F, generator=g) < 0.5] = 1
    ids[7000:] = 0
    ids = ids.cuda()
    dx = rnd(T, d, seed=3)
    emb = rnd(V, d, seed=1)
    demb = torch.zeros(V, d, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_embed_bwd(P(ids), P(dx), P(emb), None, P(demb), None, T, F, F, d, V, 0, ST()))
    want = torch.zeros(V, d, dtype=torch.float64, device="cuda")
    want.index_add_(0, ids.reshape(-1), dx.double().repeat_interleave(F, 0))
    want[0] = 0
    assert rel_l2(demb.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert float(demb[0].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ RoPE + attention
def _rope_ref(x, pos, theta=10000.0):
    # x [B,S,H,64] fp32, pos [B,S]
    inv = 1.0 / (theta ** (torch.arange(0, 64, 2, dtype=torch.float32, device=x.device) / 64))
    fr = pos.float()[:, :, None] * inv
    emb = torch.cat((fr, fr), -1)[:, :, None, :]
    x1, x2 = x[..., :32], x[..., 32:]
    return x * emb.cos() + torch.cat((-x2, x1), -1) * emb.sin()


def _tables(maxpos):
    eng = importlib.import_module("graph-gpt_amd.engine")
    c, s = eng.rope_tables(maxpos, 64, 10000.0)
    return c.cuda(), s.cuda()


@pytest.mark.parametrize("with_pos", [False, True])
def test_rope(lib, with_pos):
    B, S, H = 3, 40, 2
    d = H * 64
    qkv = rnd(B * S, 3 * d, seed=1)
    orig = qkv.clone()
    cos, sin = _tables(64)
    pos = None
    if with_pos:
        pos = torch.arange(S)[None].repeat(B, 1)
        pos[1, 20:] = 0
        pos = pos.cuda()
    L.check(lib.gget_op_rope(P(qkv), P(cos), P(sin), P(pos), B, S, H, 0, ST()))
    pr = pos if with_pos else torch.arange(S, device="cuda")[None].repeat(B, 1)
    x = orig.float().view(B, S, 3, H, 64)
    want_q, want_k = _rope_ref(x[:, :, 0], pr), _rope_ref(x[:, :, 1], pr)
    got = qkv.float().view(B, S, 3, H, 64)
    assert rel_l2(got[:, :, 0].cpu().numpy(), want_q.cpu().numpy()) < 3e-3
    assert rel_l2(got[:, :, 1].cpu().numpy(), want_k.cpu().numpy()) < 3e-3
    assert torch.equal(got[:, :, 2], x[:, :, 2])  # v untouched
    # inverse rotation brings q,k back (orthogonality), up to two bf16 roundings
    L.check(lib.gget_op_rope(P(qkv), P(cos), P(sin), P(pos), B, S, H, 1, ST()))
    assert rel_l2(qkv.float().cpu().numpy(), orig.float().cpu().numpy()) < 6e-3


@pytest.mark.parametrize("B,S,d,with_pos", [(3, 40, 128, True), (64, 32, 768, False), (256, 32, 768, True), (128, 64, 384, False),
                                            (16, 72, 1024, True), (64, 64, 1024, True)])
def test_qkv_rope_fused(lib, B, S, d, with_pos):
    """q|k|v projection with RoPE in the GEMM epilogue (every tile shape the launcher picks: 128x128, 256x128 and the
    interleaved 128x192, incl. a 192-wide tile that straddles the k|v boundary) vs projection in fp32 followed by
    hf apply_rotary_pos_emb; v columns stay un-rotated."""
    T, H = B * S, d // 64
    x, w = rnd(T, d, seed=21), rnd(3 * d, d, scale=d ** -0.5, seed=22)
    g = torch.Generator().manual_seed(5)
    pos = torch.stack([torch.randperm(S + 7, generator=g)[:S] for _ in range(B)]).cuda() if with_pos else None
    cos, sin = _tables(max(1024, S + 16))
    qkv = torch.empty(T, 3 * d, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_qkv_rope(P(x), P(w), P(qkv), P(cos), P(sin), P(pos) if with_pos else None, T, S, d, ST()))
    ref = (x.float() @ w.float().T).view(B, S, 3, H, 64)
    pr = pos if with_pos else torch.arange(S, device="cuda")[None].expand(B, S)
    want = torch.stack([_rope_ref(ref[:, :, 0], pr), _rope_ref(ref[:, :, 1], pr), ref[:, :, 2]], dim=2).reshape(T, 3 * d)
    assert rel_l2(qkv.float().cpu().numpy(), want.cpu().numpy()) < 4e-3


def _attn_ref(qkv, lens, B, S, H, causal):
    d = H * 64
    x = qkv.view(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))  # [B,H,S,64]
    w = q @ k.transpose(2, 3) * 0.125
    keym = torch.arange(S, device=qkv.device)[None, :] >= lens[:, None]  # [B,S] True = masked
    w = w.masked_fill(keym[:, None, None, :], float("-inf"))
    if causal:
        tri = torch.ones(S, S, dtype=torch.bool, device=qkv.device).tril()
        w = w.masked_fill(~tri[None, None], float("-inf"))
    p = torch.softmax(w, -1)
    o = (p @ v).transpose(1, 2).reshape(B, S, d)
    return o


@pytest.mark.parametrize("causal", [0, 1])
@pytest.mark.parametrize("rope", [False, True])
@pytest.mark.parametrize("B,S,H", [(3, 24, 2), (2, 32, 12), (2, 72, 2), (1, 160, 3), (2, 520, 2), (1, 2048, 1), (171, 32, 12), (700, 24, 3)])   # the last two: thousands of one-wave problems (the headline shape's launch geometry)
def test_attention_fwd_bwd(lib, B, S, H, causal, rope):
    """QK^T / softmax / PV and their backward, with key-length + causal masking; rope=True also checks the fused
    rotary embedding (q,k rotated on load, dq,dk rotated back) against autograd through the un-rotated q,k."""
    d = H * 64
    qkv = rnd(B * S, 3 * d, seed=7, scale=1.0)
    lens = torch.tensor([[S, max(5, S // 2), max(1, S - 3)][i % 3] for i in range(B)], dtype=torch.int32).cuda()
    cos = sin = pos = None
    if rope:
        cos, sin = _tables(max(1024, S + 16))
        pos = torch.arange(S)[None].repeat(B, 1)
        pos[0, S // 2:] += 7  # arbitrary (fine-tune style) position ids
        pos = pos.cuda()
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, causal, P(cos), P(sin), P(pos), 0.0, 0, ST()))
    qf = qkv.float().requires_grad_(True)
    x = qf.view(B, S, 3, H, 64)
    if rope:
        qr, kr = _rope_ref(x[:, :, 0], pos), _rope_ref(x[:, :, 1], pos)
        xin = torch.stack((qr, kr, x[:, :, 2]), dim=2).reshape(B * S, 3 * d)
    else:
        xin = qf
    ref = _attn_ref(xin, lens, B, S, H, causal)
    valid = (torch.arange(S, device="cuda")[None, :] < lens[:, None])  # real query rows
    got = out.float().view(B, S, d)
    e = rel_l2(got[valid].cpu().numpy(), ref.detach()[valid].cpu().numpy())
    assert e < 8e-3, f"attn fwd rel-L2 {e}"  # bf16 P and O roundings
    # backward: upstream gradient is zero on pad query rows (as in the real model)
    dout = rnd(B * S, d, seed=8)
    dout = (dout.view(B, S, d) * valid[:, :, None]).reshape(B * S, d).contiguous()
    (ref * dout.float().view(B, S, d)).sum().backward()
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    delta = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, causal, P(cos),
                                 P(sin), P(pos), 0.0, 0, ST()))
    g = dqkv.float().view(B, S, 3, d)
    w = qf.grad.view(B, S, 3, d)
    for i, nm in enumerate("qkv"):
        e = rel_l2(g[:, :, i].cpu().numpy(), w[:, :, i].cpu().numpy())
        assert e < 2e-2, f"attn bwd d{nm} rel-L2 {e}"


def _drop_mask(seed, B, H, S, p):
    """Python twin of drop_mul() in csrc/attention.hip: keep-multiplier [B,H,S(query),S(key)] of the counter-based mask."""
    bh = np.arange(B * H, dtype=np.uint64)[:, None, None]
    q = np.arange(S, dtype=np.uint64)[None, :, None]
    k = np.arange(S, dtype=np.uint64)[None, None, :]
    M32 = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) ^ ((bh * np.uint64(0x9E3779B1)) & M32)) & M32
    x = (x + q * np.uint64(0x85EBCA77) + (k >> np.uint64(1)) * np.uint64(0xC2B2AE3D)) & M32
    x ^= x >> np.uint64(16)
    x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & M32      # v_mul_u32_u24: the low 24 bits times a 24-bit constant
    w = x ^ (x >> np.uint64(15))
    f = np.where((k & np.uint64(1)) == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    thresh = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    keep = f >= thresh
    return torch.from_numpy((keep.astype(np.float32) / (1.0 - p)).reshape(B, H, S, S))


@pytest.mark.parametrize("S,short", [(72, 50), (32, 21), (256, 130), (576, 300), (1056, 777)])   # the last two: long-sequence kernels
def test_attention_dropout(lib, S, short):
    """Dropout on the softmax output (hf eager_attention_forward :210): the kernels regenerate the same counter-based
    mask in forward, dQ and dK/dV (S <= 32: the fused single-launch backward); checked against autograd with the
    identical mask, and the drop rate is ~p."""
    B, H, p, seed = 2, 3, 0.1, 12345
    d = H * 64
    qkv = rnd(B * S, 3 * d, seed=17)
    lens = torch.tensor([S, short], dtype=torch.int32).cuda()
    mask = _drop_mask(seed, B, H, S, p).cuda()
    assert abs(float((mask == 0).float().mean()) - p) < 0.015
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, 0, None, None, None, p, seed, ST()))
    qf = qkv.float().requires_grad_(True)
    x = qf.view(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    w = q @ k.transpose(2, 3) * 0.125
    keym = torch.arange(S, device="cuda")[None, :] >= lens[:, None]
    w = w.masked_fill(keym[:, None, None, :], float("-inf"))
    ref = ((torch.softmax(w, -1) * mask) @ v).transpose(1, 2).reshape(B, S, d)
    valid = ~keym
    got = out.float().view(B, S, d)
    assert rel_l2(got[valid].cpu().numpy(), ref.detach()[valid].cpu().numpy()) < 8e-3
    dout = (rnd(B * S, d, seed=18).view(B, S, d) * valid[:, :, None]).reshape(B * S, d).contiguous()
    (ref * dout.float().view(B, S, d)).sum().backward()
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    delta = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv), P(delta), B, S, H, 0, None, None,
                                 None, p, seed, ST()))
    g, wg = dqkv.float().view(B, S, 3, d), qf.grad.view(B, S, 3, d)
    for i, nm in enumerate("qkv"):
        e = rel_l2(g[:, :, i].cpu().numpy(), wg[:, :, i].cpu().numpy())
        assert e < 2e-2, f"dropout attn bwd d{nm} rel-L2 {e}"


@pytest.mark.parametrize("rope", [False, True])
@pytest.mark.parametrize("B,S,H,causal,p", [(9, 40, 2, 0, 0.0), (7, 56, 12, 0, 0.1), (6, 64, 3, 1, 0.0), (300, 40, 12, 0, 0.1), (5, 32, 2, 0, 0.0),
                                            (4, 48, 4, 1, 0.1), (3, 72, 2, 0, 0.0)])
def test_attention_varlen_by_sample_rows(lib, B, S, H, causal, p, rope):
    """The padded width S is the batch's LONGEST graph (reference src/data/collator.py:70-111); on the var-len layout every sample is
    processed by its own row count: forward blocks behind a sample's rows exit, the S <= 64 backward is one launch whose waves take one
    32-row tile or, for a sample of 33 .. 64 rows, 2 x 2 tiles (attn_bwd_small_kernel / attn_bwd_small_two_tiles).  Checked against
    autograd through the same masks (dropout: the Python twin of the counter hash; rope: q / k rotated in memory, dq / dk rotated back)
    and against the padded-layout launch of the same samples; S = 72 keeps the two-kernel path behind the same entry covered."""
    d, seed = H * 64, 4321
    pat = [S, 33, 32, 31, 5, min(S, 47), 1, 22, min(S, 64), 17]
    lens = torch.tensor([min(S, pat[i % len(pat)]) for i in range(B)], dtype=torch.int32)
    cu = torch.zeros(B + 1, dtype=torch.int64)
    cu[1:] = torch.cumsum(lens.long(), 0)
    T = int(cu[-1])
    rows = (T + 63) // 64 * 64
    valid = (torch.arange(S)[None, :] < lens[:, None]).cuda()                      # [B,S]
    sel = valid.reshape(-1).nonzero().squeeze(1)                                      # padded row of every compact row
    qkv_u = rnd(B * S, 3 * d, seed=27)
    cos = sin = pos = None
    qf = qkv_u.float().requires_grad_(True)
    x = qf.view(B, S, 3, H, 64)
    if rope:
        cos, sin = _tables(max(1024, S + 16))
        pos = torch.arange(S)[None].repeat(B, 1)
        pos[0, S // 2:] += 7
        pos = pos.cuda()
        xin = torch.stack((_rope_ref(x[:, :, 0], pos), _rope_ref(x[:, :, 1], pos), x[:, :, 2]), dim=2).reshape(B * S, 3 * d)
        qkv_mem = xin.detach().to(torch.bfloat16).contiguous()                      # rotated in memory: the engine's layout
    else:
        xin, qkv_mem = qf, qkv_u
    # reference (on what is in memory, so the bf16 rounding of the rotated q / k is not part of the comparison)
    xm = qkv_mem.float().requires_grad_(True)
    xv = xm.view(B, S, 3, H, 64)
    q, k, v = (xv[:, :, i].transpose(1, 2) for i in range(3))
    w = q @ k.transpose(2, 3) * 0.125
    w = w.masked_fill(~valid[:, None, None, :], float("-inf"))
    if causal:
        w = w.masked_fill(~torch.ones(S, S, dtype=torch.bool, device="cuda").tril()[None, None], float("-inf"))
    pr = torch.softmax(w, -1)
    if p > 0:
        pr = pr * _drop_mask(seed, B, H, S, p).cuda()
    ref = (pr @ v).transpose(1, 2).reshape(B, S, d)
    # compact buffers
    qkv_c = torch.zeros(rows, 3 * d, dtype=torch.bfloat16, device="cuda")
    qkv_c[:T] = qkv_mem[sel]
    lens_d, rb = lens.cuda(), cu[:B].to(torch.int32).cuda()
    out_c = torch.zeros(rows, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd_varlen(P(qkv_c), P(lens_d), P(rb), P(out_c), P(lse), B, S, H, causal, P(cos), P(sin), P(pos), 1, p, seed, ST()))
    e = rel_l2(out_c[:T].float().cpu().numpy(), ref.detach().reshape(B * S, d)[sel].cpu().numpy())
    assert e < 8e-3, f"var-len attention forward rel-L2 {e}"
    assert float(out_c[T:].float().abs().max()) == 0.0 if rows > T else True         # rows behind the last sample are nobody's
    # the padded-layout launch of the same samples: same kernel arithmetic per tile -> the same bits on the real rows
    out_p = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse_p = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd(P(qkv_mem), P(lens_d), P(out_p), P(lse_p), B, S, H, causal, None, None, None, p, seed, ST()))
    assert torch.equal(out_c[:T], out_p[sel]), "var-len forward differs from the padded launch on real rows"
    vq = valid[:, None, :].expand(B, H, S)
    assert torch.equal(lse.view(B, H, S)[vq], lse_p.view(B, H, S)[vq])
    # backward
    dout = rnd(B * S, d, seed=28)
    dout = (dout.view(B, S, d) * valid[:, :, None]).reshape(B * S, d).contiguous()
    (ref * dout.float().view(B, S, d)).sum().backward()
    want = xm.grad.view(B * S, 3, d)
    if rope:   # gradient w.r.t. the UN-rotated q / k: the rotation's transpose applied to the gradient of the rotated ones
        g4 = xm.grad.view(B, S, 3, H, 64)
        want = torch.stack((_rope_ref_signed(g4[:, :, 0], pos, -1.0), _rope_ref_signed(g4[:, :, 1], pos, -1.0), g4[:, :, 2]), dim=2).reshape(B * S, 3, d)
    dout_c = torch.zeros(rows, d, dtype=torch.bfloat16, device="cuda")
    dout_c[:T] = dout[sel]
    dqkv_c = torch.full((rows, 3 * d), 7.0, dtype=torch.bfloat16, device="cuda")     # (every real row must be overwritten)
    delta = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_bwd_varlen(P(qkv_c), P(out_c), P(dout_c), P(lse), P(lens_d), P(rb), P(dqkv_c), P(delta), B, S, H, causal,
                                        P(cos), P(sin), P(pos), 1, p, seed, ST()))
    g = dqkv_c[:T].float().view(T, 3, d)
    for i, nm in enumerate("qkv"):
        e = rel_l2(g[:, i].cpu().numpy(), want[sel][:, i].cpu().numpy())
        assert e < 2e-2, f"var-len attention backward d{nm} rel-L2 {e}"
    # per-sample check of the long samples alone (5 % of a real batch: a slip there would drown in the batch-wide norm)
    for b in range(min(B, 12)):
        if int(lens[b]) > 32:
            r0, r1 = int(cu[b]), int(cu[b + 1])
            ws = want[b * S: b * S + int(lens[b])]
            for i, nm in enumerate("qkv"):
                e = rel_l2(g[r0:r1, i].cpu().numpy(), ws[:, i].cpu().numpy())
                assert e < 2.5e-2, f"sample {b} ({int(lens[b])} rows) d{nm} rel-L2 {e}"


def _rope_ref_signed(x, pos, sign, theta=10000.0):
    """rotation by sign * angle (sign = -1: the transpose / inverse rotation that maps a gradient of rotated q / k back)"""
    inv = 1.0 / (theta ** (torch.arange(0, 64, 2, dtype=torch.float32, device=x.device) / 64))
    fr = pos.float()[:, :, None] * inv * sign
    emb = torch.cat((fr, fr), -1)[:, :, None, :]
    x1, x2 = x[..., :32], x[..., 32:]
    return x * emb.cos() + torch.cat((-x2, x1), -1) * emb.sin()


# ------------------------------------------------------------------------------------------ GEGLU / CE
def test_geglu(lib):
    T, ff = 77, 512
    gu, dh = rnd(T, 2 * ff, seed=1), rnd(T, ff, seed=2)
    h = torch.empty(T, ff, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_geglu_fwd(P(gu), P(h), T, ff, ST()))
    gf = gu.float().requires_grad_(True)
    ref = torch.nn.functional.gelu(gf[:, :ff]) * gf[:, ff:]
    assert rel_l2(h.float().cpu().numpy(), ref.detach().cpu().numpy()) < 6e-3
    ref.backward(dh.float())
    dgu = torch.empty_like(gu)
    L.check(lib.gget_op_geglu_bwd(P(gu), P(dh), P(dgu), T, ff, ST()))
    assert rel_l2(dgu.float().cpu().numpy(), gf.grad.cpu().numpy()) < 4e-3


@pytest.mark.parametrize("T,d,ff", [(77, 128, 512), (600, 192, 384), (1000, 768, 3072), (96, 128, 320),
                                    (5696, 768, 3072), (3000, 128, 3072), (3000, 256, 3072), (2990, 320, 3072)])
def test_gateup_geglu_fused(lib, T, d, ff):
    """gate|up projection with the gated-GELU product in the GEMM epilogue and its backward in the epilogue of the down
    dgrad GEMM: bit-identical to the un-fused op sequence (same bf16 rounding points), close to the fp32 statement of
    hf LlamaMLP.forward :174-176.  ff = 320 takes the un-fused fallback (ff % 128 != 0).  The last four shapes run the backward
    through the 192-row persistent kernel with the DEFERRED epilogue (a tile's GEGLU' inside the next tile's K-loop): the C1
    var-len shape, K = 128 / 256 (fewer K-tiles than epilogue steps / exactly as many) and a ragged last row tile."""
    x, wgu, wdown, dy = rnd(T, d, seed=1), rnd(2 * ff, d, seed=2, scale=0.08), rnd(d, ff, seed=3, scale=0.05), rnd(T, d, seed=4)
    gu = torch.zeros(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
    h = torch.zeros(T, ff, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_gateup_geglu(P(x), P(wgu), P(gu), P(h), T, d, ff, ST()))
    # un-fused sequence through the same library
    gu2, h2 = torch.zeros_like(gu), torch.zeros_like(h)
    L.check(lib.gget_op_gemm(L.GEMM_NT, 0, P(x), P(wgu), P(gu2), None, T, 2 * ff, d, d, d, 2 * ff, 1, ST()))
    L.check(lib.gget_op_geglu_fwd(P(gu2), P(h2), T, ff, ST()))
    assert torch.equal(gu, gu2), "gate|up pre-activations differ from the plain GEMM"
    assert torch.equal(h, h2), "fused GEGLU differs from GEMM + geglu_fwd"
    guf = x.float() @ wgu.float().t()
    ref = torch.nn.functional.gelu(guf[:, :ff]) * guf[:, ff:]
    assert rel_l2(h.float().cpu().numpy(), ref.cpu().numpy()) < 8e-3
    # backward
    dgu = torch.zeros(T, 2 * ff, dtype=torch.bfloat16, device="cuda")
    dh_s = torch.zeros(T, ff, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_down_dgrad_geglu(P(dy), P(wdown), P(gu), P(dgu), P(dh_s), T, d, ff, ST()))
    dh2, dgu2 = torch.zeros(T, ff, dtype=torch.bfloat16, device="cuda"), torch.zeros_like(dgu)
    L.check(lib.gget_op_gemm(L.GEMM_NN, 0, P(dy), P(wdown), P(dh2), None, T, ff, d, d, ff, ff, 1, ST()))
    L.check(lib.gget_op_geglu_bwd(P(gu), P(dh2), P(dgu2), T, ff, ST()))
    assert torch.equal(dgu, dgu2), "fused GEGLU backward differs from GEMM + geglu_bwd"
    gf = gu.float().requires_grad_(True)
    (torch.nn.functional.gelu(gf[:, :ff]) * gf[:, ff:]).backward(dy.float() @ wdown.float())
    assert rel_l2(dgu.float().cpu().numpy(), gf.grad.cpu().numpy()) < 6e-3


@pytest.mark.parametrize("V,ld", [(756, 768), (300, 320), (97, 128), (1500, 1536), (41245, 41280), (211, 212)])
def test_cross_entropy(lib, V, ld):
    """Row-in-registers kernels (ld <= 512 / 1024 / 2048) and the generic kernel (wide vocabulary, or ld % 8 != 0)."""
    rows = 500
    logits = torch.zeros(rows, ld, dtype=torch.bfloat16, device="cuda")
    logits[:, :V] = rnd(rows, V, seed=1, scale=2.0)
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2)).to(torch.int32).cuda()
    n_rows = torch.tensor([rows - 37], dtype=torch.int32).cuda()
    loss_sum = torch.zeros(1, dtype=torch.float32, device="cuda")
    dl = torch.full((rows, ld), 3.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_ce_fwd_bwd(P(logits), ld, P(labels), None, P(n_rows), rows, V, P(loss_sum), P(dl), 0.0, 1, ST()))
    n = rows - 37
    lf = logits[:n, :V].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, labels[:n].long())
    ref.backward()
    got = loss_sum.item() / n
    assert abs(got - ref.item()) < 1e-5 * abs(ref.item()) + 1e-6
    assert rel_l2(dl[:n, :V].float().cpu().numpy(), lf.grad.cpu().numpy()) < 5e-3
    assert torch.all(dl[:n, V:] == 0), "pad columns of dlogits must be zero (they feed the dgrad GEMM)"
    assert torch.all(dl[n:] == 3.0), "rows beyond the device-side count must not be touched"


# ------------------------------------------------------------------------------------------ in-model SMTP masking (N1)
@pytest.mark.parametrize("B,S,F,V,rate,power,rep,glob", [(5, 24, 13, 756, 1.0, 1.0, 0.0, False), (64, 40, 4, 300, 0.5, 1.0, 0.4, False),
                                                        (7, 16, 1, 97, 0.3, 0.5, 0.5, True), (256, 32, 13, 756, 1.0, 1.0, 0.0, False)])
def test_smtp2d_kernel_matches_oracle(lib, B, S, F, V, rate, power, rep, glob):
    """gget_op_smtp2d vs the oracle (pinned against the reference's prepare_for_2d_smtp_inputs_labels) fed with the Python
    twin of the kernel's counter-hash draws: masked ids and labels are bit-exact; the batch layout is the reference's
    (F feature columns + 4 pos_deco columns, node index in column F+2)."""
    from oracle import gget_oracle as O
    smtp = importlib.import_module("graph-gpt_amd.smtp")
    g = torch.Generator().manual_seed(B * 131 + S)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    full = torch.zeros(B, S, F + 4, dtype=torch.int64)
    full[:, :, :F] = torch.randint(2, V, (B, S, F), generator=g)
    for b in range(B):
        full[b, lens[b]:, :F] = 0
        full[b, : lens[b], F + 2] = torch.randint(0, max(2, int(lens[b]) * 2 // 3), (int(lens[b]),), generator=g)
    seed = 0xC0FFEE + B
    dev = full.cuda()
    got_ids, got_lab = smtp.smtp2d_mask(dev, dev[:, :, F + 2], F, smtp_2d_rate=rate, power=power, replace_rate=rep, vocab=V,
                                        global_2d_mask=glob, seed=seed)
    us, ur, uc, sh, urep = smtp.draws(seed, B, S, F)
    want_ids, want_lab = O.smtp_2d_inputs_labels(full[:, :, :F].contiguous(), full[:, :, F + 2].contiguous(), us, ur, uc, sh, urep,
                                                 smtp_2d_rate=rate, power=power, replace_rate=rep, vocab=V, global_2d_mask=glob)
    assert torch.equal(got_lab.cpu(), want_lab)
    assert torch.equal(got_ids.cpu(), want_ids)
    masked = (want_lab != -100)
    assert masked.any() and not masked.all()
    # statistics of the draws themselves: uniform in [0,1), shift ~ N(0, 10^2)
    assert float(uc.min()) >= 0 and float(uc.max()) < 1
    if B * S * F > 5000:
        assert abs(float(uc.mean()) - 0.5) < 0.02
        assert abs(float(sh.std()) - 10.0) < 0.5 and abs(float(sh.mean())) < 0.5


# ------------------------------------------------------------------------------------------ generation confidence (N3)
@pytest.mark.parametrize("V,mode,temperature,top_p,top_k,alg_temp",
                         [(300, 0, 0.8, 0.9, 20, 0.0), (756, 0, 0.5, 0.0, 30, 0.4), (300, 1, 1.0, 0.95, 0, 0.0),
                          (97, 2, 0.7, 0.0, 0, 0.0), (756, 0, 0.0, 0.0, 0, 0.0), (1500, 0, 1.3, 0.8, 50, 0.0),
                          (300, 0, 0.0, 0.5, 5, 0.3)])
def test_token_sample_matches_oracle(lib, V, mode, temperature, top_p, top_k, alg_temp):
    """The sampling kernel of the generation loop (gget_op_token_sample: temperature, top-p, top-k, categorical draw by inverse
    CDF, margin / entropy confidence, Gumbel perturbation) against the oracle's restatement of the reference's sample_tokens
    fed with the Python twin of the kernel's draws.  A candidate may differ only where the uniform sits within rounding
    distance of a CDF step (fp32 summation order); confidences agree to 1e-5."""
    gen = importlib.import_module("graph-gpt_amd.generation")
    from oracle import gget_oracle as O
    R, ld, seed = 333, ((V + 63) // 64) * 64, 777
    logits = torch.zeros(R, ld, dtype=torch.bfloat16, device="cuda")
    logits[:, :V] = rnd(R, V, seed=5, scale=2.0)
    logits[7, :V] = 0.25                      # a row of ties
    conf = torch.empty(R, dtype=torch.float32, device="cuda")
    tok = torch.empty(R, dtype=torch.int64, device="cuda")
    L.check(lib.gget_op_token_sample(P(logits), ld, R, V, mode, temperature, top_p, top_k, alg_temp, seed, P(conf), P(tok), ST()))
    inv = np.float32(1.0 / 16777216.0)
    smtp = importlib.import_module("graph-gpt_amd.smtp")
    u = torch.from_numpy(smtp._rng24(seed, 32, np.arange(R), 0).astype(np.float32) * inv)
    u2 = torch.from_numpy(smtp._rng24(seed, 33, np.arange(R), 0).astype(np.float32) * inv)
    lf = logits[:, :V].float().cpu()
    c_ref, x_ref = O.sample_tokens(lf, temperature=temperature, top_p=top_p if top_p > 0 else None,
                                   top_k=top_k if top_k > 0 else None, margin_confidence=mode == 1, neg_entropy=mode == 2, u=u)
    got_t, got_c = tok.cpu(), conf.cpu()
    diff = got_t != x_ref
    if diff.any():
        # allowed only next to a CDF step
        z = lf / temperature if temperature > 0 else lf
        if top_p > 0: z = O.top_p_logits(z, top_p)
        if top_k > 0: z = O.top_k_logits(z, top_k)
        cdf = torch.softmax(z, -1).cumsum(-1)
        near = ((cdf - u[:, None]).abs().min(dim=-1).values < 2e-6)
        assert temperature > 0 and bool((near | ~diff).all()), f"{int(diff.sum())} candidates differ away from CDF steps"
        assert int(diff.sum()) <= 3
    same = ~diff
    if mode == 0 and temperature > 0:
        pass   # confidence = probability of the candidate: compare where the candidates agree (below)
    if alg_temp > 0:
        c_ref = c_ref / alg_temp + (-torch.log(-torch.log(u2 + 1e-9) + 1e-9))
    np.testing.assert_allclose(got_c[same].numpy(), c_ref[same].numpy(), rtol=2e-4, atol=2e-5)
    if temperature == 0:
        assert torch.equal(got_t, lf.argmax(-1)) or top_p > 0 or top_k > 0


def test_unmask_origin_kernel(lib):
    gen = importlib.import_module("graph-gpt_amd.generation")
    B, N, seed = 5, 77, 4242
    g = torch.Generator().manual_seed(1)
    x = torch.randint(1, 4, (B, N), generator=g)
    cand = torch.randint(10, 300, (B, N), generator=g)
    xd = x.cuda().clone()
    s_it = gen.iteration_seed(seed, 2)
    L.check(lib.gget_op_unmask_origin(P(xd), P(cand.cuda()), B, N, 0.37, s_it, 1, ST()))
    u = gen.draws(seed, 2, B, N)[2]
    want = torch.where((x == 1) & (u < np.float32(0.37)), cand, x)
    assert torch.equal(xd.cpu(), want) and ((x == 1) & (want != 1)).any() and ((x == 1) & (want == 1)).any()


@pytest.mark.parametrize("R,V", [(50, 97), (1000, 756), (333, 41245)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_token_confidence(lib, R, V, mode):
    """gget_op_token_confidence vs sample_tokens at temperature 0 (oracle) on the same bf16 logits: identical arg-max
    tokens, confidences to fp32 round-off."""
    from oracle import gget_oracle as O
    ld = (V + 63) // 64 * 64
    g = torch.Generator().manual_seed(R + V)
    logits = (torch.randn(R, ld, generator=g) * 3).to(torch.bfloat16)
    logits[0, :V] = logits[0, 0]          # a full tie: lowest index wins (torch.max semantics)
    dev = logits.cuda()
    conf = torch.empty(R, dtype=torch.float32, device="cuda")
    tok = torch.empty(R, dtype=torch.int64, device="cuda")
    L.check(lib.gget_op_token_confidence(P(dev), ld, R, V, mode, P(conf), P(tok), ST()))
    want_c, want_t = O.sample_tokens_t0(logits[:, :V].float(), margin_confidence=(mode == 1), neg_entropy=(mode == 2))
    got_t = tok.cpu()
    # bf16 logits tie often: the kernel and torch must agree on the VALUE of the chosen logit, and on the index when unique
    picked = logits[:, :V].float().gather(1, got_t[:, None])[:, 0]
    assert torch.equal(picked, logits[:, :V].float().max(dim=1).values)
    uniq = (logits[:, :V].float() == picked[:, None]).sum(1) == 1
    assert torch.equal(got_t[uniq], want_t[uniq])
    assert int(got_t[0]) == 0
    np.testing.assert_allclose(conf.cpu().numpy(), want_c.numpy(), rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------------------------------ packed rows (N2)
def _packed_ranges(B, S, seed, pad_tail=True):
    """Block-diagonal layout: random graph lengths back to back; returns (mask3d [B,S,S] i64, lo, hi [B,S] i32)."""
    rng = np.random.RandomState(seed)
    m = np.zeros((B, S, S), np.int64)
    lo = np.zeros((B, S), np.int32)
    hi = np.full((B, S), -1, np.int32)
    for b in range(B):
        n = 0
        limit = S - (rng.randint(1, 9) if (pad_tail and b % 2 == 0) else 0)
        while n < limit:
            ln = int(min(limit - n, rng.randint(3, max(4, S // 3))))
            m[b, n:n + ln, n:n + ln] = 1
            lo[b, n:n + ln], hi[b, n:n + ln] = n, n + ln - 1
            n += ln
    return m, lo, hi


@pytest.mark.parametrize("B,S,H,causal,drop", [(3, 24, 2, 0, 0.0), (2, 72, 2, 0, 0.0), (2, 160, 3, 0, 0.0), (1, 520, 2, 0, 0.0),
                                               (2, 96, 2, 1, 0.0), (2, 72, 2, 0, 0.1), (2, 520, 2, 0, 0.1), (1, 1024, 1, 1, 0.1)])
def test_attention_packed_ranges(lib, B, S, H, causal, drop):
    """Attention on packed rows (per-token key ranges from a block-diagonal mask) vs autograd with the explicit [S,S] mask:
    forward and all three gradients; rows of the padding tail produce zeros.  drop > 0: with attention dropout (the packed
    pre-train workload runs it), same counter-based mask through the Python twin - S >= 256 are the long-sequence kernels."""
    d = H * 64
    dseed = 4321
    dmask = _drop_mask(dseed, B, H, S, drop).cuda() if drop > 0 else None
    m3, lo_np, hi_np = _packed_ranges(B, S, seed=S + B)
    qkv = rnd(B * S, 3 * d, seed=23)
    mask3d = torch.from_numpy(m3).cuda()
    lo = torch.empty(B, S, dtype=torch.int32, device="cuda")
    hi = torch.empty(B, S, dtype=torch.int32, device="cuda")
    L.check(lib.gget_op_ranges_from_mask3d(P(mask3d), P(lo), P(hi), B, S, ST()))
    assert torch.equal(lo.cpu(), torch.from_numpy(lo_np)) and torch.equal(hi.cpu(), torch.from_numpy(hi_np))
    out = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd_ranges(P(qkv), P(lo), P(hi), P(out), P(lse), B, S, H, causal, drop, dseed, ST()))
    x = qkv.float().view(B, S, 3, H, 64).detach().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    w = q @ k.transpose(2, 3) * 0.125
    allow = mask3d.bool()
    if causal:
        allow = allow & torch.ones(S, S, dtype=torch.bool, device="cuda").tril()[None]
    w = w.masked_fill(~allow[:, None], float("-inf"))
    p = torch.softmax(w, -1).nan_to_num(0.0)           # padding rows: no key at all
    if dmask is not None:
        p = p * dmask
    ref = (p @ v).transpose(1, 2).reshape(B * S, d)
    valid = torch.from_numpy(hi_np >= lo_np).cuda().view(B * S)
    assert rel_l2(out.float()[valid].cpu().numpy(), ref[valid].detach().cpu().numpy()) < 8e-3
    assert torch.all(out[~valid] == 0)
    dout = rnd(B * S, d, seed=29)
    dout[~valid] = 0
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_bwd_ranges(P(qkv), P(out), P(dout), P(lse), P(lo), P(hi), P(dqkv), P(delta), B, S, H, causal, drop, dseed, ST()))
    got = dqkv.float().view(B, S, 3, H, 64)
    for i, nm in enumerate("qkv"):
        e = rel_l2(got[:, :, i].cpu().numpy(), x.grad[:, :, i].cpu().numpy())
        assert e < 2e-2, f"packed attn bwd d{nm} rel-L2 {e}"


@pytest.mark.parametrize("B,S,H,causal,drop,packed", [(2, 256, 2, 0, 0.0, False), (2, 520, 3, 0, 0.1, False), (1, 1056, 2, 1, 0.1, False),
                                                      (1, 2048, 2, 0, 0.1, False), (2, 520, 2, 0, 0.1, True), (1, 1024, 1, 1, 0.0, True),
                                                      (3, 300, 12, 0, 0.1, False)])
def test_attention_bwd_fused(lib, B, S, H, causal, drop, packed):
    """One-pass long-sequence backward (gget_op_attn_bwd_fused: S, dP and the softmax backward once; dQ through per-key-block bf16
    partials summed in fp32) against autograd with the identical dropout mask AND against the two-kernel backward of the same library:
    all three to a few bf16 roundings (dQ is accumulated in another order); a second call reproduces the first bit for bit."""
    d = H * 64
    dseed = 777
    dmask = _drop_mask(dseed, B, H, S, drop).cuda() if drop > 0 else None
    qkv = rnd(B * S, 3 * d, seed=31)
    lens = lo = hi = None
    if packed:
        m3, lo_np, hi_np = _packed_ranges(B, S, seed=S + B)
        lo, hi = torch.from_numpy(lo_np).cuda(), torch.from_numpy(hi_np).cuda()
        allow = torch.from_numpy(m3).cuda().bool()
        valid = torch.from_numpy(hi_np >= lo_np).cuda().view(B * S)
    else:
        lens = torch.tensor([[S, max(5, S // 2 + 3), S - 7][i % 3] for i in range(B)], dtype=torch.int32).cuda()
        keyok = torch.arange(S, device="cuda")[None, :] < lens[:, None]
        allow = keyok[:, None, :].expand(B, S, S).clone()
        valid = keyok.reshape(B * S)
    if causal:
        allow = allow & torch.ones(S, S, dtype=torch.bool, device="cuda").tril()[None]
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    if packed:
        L.check(lib.gget_op_attn_fwd_ranges(P(qkv), P(lo), P(hi), P(out), P(lse), B, S, H, causal, drop, dseed, ST()))
    else:
        L.check(lib.gget_op_attn_fwd(P(qkv), P(lens), P(out), P(lse), B, S, H, causal, None, None, None, drop, dseed, ST()))
    x = qkv.float().view(B, S, 3, H, 64).detach().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    w = (q @ k.transpose(2, 3) * 0.125).masked_fill(~allow[:, None], float("-inf"))
    p = torch.softmax(w, -1).nan_to_num(0.0)
    if dmask is not None:
        p = p * dmask
    ref = (p @ v).transpose(1, 2).reshape(B * S, d)
    dout = rnd(B * S, d, seed=37)
    dout[~valid] = 0
    ref.backward(dout.float())
    # the two-kernel form
    dqkv2 = torch.zeros_like(qkv)
    delta = torch.full((B * H * S,), float("nan"), dtype=torch.float32, device="cuda")
    if packed:
        L.check(lib.gget_op_attn_bwd_ranges(P(qkv), P(out), P(dout), P(lse), P(lo), P(hi), P(dqkv2), P(delta), B, S, H, causal, drop, dseed, ST()))
    else:
        L.check(lib.gget_op_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(lens), P(dqkv2), P(delta), B, S, H, causal, None, None, None, drop, dseed, ST()))
    # the fused form, twice
    acc = torch.full(((S + 255) // 256, B * S, d), float("nan"), dtype=torch.bfloat16, device="cuda")    # scratch: contents irrelevant
    runs = []
    for _ in range(2):
        dqkv = torch.zeros_like(qkv)
        delta1 = torch.full((B * H * S,), float("nan"), dtype=torch.float32, device="cuda")
        L.check(lib.gget_op_attn_bwd_fused(P(qkv), P(out), P(dout), P(lse), P(lens), P(lo), P(hi), P(dqkv), P(delta1), P(acc), B, S, H, causal, drop, dseed, ST()))
        runs.append(dqkv.float().view(B, S, 3, H, 64))
    got, got2 = runs
    assert torch.isfinite(got).all()
    for i, nm in enumerate("qkv"):
        e = rel_l2(got[:, :, i].cpu().numpy(), x.grad[:, :, i].cpu().numpy())
        assert e < 2e-2, f"fused attn bwd d{nm} rel-L2 {e}"
    two = dqkv2.float().view(B, S, 3, H, 64)
    vr = valid.view(B, S)
    # (delta = rowsum(dO * O) is summed in another order than inside the dQ kernel: dK / dV agree to a few bf16 roundings, not bit for bit)
    for i in (0, 1, 2):
        assert rel_l2(got[:, :, i][vr].cpu().numpy(), two[:, :, i][vr].cpu().numpy()) < 3e-3
    assert torch.equal(got2, got), "the fused backward must be reproducible"


@pytest.mark.parametrize("B,S,F,power", [(6, 24, 13, 1.0), (64, 32, 13, 1.0), (5, 40, 4, 2.0), (3, 2048, 13, 0.5), (4, 16, 1, 1.0)])
def test_smtp_rows_kernel_matches_oracle(lib, B, S, F, power):
    """Collator masking on the device (gget_op_smtp_rows) vs the oracle's _mask_stacked_input_ids_v2 fed with the cell list
    of the Python twin of the kernel's keys: exact ids / labels, exactly ceil(len*F*alpha) masked cells per sample."""
    from oracle import gget_oracle as O
    smtp = importlib.import_module("graph-gpt_amd.smtp")
    g = torch.Generator().manual_seed(B + S)
    lens = torch.randint(max(2, S // 3), S + 1, (B,), generator=g)
    ids = torch.randint(2, 700, (B, S, F), generator=g)
    for b in range(B):
        ids[b, lens[b]:] = 0
    ids[0, 1, 0] = 0            # a pad-valued cell inside a sequence
    seed = 4242 + B
    got_ids, got_lab, got_w = smtp.smtp_mask_rows(ids.cuda(), lens.cuda(), power=power, seed=seed, dlm_wgt=True)
    sel = smtp.row_mask_selection(seed, lens.numpy(), S, F, power=power)
    for b in range(B):
        n = int(lens[b])
        idx, alpha, wgt = sel[b]
        assert len(idx) == int(np.ceil(n * F * alpha))
        want_ids, want_lab = O.mask_stacked_input_ids_v2(ids[b, :n].numpy(), idx)
        np.testing.assert_array_equal(got_ids[b, :n].cpu().numpy(), want_ids)
        np.testing.assert_array_equal(got_lab[b, :n].cpu().numpy(), want_lab)
        assert torch.all(got_lab[b, n:] == -100) and torch.all(got_ids[b, n:] == 0)
        assert int((got_lab[b] != -100).sum()) == len(idx)
        assert abs(float(got_w[b]) - wgt) <= 1e-6 * wgt
        a2, w2 = O.smtp_mask_ratio((1.0 - alpha) ** (1.0 / power) and ((1.0 - alpha) ** (1.0 / power) - 0.01) / 0.98, 0.01, 0.99, power)
        assert abs(a2 - alpha) < 1e-9


@pytest.mark.parametrize("varlen", [False, True])
@pytest.mark.parametrize("B,S,H,causal,p", [(5, 24, 2, 0, 0.0), (3, 32, 12, 0, 0.1), (7, 32, 12, 1, 0.0), (260, 32, 12, 0, 0.1), (4, 16, 4, 0, 0.0),
                                            (3, 32, 8, 0, 0.0)])
def test_attn_oproj_norm_fused_forward(lib, B, S, H, causal, p, varlen):
    """Round 5: attention + o projection + residual + RMSNorm of a decoder layer in one launch for S <= 32 (one workgroup per
    sample; attention.hip attn_oproj_fwd_kernel) against the three launches it replaces - gget_op_attn_fwd (bit-equal attention
    output and lse: same arithmetic, same dropout hash), an fp32 statement of x_mid = x_in + attn Wo^T (one bf16 rounding) and of
    hf LlamaRMSNorm on the bf16 x_mid - on the padded [B,S] grid and on the var-len token layout (sample b at rows
    [cu[b], cu[b] + len[b]); rows of the NEXT sample must not be touched)."""
    d = H * 64
    lens = torch.tensor([[S, max(3, S // 2), max(1, S - 3), 1][i % 4] for i in range(B)], dtype=torch.int32)
    if varlen:
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(lens, 0)
        rows = (int(cu[-1]) + 63) // 64 * 64
        row_base = cu[:B].contiguous().cuda()
    else:
        rows, row_base = B * S, None
    qkv = rnd(rows, 3 * d, seed=27)
    wo = rnd(d, d, seed=28, scale=0.03)
    x_in = rnd(rows, d, seed=29)
    nw = (1.0 + 0.1 * torch.randn(d, generator=torch.Generator().manual_seed(30))).to(torch.bfloat16).cuda()
    lens_d = lens.cuda()
    seed = 4321
    sentinel = 7.0
    outs = {k: torch.full((rows, d), sentinel, dtype=torch.bfloat16, device="cuda") for k in ("attn", "xmid", "xn")}
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    rstd = torch.full((rows,), sentinel, dtype=torch.float32, device="cuda")
    taken = C.c_int32(0)
    wo_f, wo_b = torch.empty_like(wo), torch.empty_like(wo)
    L.check(lib.gget_op_pack_wo(P(wo), 0, P(wo_f), P(wo_b), d, 1, ST()))
    L.check(lib.gget_op_attn_oproj_fwd(P(qkv), P(lens_d), P(row_base), P(outs["attn"]), P(lse), P(wo_f), P(x_in), P(outs["xmid"]), P(nw),
                                       P(outs["xn"]), P(rstd), B, S, H, causal, 1e-6, p, seed, ST(), C.byref(taken)))
    assert taken.value == 1
    torch.cuda.synchronize()
    # the rows of the samples (padded layout: every row of the grid is computed like the three launches do)
    if varlen:
        live = torch.zeros(rows, dtype=torch.bool)
        for b in range(B):
            live[int(cu[b]): int(cu[b]) + int(lens[b])] = True
    else:
        live = torch.ones(rows, dtype=torch.bool)
    live = live.cuda()
    for k in ("attn", "xmid", "xn"):
        assert bool((outs[k][~live].float() == sentinel).all()), f"{k}: rows outside the samples were written"
    assert bool((rstd[~live] == sentinel).all())
    # 1. attention output and lse: the single-wave kernel's, bit for bit
    if varlen:
        # (the op entry of the separate kernel has no row_base argument: compare on a padded copy of the same samples)
        qkv_p = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
        for b in range(B):
            qkv_p[b * S: b * S + int(lens[b])] = qkv[int(cu[b]): int(cu[b]) + int(lens[b])]
    else:
        qkv_p = qkv
    ref_attn = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    ref_lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd(P(qkv_p), P(lens_d), P(ref_attn), P(ref_lse), B, S, H, causal, None, None, None, p, seed, ST()))
    torch.cuda.synchronize()
    for b in range(B):
        n = int(lens[b])
        r0 = int(cu[b]) if varlen else b * S
        assert torch.equal(outs["attn"][r0: r0 + n], ref_attn[b * S: b * S + n]), f"sample {b}: attention output differs"
        a = lse.view(B, H, S)[b, :, :n]
        assert torch.equal(a, ref_lse.view(B, H, S)[b, :, :n]), f"sample {b}: lse differs"
    # 2. x_mid = bf16(x_in + attn Wo^T) in fp32 from the kernel's own bf16 attention output
    want_mid = (x_in.float() + outs["attn"].float() @ wo.float().t())
    got_mid = outs["xmid"].float()
    err = (got_mid[live] - want_mid[live]).abs().max().item()
    scale = want_mid[live].abs().max().item()
    assert err <= 2 ** -7 * scale, f"x_mid max-abs {err} (scale {scale})"          # one bf16 rounding of values up to `scale`
    assert rel_l2(got_mid[live].cpu().numpy(), want_mid[live].cpu().numpy()) < 3e-3
    # 3. RMSNorm of the kernel's bf16 x_mid, hf's rounding points (:62-67)
    xm = outs["xmid"].float()
    want_rstd = torch.rsqrt((xm * xm).mean(-1) + 1e-6)
    assert torch.allclose(rstd[live], want_rstd[live], rtol=2e-6, atol=0)
    want_xn = nw.float() * (xm * rstd[:, None]).to(torch.bfloat16).float()
    assert torch.equal(outs["xn"][live], want_xn.to(torch.bfloat16)[live])


def test_attn_oproj_not_taken_outside_its_shapes(lib):
    """S > 32 or a head count without an instantiation (H = 16: the tiles of 16 waves do not fit the LDS): *taken = 0 and nothing is written."""
    for S, H in ((40, 2), (32, 3), (32, 16)):
        d = H * 64
        z = torch.zeros(2 * S, 3 * d, dtype=torch.bfloat16, device="cuda")
        out = torch.full((2 * S, d), 3.0, dtype=torch.bfloat16, device="cuda")
        lens = torch.tensor([S, S], dtype=torch.int32).cuda()
        taken = C.c_int32(5)
        L.check(lib.gget_op_attn_oproj_fwd(P(z), P(lens), None, P(out), None, P(z), P(z), P(out), P(z), P(out), None, 2, S, H, 0, 1e-6, 0.0, 0, ST(),
                                           C.byref(taken)))
        torch.cuda.synchronize()
        assert taken.value == 0 and bool((out.float() == 3.0).all())


def test_pack_wo_layout(lib):
    """gget_op_pack_wo: fwd[((T KS + s) 64 + lane) 8 + e] = w[16 T + lane % 16][32 s + 8 (lane / 16) + e], bwd the same of w^T; several
    layers a stride apart."""
    d, Ls, stride = 128, 3, 128 * 128 + 256
    buf = rnd(Ls * stride, seed=41)
    fwd = torch.empty(Ls, d * d, dtype=torch.bfloat16, device="cuda")
    bwd = torch.empty_like(fwd)
    L.check(lib.gget_op_pack_wo(P(buf), stride, P(fwd), P(bwd), d, Ls, ST()))
    torch.cuda.synchronize()
    KS = d // 32
    lane = torch.arange(64)
    for l in range(Ls):
        w = buf[l * stride: l * stride + d * d].view(d, d).cpu()
        for src, got in ((w, fwd[l].cpu()), (w.t().contiguous(), bwd[l].cpu())):
            want = torch.empty(d // 16, KS, 64, 8, dtype=torch.bfloat16)
            for T in range(d // 16):
                for s_ in range(KS):
                    rows = 16 * T + lane % 16
                    cols = 32 * s_ + 8 * (lane // 16)
                    want[T, s_] = torch.stack([src[rows, cols + e] for e in range(8)], dim=1)
            assert torch.equal(got.view(-1), want.view(-1))


@pytest.mark.parametrize("varlen", [False, True])
@pytest.mark.parametrize("B,S,H,causal,p,rope", [(5, 24, 2, 0, 0.0, False), (3, 32, 12, 0, 0.1, True), (7, 32, 12, 1, 0.0, True), (260, 32, 12, 0, 0.1, False),
                                                 (4, 16, 4, 0, 0.0, True), (3, 32, 8, 0, 0.0, False),
                                                 # 32 < S <= 64, var-len layout: samples of 33 .. 64 rows take phases A / B over two row tiles, their
                                                 # attention backward is attn_bwd_long_kernel's from the dattn rows written for them
                                                 (6, 40, 12, 0, 0.1, True), (5, 56, 2, 1, 0.0, False), (9, 64, 8, 0, 0.1, False), (260, 48, 12, 0, 0.1, True),
                                                 (7, 40, 4, 1, 0.1, False)])
def test_attn_oproj_norm_fused_backward(lib, B, S, H, causal, p, rope, varlen):
    """The backward counterpart (attn_oproj_bwd_kernel: RMSNorm backward + o projection dgrad + attention backward per sample) against the
    three launches it replaces, run through their own op entries on the same inputs: dx_mid and the norm weight gradient of
    gget_op_rmsnorm_bwd, dattn = dx_mid Wo by an fp32 matmul rounded to bf16 (what the GEMM stores), dqkv of gget_op_attn_bwd (same
    arithmetic and dropout hash; its dO operand is the bf16 dattn)."""
    if S > 32 and not varlen:
        pytest.skip("the padded layout is covered for S <= 32")
    d = H * 64
    lens = torch.tensor([[S, max(3, S // 2), max(1, S - 3), 1][i % 4] for i in range(B)], dtype=torch.int32)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    if varlen:
        rows = (int(cu[-1]) + 63) // 64 * 64
        row_base = cu[:B].contiguous().cuda()
        r0 = [int(cu[b]) for b in range(B)]
    else:
        rows, row_base = B * S, None
        r0 = [b * S for b in range(B)]
    nrow = [int(lens[b]) if varlen else S for b in range(B)]          # rows the sample owns in the buffers
    live = torch.zeros(rows, dtype=torch.bool)
    for b in range(B):
        live[r0[b]: r0[b] + nrow[b]] = True
    live = live.cuda()
    qkv = rnd(rows, 3 * d, seed=51)
    wo = rnd(d, d, seed=52, scale=0.03)
    x_mid, dxn, dres = rnd(rows, d, seed=53), rnd(rows, d, seed=54, scale=0.5), rnd(rows, d, seed=55, scale=0.5)
    nw = (1.0 + 0.1 * torch.randn(d, generator=torch.Generator().manual_seed(56))).to(torch.bfloat16).cuda()
    rstd = torch.rsqrt((x_mid.float() ** 2).mean(-1) + 1e-6)
    lens_d = lens.cuda()
    seed = 9876
    cos = sin = pos = None
    if rope:
        cos, sin = _tables(1024)
        pos = (torch.arange(S)[None].repeat(B, 1) + torch.arange(B)[:, None] % 5).cuda()
    # forward pieces the backward needs: lse of the attention on these q, k, v (padded copy for the separate kernels)
    qkv_p = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    for b in range(B):
        qkv_p[b * S: b * S + nrow[b]] = qkv[r0[b]: r0[b] + nrow[b]]
    out_p = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_attn_fwd(P(qkv_p), P(lens_d), P(out_p), P(lse), B, S, H, causal, None, None, None, p, seed, ST()))
    # ---- fused
    wo_f, wo_b = torch.empty_like(wo), torch.empty_like(wo)
    L.check(lib.gget_op_pack_wo(P(wo), 0, P(wo_f), P(wo_b), d, 1, ST()))
    sentinel = 5.0
    dx_mid = torch.full((rows, d), sentinel, dtype=torch.bfloat16, device="cuda")
    dqkv = torch.full((rows, 3 * d), sentinel, dtype=torch.bfloat16, device="cuda")
    copies, cstride = 4, 1024
    dw = torch.zeros(copies * cstride, dtype=torch.float32, device="cuda")
    taken = C.c_int32(0)
    dattn_long = torch.full((rows, d), sentinel, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gget_op_attn_oproj_bwd(P(dxn), P(x_mid), P(nw), P(rstd), P(dres), P(dx_mid), P(dw), copies, cstride, P(wo_b), P(qkv), P(lse),
                                       P(lens_d), P(row_base), P(dqkv), B, S, H, causal, P(cos), P(sin), P(pos), p, seed, rows, ST(), C.byref(taken),
                                       P(dattn_long)))
    assert taken.value == 1
    torch.cuda.synchronize()
    # (dattn exists in memory for the rows of 33 .. 64-row samples only)
    long_rows = torch.zeros(rows, dtype=torch.bool)
    for b in range(B):
        if varlen and nrow[b] > 32:
            long_rows[r0[b]: r0[b] + nrow[b]] = True
    long_rows = long_rows.cuda()
    assert bool((dattn_long[~long_rows].float() == sentinel).all()), "dattn rows of one-tile samples / foreign rows were written"
    # ---- the three launches
    ref_dx = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
    ref_dw = torch.zeros(d, dtype=torch.float32, device="cuda")
    L.check(lib.gget_op_rmsnorm_bwd(P(dxn), P(x_mid), P(nw), P(rstd), P(dres), P(ref_dx), P(ref_dw), rows, d, ST()))
    torch.cuda.synchronize()
    assert torch.equal(dx_mid[live], ref_dx[live]), "dx_mid differs from rmsnorm_bwd_kernel's"
    if varlen:
        tail = torch.arange(rows, device="cuda") >= int(cu[-1])
        assert bool((dx_mid[tail].float() == 0).all()), "the pad rows behind the last sample must get a zero gradient"
        assert bool((dx_mid[~live & ~tail].float() == sentinel).all()) if bool((~live & ~tail).any()) else True
    else:
        assert bool(live.all())
    # norm weight gradient: the separate kernel summed every row of the buffer, the fused one the samples' rows
    xh = x_mid.float() * rstd[:, None]
    want_dw = (dxn.float() * xh)[live].sum(0)
    got_dw = dw.view(copies, cstride)[:, :d].sum(0)
    assert rel_l2(got_dw.cpu().numpy(), want_dw.cpu().numpy()) < 1e-4
    dattn = (ref_dx.float() @ wo.float()).to(torch.bfloat16)            # [rows, d]: dgrad of y = a Wo^T
    if bool(long_rows.any()):
        assert rel_l2(dattn_long[long_rows].float().cpu().numpy(), dattn[long_rows].float().cpu().numpy()) < 3e-3
    dattn_p = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    for b in range(B):
        dattn_p[b * S: b * S + nrow[b]] = dattn[r0[b]: r0[b] + nrow[b]]
    ref_dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    delta = torch.zeros(B * H * S, dtype=torch.float32, device="cuda")
    # (the separate op rotates with q / k UN-rotated in memory when tables are given; the engine's layout - rotated q / k, gradients
    #  rotated back - is what the fused kernel implements, so the reference here runs without tables and the rotation is undone by hand)
    L.check(lib.gget_op_attn_bwd(P(qkv_p), P(out_p), P(dattn_p), P(lse), P(lens_d), P(ref_dqkv), P(delta), B, S, H, causal, None, None, None,
                                 p, seed, ST()))
    torch.cuda.synchronize()
    for b in range(B):
        n = nrow[b]
        got = dqkv[r0[b]: r0[b] + n].float().view(n, 3, H, 64)
        want = ref_dqkv[b * S: b * S + n].float().view(n, 3, H, 64).clone()
        if rope:       # gradient of a rotated row back to the un-rotated one: x = R(-theta) x'
            for i in (0, 1):
                want[:, i] = _rope_ref(want[None, :, i], -pos[b: b + 1, :n])[0]      # (rotation by the negative angle)
        tol = 2e-2 if rope else 0.0      # (the un-rotation happens on fp32 accumulators in the kernel, on bf16 values here)
        if rope:
            assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < tol, f"sample {b}"
        else:
            err = (got - want).abs().max().item()
            # dO reaches the MFMAs as bf16 in both forms; the fused one rounds its own fp32 dattn, the reference the matmul above: equal up to
            # rounding flips of single dattn elements
            assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 4e-3, f"sample {b}: {err}"
    assert bool((dqkv[~live].float() == sentinel).all())
