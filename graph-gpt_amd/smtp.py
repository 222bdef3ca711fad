"""In-model SMTP masking on the device and the bit-exact Python twin of its draws.

`smtp2d_mask` calls the HIP kernel behind `gget_op_smtp2d` (reference prepare_for_2d_smtp_inputs_labels,
src/models/graphgpt/modeling_helpers.py:399-468).  `draws` regenerates, with numpy integer arithmetic, exactly the random
numbers the kernel uses, in the shape the oracle (`oracle.gget_oracle.smtp_2d_inputs_labels`) takes them - that is how
the parity tests compare kernel and oracle cell by cell."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

_M32 = np.uint64(0xFFFFFFFF)


def _rng24(seed: int, stream: int, a, b):
    """24-bit counter hash, identical to `smtp_rng` in csrc/kernels.hip (32-bit wrap-around arithmetic)."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    x = (np.uint64(seed & 0xFFFFFFFF) ^ (np.uint64(stream) * np.uint64(0x9E3779B1) & _M32))
    x = (x + ((a * np.uint64(0x85EBCA77)) & _M32) + ((b * np.uint64(0xC2B2AE3D)) & _M32)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return (x >> np.uint64(8)).astype(np.int64)


def draws(seed: int, B: int, S: int, F: int):
    """(u_sample [B], u_rate [B], u_cell [B,S,F], token_shift [B,S,F], u_replace [B,S,F]) as torch tensors; u_cell is
    indexed by (sample, NODE index, feature) like the reference's `mask_per_node`."""
    inv = np.float32(1.0 / 16777216.0)
    b = np.arange(B)
    u_sample = _rng24(seed, 0, b, 0).astype(np.float32) * inv
    u_rate = _rng24(seed, 1, b, 0).astype(np.float32) * inv
    cell = np.arange(S * F)[None, :]
    u_cell = (_rng24(seed, 2, b[:, None], cell).astype(np.float32) * inv).reshape(B, S, F)
    u_rep = (_rng24(seed, 3, b[:, None], cell).astype(np.float32) * inv).reshape(B, S, F)
    s12 = np.zeros((B, S * F), np.int64)
    for k in range(12):
        s12 += _rng24(seed, 16 + k, b[:, None], cell)
    num = 10 * s12 - 60 * 16777216
    q = num >> 24
    r = num & 16777215
    q = q + ((r > 8388608) | ((r == 8388608) & ((q & 1) == 1)))
    shift = q.reshape(B, S, F).astype(np.float32)          # integer-valued: the oracle's round() is then exact
    T = torch.from_numpy
    return T(u_sample), T(u_rate), T(u_cell), T(shift), T(u_rep)


def smtp2d_mask(input_ids: torch.Tensor, node_idx: torch.Tensor, stacked_feat: int, *, smtp_2d_rate: float = 1.0,
                power: float = 1.0, replace_rate: float = 0.0, vocab: int, global_2d_mask: bool = False, seed: int = 0):
    """input_ids int64 [B,S,>=F] on the GPU (only the first F columns are read), node_idx int64 [B,S] (any stride, e.g. a
    column view of the same tensor).  Returns (masked ids [B,S,F], labels [B,S,F])."""
    if not input_ids.is_cuda:
        raise L.GgetError("smtp2d_mask runs on the GPU only (the CPU statement of it is test infrastructure under oracle/)")
    lib = L.load()
    assert input_ids.dtype == torch.int64 and node_idx.dtype == torch.int64
    B, S = input_ids.shape[:2]
    assert input_ids.stride(2) == 1 and input_ids.stride(0) == S * input_ids.stride(1)
    assert node_idx.shape == (B, S) and node_idx.stride(0) == S * node_idx.stride(1)
    out = torch.empty(B, S, stacked_feat, dtype=torch.int64, device=input_ids.device)
    lab = torch.empty_like(out)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.gget_op_smtp2d(C.c_void_p(input_ids.data_ptr()), int(input_ids.stride(1)), C.c_void_p(node_idx.data_ptr()),
                               int(node_idx.stride(1)), C.c_void_p(out.data_ptr()), C.c_void_p(lab.data_ptr()), B, S,
                               stacked_feat, float(smtp_2d_rate), float(power), float(replace_rate), int(vocab),
                               int(bool(global_2d_mask)), int(seed) & 0xFFFFFFFF, st))
    return out, lab


def row_mask_selection(seed: int, lengths, S: int, F: int, umr_min: float = 0.01, umr_max: float = 0.99, power: float = 1.0):
    """Twin of `smtp_rows_kernel`: per sample (chosen cell indices sorted by key, mask ratio alpha, dLM weight)."""
    out = []
    for b, ln in enumerate(np.asarray(lengths).tolist()):
        n = int(min(max(ln, 0), S)) * F
        r = float(_rng24(seed, 9, b, 0)) / 16777216.0
        t = umr_min + (umr_max - umr_min) * r
        alpha = 1.0 - t ** power
        k = int(np.ceil(n * alpha))
        cells = np.arange(n)
        keys = (_rng24(seed, 8, b, cells).astype(np.int64) << 20) | cells
        order = np.argsort(keys, kind="stable")[:k]
        out.append((order, alpha, power / t))
    return out


def smtp_mask_rows(input_ids: torch.Tensor, lengths: torch.Tensor, *, umr_min: float = 0.01, umr_max: float = 0.99,
                   power: float = 1.0, seed: int = 0, dlm_wgt: bool = False):
    """Device-side collator masking of a right-padded batch ids [B,S,F] (int64, GPU) with lengths [B] (int32, GPU):
    returns (masked ids, labels[, wgt])."""
    if not input_ids.is_cuda:
        raise L.GgetError("smtp_mask_rows runs on the GPU only (its CPU statement is test infrastructure under oracle/)")
    lib = L.load()
    B, S, F = input_ids.shape
    ids = input_ids.contiguous()
    lens = lengths.to(device=ids.device, dtype=torch.int32).contiguous()
    out, lab = torch.empty_like(ids), torch.empty_like(ids)
    wgt = torch.empty(B, dtype=torch.float32, device=ids.device) if dlm_wgt else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.gget_op_smtp_rows(C.c_void_p(ids.data_ptr()), C.c_void_p(lens.data_ptr()), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(lab.data_ptr()), C.c_void_p(wgt.data_ptr()) if wgt is not None else None, B, S, F,
                                  float(umr_min), float(umr_max), float(power), int(seed) & 0xFFFFFFFF, st))
    return (out, lab, wgt) if dlm_wgt else (out, lab)
