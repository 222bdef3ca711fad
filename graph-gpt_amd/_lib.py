"""ctypes binding of libgget_hip.so (C ABI: include/gget.h).  There is NO CPU fallback: if the
library is missing or does not load, importing the engine raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GGET_LIB_PATH") or os.path.join(HERE, "lib", "libgget_hip.so")   # override: A/B builds in one run

i32, i64, u64, f32, vp, cp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p, C.c_char_p


class GgetConfig(C.Structure):
    _fields_ = [(n, i32) for n in ("kind", "vocab_size", "hidden_size", "intermediate_size", "num_layers", "num_heads",
                                   "stacked_feat", "next_n_token", "gated_agg", "causal", "max_position", "num_labels",
                                   "score_bias", "pad_token_id")] + \
               [("rms_eps", f32), ("rope_theta", f32), ("layer_scale_init", f32), ("max_tokens", i32), ("max_batch", i32),
                ("path_pdrop", f32), ("mlp_pdrop", f32), ("head_mlp_layers", i32), ("head_mlp", i32 * 4), ("embed_dim", i32)]


class GgetSizes(C.Structure):
    _fields_ = [(n, u64) for n in ("n_params", "param_bf16_bytes", "master_bytes", "adam_bytes", "grad_bf16_bytes",
                                   "workspace_bytes")]


class GgetBuffers(C.Structure):
    _fields_ = [(n, vp) for n in ("param_bf16_dev", "master_dev", "adam_m_dev", "adam_v_dev", "grad_bf16_dev",
                                  "workspace_dev", "rope_cos_dev", "rope_sin_dev")]


class GgetParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ndim", i32), ("shape", i64 * 2), ("offset", u64), ("layer", i32)]


# name -> (restype, argtypes); every symbol declared in include/gget.h
SIGNATURES = {
    "gget_last_error": (cp, []),
    "gget_version": (i32, []),
    "gget_query_sizes": (i32, [C.POINTER(GgetConfig), C.POINTER(GgetSizes)]),
    "gget_create": (i32, [C.POINTER(GgetConfig), C.POINTER(GgetBuffers), C.POINTER(vp)]),
    "gget_destroy": (i32, [vp]),
    "gget_param_count": (i32, [vp]),
    "gget_param_info": (i32, [vp, i32, C.POINTER(GgetParamInfo)]),
    "gget_bucket_count": (i32, [vp]),
    "gget_bucket_range": (i32, [vp, i32, C.POINTER(u64), C.POINTER(u64)]),
    "gget_sync_params": (i32, [vp, vp]),
    "gget_forward_pretrain": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "gget_forward_pretrain_packed": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "gget_forward_task": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "gget_backward": (i32, [vp, f32, vp]),
    "gget_backward_begin": (i32, [vp, f32, vp]),
    "gget_backward_layer": (i32, [vp, i32, vp]),
    "gget_backward_end": (i32, [vp, vp]),
    "gget_adamw_step": (i32, [vp, f32, f32, f32, f32, f32, f32, f32, i32, vp, vp]),
    "gget_comm_unique_id": (i32, [vp]),
    "gget_comm_init": (i32, [vp, i32, i32, vp]),
    "gget_comm_destroy": (i32, [vp]),
    "gget_comm_move": (i32, [vp, vp]),
    "gget_allreduce_grads_async": (i32, [vp, i32, i32, vp]),
    "gget_allreduce_range_async": (i32, [vp, u64, u64, i32, vp]),
    "gget_comm_init_loopback": (i32, [vp, i32]),
    "gget_head_counts": (i32, [vp, C.POINTER(i32 * 2), vp]),
    "gget_head_logits": (i32, [vp, C.POINTER(vp), C.POINTER(i32)]),
    "gget_hidden_states": (i32, [vp, C.POINTER(vp)]),
    "gget_layer_hidden_states": (i32, [vp, i32, C.POINTER(vp)]),
    "gget_hidden_states_grid": (i32, [vp, i32, vp, vp]),
    "gget_op_gemm": (i32, [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "gget_op_gemm_streamk": (i32, [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "gget_op_gemm_streamk_bytes": (u64, []),
    "gget_debug_set": (i32, [i32, i32]),
    "gget_debug_occupy": (i32, [vp, u64, i32, i32, i32, vp]),
    "gget_op_gemm_grouped": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gget_op_rmsnorm_fwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "gget_op_rmsnorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "gget_op_embed_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "gget_op_embed_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "gget_op_qkv_rope": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "gget_op_smtp2d": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, f32, f32, i32, i32, C.c_uint32, vp]),
    "gget_op_token_sample": (i32, [vp, i32, i32, i32, i32, f32, f32, i32, f32, C.c_uint32, vp, vp, vp]),
    "gget_op_unmask_origin": (i32, [vp, vp, i32, i32, f32, C.c_uint32, i32, vp]),
    "gget_op_token_confidence": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "gget_op_smtp_rows": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, C.c_double, C.c_double, C.c_double, C.c_uint32, vp]),
    "gget_op_rope": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "gget_op_attn_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, f32, C.c_uint32, vp]),
    "gget_op_attn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, f32, C.c_uint32, vp]),
    "gget_op_copy_from_host": (i32, [vp, vp, C.c_uint64, vp]),
    "gget_op_attn_fwd_varlen": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, f32, C.c_uint32, vp]),
    "gget_op_attn_bwd_varlen": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, f32, C.c_uint32, vp]),
    "gget_op_attn_oproj_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, C.c_uint32, vp, vp]),
    "gget_op_pack_wo": (i32, [vp, C.c_uint64, vp, vp, i32, i32, vp]),
    "gget_op_attn_oproj_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, C.c_uint64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, f32,
                                      C.c_uint32, i32, vp, vp, vp]),
    "gget_op_attn_fwd_ranges": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint32, vp]),
    "gget_op_attn_bwd_ranges": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint32, vp]),
    "gget_op_ranges_from_mask3d": (i32, [vp, vp, vp, i32, i32, vp]),
    "gget_op_attn_bwd_fused": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint32, vp]),
    "gget_set_dropout": (i32, [vp, f32, f32, C.c_uint32]),
    "gget_set_auc": (i32, [vp, i32, C.c_uint32]),
    "gget_set_token_count": (i32, [vp, C.c_int64]),
    "gget_set_option": (i32, [vp, i32, i32]),
    "gget_varlen_status": (i32, [vp, vp, vp]),
    "gget_position_status": (i32, [vp, vp, vp]),
    "gget_deferred_status": (i32, [vp, vp, vp]),
    "gget_set_focal_gamma": (i32, [vp, f32]),
    "gget_set_stack_method": (i32, [vp, i32]),
    "gget_set_rope_range": (i32, [vp, f32]),
    "gget_set_raw_embeds": (i32, [vp, vp, i32]),
    "gget_debug_probe": (i32, [vp, i32, C.POINTER(f32)]),
    "gget_debug_gemm_probe": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(i32)]),
    "gget_set_dropout_ex": (i32, [vp, f32, f32, f32]),
    "gget_op_gateup_geglu": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "gget_op_down_dgrad_geglu": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "gget_op_geglu_fwd": (i32, [vp, vp, i32, i32, vp]),
    "gget_op_geglu_bwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "gget_op_ce_fwd_bwd": (i32, [vp, i32, vp, vp, vp, i32, i32, vp, vp, f32, i32, vp]),
}

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
OPT_NORM_FROM_BACKWARD = 1   # gget_set_option
OPT_SKIP_NONFINITE_STEP = 2
TOKENS_AUTO = -2   # gget_set_token_count: count the real tokens on the device (include/gget.h GGET_TOKENS_AUTO)
EPI_NONE, EPI_RESIDUAL, EPI_ATOMIC_F32, EPI_SLAB_F32 = 0, 1, 2, 3


def gemm_grouped(lib, mode, problems, stream):
    """problems: list of (A, B, C, M, N, K, lda, ldb, ldc) with torch tensors; one persistent launch."""
    n = len(problems)
    PA, IA = vp * n, i32 * n
    ptr = lambda k: PA(*[p[k].data_ptr() for p in problems])
    num = lambda k: IA(*[int(p[k]) for p in problems])
    return lib.gget_op_gemm_grouped(mode, n, ptr(0), ptr(1), ptr(2), num(3), num(4), num(5), num(6), num(7), num(8), stream)
PROBLEM_SINGLE_LABEL, PROBLEM_REGRESSION_L1, PROBLEM_REGRESSION_MSE, PROBLEM_MULTI_LABEL, PROBLEM_AUC = 0, 1, 2, 3, 4
PROBLEM_TOKEN_CE = 5   # loss_type = "token_ce": score + cross-entropy on every labelled row, logits [B,S,C]

_lib = None


class GgetError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """dlopen the engine.  Raises (never falls back) when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # The HIP runtime must be the one PyTorch ships (torch/lib/libamdhip64.so): torch owns the device memory and the
    # streams we are handed, and a second runtime loaded first does not see the GPU on the pool's boxes.  Importing
    # torch first makes our DT_NEEDED libamdhip64 resolve to the already-loaded copy.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise GgetError(f"{path} is missing: build it with `python graph-gpt_amd/build.py` "
                        "(or __graft_entry__.build()); this engine has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise GgetError(f"gget error {rc}: {load().gget_last_error().decode()}")
