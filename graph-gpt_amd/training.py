"""Training-side surface for the hot path: the engine object the reference gets from
`deepspeed.initialize` (SURVEY.md 8b "Engine protocol": .module/.backward/.step/.save_checkpoint/
.load_checkpoint/.global_steps/.device), the per-step loop of `batch_training` / `ft_batch_training`
(reference src/utils/training_utils.py:7-95, :98-205), the LR schedules (loss_utils.py:322-367,
ds_config2_pt.json:20-28) and a lean `TrainingPipeline` / `TrainingMode` (reference
src/training/pipeline.py:60-95, mode.py:46-89) that drives them from any iterable of collated batches.

Data parallelism: one process per GPU; gradients live in ONE flat bf16 array cut into
L+2 buckets in the order backward completes them; each bucket is all-reduced (RCCL, sum) on a side HIP
stream as soon as its backward stage is enqueued, so communication overlaps the remaining backward;
1/world is folded into the fused AdamW.  No DeepSpeed, no DDP hooks.
"""
from __future__ import annotations

import abc
import math
import os
import time
from typing import Any, Callable, Dict, Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .modeling import GraphGPTPretrainBase, GraphGPTTaskModel, _GgetModel


# ----------------------------------------------------------------------------- LR schedules
def one_cycle_lr(step: int, max_lr: float, total_steps: int, pct_start: float, min_lr: float = 0.0) -> float:
    """torch OneCycleLR(anneal='cos', div_factor=25) exactly as `_py_one_cycle` configures it
    (reference src/utils/loss_utils.py:322-367); `step` is the 0-based scheduler step."""
    initial = max_lr / 25.0
    final = min_lr if min_lr > 0 else initial / 1e4
    up_end = float(pct_start * total_steps) - 1
    down_end = total_steps - 1

    def cos_anneal(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    if step <= up_end:
        return cos_anneal(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos_anneal(max_lr, final, (step - up_end) / (down_end - up_end))


def warmup_decay_lr(step: int, max_lr: float, min_lr: float, warmup: int, total: int) -> float:
    """DeepSpeed WarmupDecayLR as named by examples/ds_config2_pt.json:20-28 (log warm-up, linear decay)."""
    if step < warmup:
        gamma = math.log(step + 1) / math.log(max(2, warmup))
    else:
        gamma = max(0.0, (total - step) / max(1.0, total - warmup))
    return min_lr + (max_lr - min_lr) * gamma


class OptimConfig:
    """Defaults = the reference pre-train launch script (examples/graph_lvl/pcqm4m_v2_pretrain.sh:53-57)."""

    def __init__(self, lr=3e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0, min_lr=0.0,
                 warmup_num_steps=0, total_num_steps=0, schedule="constant", onecycle_extra_step=1, gradient_accumulation_steps=1):
        # micro-batches per optimizer step (the DeepSpeed engine's `gradient_accumulation_steps`, conf_utils.py:59-66): see GgetEngine.step
        self.gradient_accumulation_steps = max(1, int(gradient_accumulation_steps))
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.max_grad_norm, self.min_lr = max_grad_norm, min_lr
        self.warmup_num_steps, self.total_num_steps, self.schedule = warmup_num_steps, total_num_steps, schedule
        # OneCycleLR's total_steps: total_num_steps + 1 on the reference's DDP path (opt_utils.py:30, "to avoid error of
        # lr_scheduler.step() in last step"), total_num_steps on its DeepSpeed fine-tune path (conf_utils.py:124)
        self.onecycle_extra_step = int(onecycle_extra_step)

    def lr_at(self, step: int) -> float:
        if self.schedule == "onecycle" and self.total_num_steps > 0:
            total = self.total_num_steps + self.onecycle_extra_step
            return one_cycle_lr(min(step, total - 1), self.lr, total,
                                self.warmup_num_steps / max(1, self.total_num_steps), self.min_lr)
        if self.schedule == "warmup_decay" and self.total_num_steps > 0:
            return warmup_decay_lr(step, self.lr, self.min_lr, self.warmup_num_steps, self.total_num_steps)
        return self.lr


# ----------------------------------------------------------------------------- DP exchange step
class _Fp32Reduce:
    """Handle of an fp32-accumulated bucket reduction: wait() finishes the collective and rounds the sum back into the bf16
    gradient slice (one rounding instead of the world-1 a bf16 ring sum applies)."""

    def __init__(self, work, wide, dst):
        self.work, self.wide, self.dst = work, wide, dst

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.dst.copy_(self.wide)


def all_reduce_bucket(flat: torch.Tensor, bucket, group=None, async_op: bool = True, fp32_accumulate: bool = False):
    """Sum-all-reduce ONE gradient bucket (a contiguous [offset, offset+count) slice of the flat gradient array).
    Device-agnostic: RCCL on GPU tensors, gloo on CPU tensors (the CPU tests drive exactly this function).
    fp32_accumulate: widen the slice to fp32 for the reduction (twice the wire bytes, a single final rounding)."""
    off, cnt = bucket
    sl = flat[off: off + cnt]
    if not fp32_accumulate:
        return dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    wide = sl.to(torch.float32)
    work = dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    h = _Fp32Reduce(work if async_op else None, wide, sl)
    if not async_op:
        h.wait()
        return None
    return h


def shard_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed: ranks draw independent batches (reference misc_utils.py:536-538 seeds with
    `initial_seed - rank`; the synthetic generator uses base + rank)."""
    return int(base_seed) + int(rank)


def pretrain_rank_sampler(sample_idx, epochs: int, seed: int, rank: int):
    """Pre-training partition rule (reference get_pt_train_valid_test_sampler loader_utils.py:328-333, reset_pt_train_sampler
    :412-442, seeding misc_utils.py:536-538): every rank keeps the FULL index list repeated `epochs` times and shuffles it
    with its own generator seeded `seed - rank` - ranks draw independently, not disjointly (the token budget, not the
    epoch, bounds the run).  Python's `random` module like the reference, so the order is the reference's order."""
    import random
    idx = [int(i) for i in sample_idx] * max(1, int(epochs))
    random.Random(int(seed) - int(rank)).shuffle(idx)
    return idx


def finetune_rank_sampler(sample_idx, world_size: int, rank: int, seed: int, epoch: int = 0):
    """Fine-tune partition rule (reference distribute_sampler_with_rnd_seed loader_utils.py:78-90, called with
    seed = finetune.seed + epoch at :622-627): one permutation per epoch shared by all ranks, truncated to a multiple of the
    world size, rank r takes positions r, r + world, ...  -> disjoint shards of equal length that change every epoch."""
    sample_idx = torch.as_tensor(sample_idx)
    g = torch.Generator()
    g.manual_seed(int(seed) + int(epoch))
    indices = torch.randperm(len(sample_idx), generator=g).tolist()
    total = (len(sample_idx) // world_size) * world_size
    return sample_idx[indices[rank:total:world_size]].tolist()


def eval_rank_sampler(sample_idx, world_size: int, rank: int, shuffle_seed: Optional[int] = None):
    """Evaluation partition rule (reference distribute_sampler loader_utils.py:70-75, used for the valid / test samplers at
    :256, :270): indices sorted, rank r keeps those at sorted positions i with i % world == r (every sample exactly once
    across ranks, shard sizes differ by at most one), then shuffled locally (order is irrelevant to the metrics)."""
    import random
    vec = sorted(int(i) for i in sample_idx)
    out = [vec[i] for i in range(len(vec)) if i % world_size == rank]
    if shuffle_seed is not None:
        random.Random(shuffle_seed).shuffle(out)
    return out


def schedule_steps(total_tokens: float, tokens_per_sample: float, batch_size: int, world_size: int) -> int:
    """Optimizer steps of a token-budgeted run (reference base_configs.py:54-60): the global batch is world * batch_size."""
    return int(total_tokens // (tokens_per_sample * batch_size * world_size))


# ----------------------------------------------------------------------------- engine protocol
class GgetEngine:
    """What `deepspeed.initialize(model=...)` returns in the reference (pretrain_mode.py:281-287), rebuilt on
    the HIP engine: forward via `engine(...)`, `engine.backward(loss)`, `engine.step()`."""

    def __init__(self, model: _GgetModel, optim: Optional[OptimConfig] = None, process_group=None):
        self.module = model
        self.optim = optim or OptimConfig()
        self.global_steps = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self._comm_stream = None
        self._pending = []
        self.force_staged = bool(int(os.environ.get("GGET_FORCE_STAGED", "0")))  # run the bucketed path at world 1
        # GGET_DP_OVERLAP=0: one all-reduce of the whole flat gradient array after the monolithic backward instead of the
        # bucketed exchange overlapped with it (to be decided by measurement on a multi-GPU node)
        self.overlap = bool(int(os.environ.get("GGET_DP_OVERLAP", "1")))
        # GGET_DP_FP32_REDUCE=1: reduce every bucket in fp32 (see all_reduce_bucket).  GGET_DP_BACKEND=abi: issue the
        # collectives through the C ABI (gget_comm_init / gget_allreduce_grads_async = RCCL on a HIP side stream, no
        # torch.distributed on the data path; the unique id travels once over the existing process group).
        self.fp32_reduce = bool(int(os.environ.get("GGET_DP_FP32_REDUCE", "0")))
        self.abi_comm = os.environ.get("GGET_DP_BACKEND", "torch") == "abi"
        # GGET_DP_LOOPBACK_WORLD=W (with GGET_DP_BACKEND=abi, single process): the C-ABI exchange runs as rank 0 of W ranks that all hold
        # this rank's gradients (gget_comm_init_loopback) - the schedule of a W-rank job (bucket ranges, side-stream waits, 1/W folded
        # into AdamW) on a one-GPU box; the step must equal the single-rank step
        self.loopback_world = int(os.environ.get("GGET_DP_LOOPBACK_WORLD", "0")) if self.abi_comm else 0
        if self.loopback_world > 0:
            assert self.world == 1, "the loopback communicator replaces the process group: run it in a single process"
            self.world = self.loopback_world
            self.force_staged = True
        # GGET_DP_BUCKET_MB=N: consecutive buckets (completion order) are exchanged in ONE collective once they add up to >= N MiB -
        # fewer, larger messages (14 buckets of ~19 MB for the base model; 60 -> 4-5 collectives).  0 (default) = one per bucket.
        self.bucket_mb = float(os.environ.get("GGET_DP_BUCKET_MB", "0"))
        self._groups = None
        # measurement switch (bench.py `dp.exposed_comm_ms`): False runs the same staged backward WITHOUT issuing the collectives -
        # the ranks then drift apart, so it is only ever set for a few untimed-for-throughput diagnostic steps
        self.exchange = True
        self.reserved_cus = 0
        self._dp_menu_set = False      # this engine switched the process-wide GEMM launch menu (gget_debug_set keys 2 / 13 / 15): close() restores it
        # gradient accumulation (DeepSpeed branch, conf_utils.py:59-66 -> the DS engine steps at the boundary only): the micro-batches'
        # gradients are summed in an fp32 copy of the flat gradient array; step() k - 1 times out of k only does that
        self.micro_steps = 0
        self._grad_acc = None
        # GradScaler's rule of the reference's DDP branch (training_utils.py:46-86): an optimizer step whose gradient norm is inf / NaN is
        # skipped (weights and Adam state untouched, Adam's step count not advanced) while the LR schedule still advances
        self.skip_nonfinite = False
        self.skipped_steps = 0
        if self.world > 1 and torch.cuda.is_available():
            # a collective's kernel shares the chip with the compute stream from now on: the GEMM launcher keeps LDS headroom on every
            # CU (no launch with two LDS-filling workgroups per CU; gget_debug_set key 2)
            from . import _lib as L
            # (rounds 2 - 4 set this for every multi-rank job; round 5's stand-in with RCCL's real register footprint - tools/dp_standin.py -
            #  shows the rule buys nothing against such a kernel and costs 0.08 ms alone, 0.2 ms beside it: opt-in now, GGET_DP_LDS_HEADROOM=1)
            if bool(int(os.environ.get("GGET_DP_LDS_HEADROOM", "0"))):
                L.check(L.load().gget_debug_set(2, 2))
                self._dp_menu_set = True
            # (superseded below, in a real multi-process job, by the stronger rule: CUs of their own for the collective - the two-per-CU launch
            #  returns then, and the per-sample backward runs whenever its grid fits the CUs that are left)
            # ... and, in a real multi-process job that asks for it, the GEMM launches leave GGET_DP_RESERVE_CUS CUs (default 0 = off) FREE for the
            # collective's workgroups, which are held to as many channels (NCCL_MAX_NCHANNELS, unless the user set it): an RCCL workgroup
            # (256 threads x 261 - 280 registers, 19.7 KiB LDS) cannot share a CU with any 8-wave GEMM workgroup, and a GEMM launch that finds
            # one of "its" CUs taken runs a second round (csrc/gemm.hip g_gemm_cu_reserve).  The RMSNorm backward goes
            # back to its many-small-blocks form for the same reason.
            real_world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
            if real_world > 1:
                self.reserved_cus = max(0, int(os.environ.get("GGET_DP_RESERVE_CUS", "0")))
                if self.reserved_cus:
                    # (NCCL_MAX_NCHANNELS is read when the communicator is created - before this constructor runs: dp_env_defaults()
                    #  sets it, from the same variable, ahead of init_process_group; bench.py and set_dist_env call it)
                    L.check(L.load().gget_debug_set(15, self.reserved_cus))
                    L.check(L.load().gget_debug_set(13, 0))
                    L.check(L.load().gget_debug_set(2, 1))
                    self._dp_menu_set = True
        model.materialize_grads = False  # fused path: gradients stay in the flat bf16 arena
        model._managed_by_engine = True  # the bucketed exchange below replaces the all-reduce of _autograd_backward

    def close(self):
        """Give back what this engine changed process-wide: the data-parallel GEMM launch menu (keys 2 / 13 / 15 of gget_debug_set are
        globals of the library, not of a handle - ADVICE r5) goes back to the single-GPU defaults.  Idempotent; also run when the object dies."""
        if getattr(self, "_dp_menu_set", False):
            self._dp_menu_set = False
            try:
                from . import _lib as L
                lib = L.load()
                lib.gget_debug_set(15, 0)
                lib.gget_debug_set(13, 1)
                lib.gget_debug_set(2, 0)
            except Exception:
                pass

    def __del__(self):
        self.close()

    @property
    def device(self):
        return self.module.device

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def _ensure_abi_comm(self, e):
        # readiness belongs to the ENGINE INSTANCE: a model that re-creates its engine for a larger batch hands the
        # communicator over (Engine.comm_adopt), so this collective bootstrap runs once per job, on every rank together
        if e.comm_world > 0:
            return
        if self.loopback_world > 0:
            e.comm_init_loopback(self.loopback_world)
            return
        rank = dist.get_rank(self.pg) if self.world > 1 else 0
        uid = [e.comm_unique_id() if rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(uid, src=0, group=self.pg)
        e.comm_init(rank, self.world, uid[0])

    # -- backward with bucketed all-reduce overlapped on a side stream
    def backward(self, loss=None):
        e = self.module._engine
        if self.abi_comm:
            self._ensure_abi_comm(e)
        # single-rank step: nothing touches the gradient array between this backward and AdamW - the engine MAY take the layers' share
        # of the gradient norm from its weight-gradient launches (include/gget.h GGET_OPT_NORM_FROM_BACKWARD).  Opt-in
        # (GGET_NORM_FROM_BACKWARD=1): measured in the step it saves its 31 us of norm pass and loses them again in AdamW, whose
        # gradient reads the full pass had warmed the memory-side cache for (7.095 against 7.093 ms, profiles/r04_step_experiments.txt)
        fold = self.world == 1 and not self.force_staged and bool(int(os.environ.get("GGET_NORM_FROM_BACKWARD", "0") or 0))
        if getattr(e, "_norm_fold", None) != fold:
            from . import _lib as L
            e.set_option(L.OPT_NORM_FROM_BACKWARD, int(fold))
            e._norm_fold = fold
        if self.world == 1 and not self.force_staged:
            e.backward()
            return
        if not self.overlap:
            e.backward()
            if not self.exchange:
                return
            if self.abi_comm:
                e.allreduce_grads_async(-1, self.fp32_reduce)
            elif self.world > 1:
                all_reduce_bucket(e.grad_bf16, (0, e.grad_bf16.numel()), self.pg, async_op=False, fp32_accumulate=self.fp32_reduce)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=e.device)
        main = torch.cuda.current_stream()
        L_ = e.spec.num_layers

        groups = self.exchange_groups(e)

        def reduce_bucket(b):
            if b not in groups:          # a bucket inside a coalesced group: exchanged with the group's last bucket
                return
            off, cnt = groups[b]
            ev = torch.cuda.Event()
            ev.record(main)
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                if not self.exchange:
                    self._pending.append(None)
                elif self.abi_comm:   # RCCL through the C ABI on the side stream (works at world 1 too: a one-rank communicator)
                    e.allreduce_range_async(off, cnt, self.fp32_reduce, self._comm_stream)
                    self._pending.append(None)
                elif self.world > 1 or (self.force_staged and dist.is_available() and dist.is_initialized()):
                    # (a ONE-rank process group with GGET_FORCE_STAGED=1 still issues the collectives: the real backend - RCCL - runs the
                    #  whole exchange schedule on a one-GPU box, tests/test_gpu_dist.py::test_torch_rccl_one_rank_group_through_staged_backward)
                    self._pending.append(all_reduce_bucket(e.grad_bf16, (off, cnt), self.pg, async_op=True,
                                                           fp32_accumulate=self.fp32_reduce))
                else:  # single-rank dry run of the staged path (tests): the exchange is the identity
                    self._pending.append(None)

        e.backward_begin()
        reduce_bucket(0)
        for i in range(L_ - 1, -1, -1):
            e.backward_layer(i)
            reduce_bucket(L_ - i)
        e.backward_end()
        reduce_bucket(L_ + 1)

    def exchange_groups(self, e) -> Dict[int, Any]:
        """{last bucket of a group: (offset, count)} - what one collective covers.  Buckets are numbered in completion order and laid
        out back to front in the flat array, so consecutive buckets are adjacent ranges; a group is closed when it reaches
        GGET_DP_BUCKET_MB (or at the last bucket).  Non-adjacent neighbours (never the case for the engine's layout) close a group too."""
        if self._groups is not None and self._groups[0] is e:
            return self._groups[1]
        groups, lo, hi = {}, None, None
        thresh = self.bucket_mb * 2 ** 20 / 2        # elements (bf16)
        nb = len(e.buckets)
        for b, (off, cnt) in enumerate(e.buckets):
            if lo is not None and (off + cnt == lo or off == hi):
                lo, hi = min(lo, off), max(hi, off + cnt)
            else:
                if lo is not None:
                    groups[b - 1] = (lo, hi - lo)
                lo, hi = off, off + cnt
            if hi - lo >= thresh or b == nb - 1:
                groups[b] = (lo, hi - lo)
                lo = hi = None
        self._groups = (e, groups)
        return groups

    # -- the data-parallel launch menu, decided by measurement on the machine the job runs on
    def set_dp_menu(self, overlap: Optional[bool] = None, reserve_cus: Optional[int] = None):
        """Switch the exchange arrangement of a running engine: `overlap` = bucketed collectives beside the backward (True) or one
        collective behind it (False); `reserve_cus` = CUs the GEMM launch plans leave free for the collective's workgroups (the GEMM side
        of GGET_DP_RESERVE_CUS; the collective library's channel count is fixed when its communicator is created - dp_env_defaults)."""
        if overlap is not None:
            self.overlap = bool(overlap)
        if reserve_cus is not None and torch.cuda.is_available():
            from . import _lib as L
            lib = L.load()
            r = max(0, int(reserve_cus))
            L.check(lib.gget_debug_set(15, r))
            L.check(lib.gget_debug_set(13, 0 if r else 1))
            L.check(lib.gget_debug_set(2, 1 if r else (2 if bool(int(os.environ.get("GGET_DP_LDS_HEADROOM", "0"))) else 0)))
            self.reserved_cus = r
            self._dp_menu_set = self._dp_menu_set or r > 0

    def probe_dp_menu(self, step_fn: Callable[[], Any], steps: int = 10, warm: int = 2, menus=None) -> Dict[str, Any]:
        """No 1 -> 8 GPU curve of this engine has been measured: instead of a default chosen from a model, a
        multi-rank job TIMES the arrangements on its own machine at start-up and keeps the fastest - `steps` training steps each
        (after `warm`), max over ranks, the decision broadcast from rank 0 so that every rank takes the same.  `step_fn()` runs one
        step (forward + backward + step) on this rank.  The probe steps are ordinary training steps (the replicas stay identical).
        Returns {"menus": [...], "chosen": {...}}; a single-rank engine returns without measuring."""
        if self.world <= 1 or not (dist.is_available() and dist.is_initialized()):
            return {"menus": [], "chosen": {"overlap": self.overlap, "reserve_cus": self.reserved_cus}, "probed": False}
        if menus is None:
            menus = [dict(overlap=True, reserve_cus=0), dict(overlap=False, reserve_cus=0), dict(overlap=True, reserve_cus=32)]
        dev = self.module.device if dist.get_backend(self.pg) == "nccl" else torch.device("cpu")
        rows = []
        for m in menus:
            self.set_dp_menu(**m)
            for _ in range(warm):
                step_fn()
            torch.cuda.synchronize()
            dist.barrier(self.pg)
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            rows.append(dict(m, ms_per_step=float(t[0]) / steps * 1e3))
        best = min(range(len(rows)), key=lambda i: rows[i]["ms_per_step"])
        pick = torch.tensor([best], dtype=torch.int64, device=dev)
        dist.broadcast(pick, src=0, group=self.pg)
        chosen = menus[int(pick[0])]
        self.set_dp_menu(**chosen)
        return {"menus": rows, "chosen": dict(chosen), "probed": True, "steps_per_menu": steps}

    def describe_dp(self) -> Dict[str, Any]:
        """What the data-parallel exchange of this engine looks like (bench.py prints it on N > 1 lines)."""
        e = self.module._engine
        live = dist.is_available() and dist.is_initialized() and (self.world > 1 or self.force_staged)
        backend = dist.get_backend(self.pg) if live else "none"
        info = {"world": self.world, "backend": ("rccl-via-c-abi" if self.abi_comm else f"torch.distributed/{backend}"),
                "n_buckets": len(e.buckets) if e is not None else None, "overlap_with_backward": bool(self.overlap),
                "reduce_dtype": "fp32" if self.fp32_reduce else "bf16",
                "bucket_mb": [round(c * 2 / 2 ** 20, 1) for _, c in e.buckets] if e is not None else None,
                "collectives_per_step": len(self.exchange_groups(e)) if e is not None else None,
                "collective_mb": [round(c * 2 / 2 ** 20, 1) for _, c in self.exchange_groups(e).values()] if e is not None else None,
                "reserved_cus": self.reserved_cus, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}
        try:
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception:
            info["rccl_version"] = None
        return info

    def set_skip_nonfinite(self, on: bool):
        """The DDP branch's step rule (see __init__): the engine leaves weights / Adam state alone when the gradient norm is not finite
        (GGET_OPT_SKIP_NONFINITE_STEP); step() then takes Adam's step count back and counts the skipped step."""
        on = bool(on)
        e = self.module._engine
        if e is not None and getattr(e, "_skip_nonfinite", None) != on:
            from . import _lib as L
            e.set_option(L.OPT_SKIP_NONFINITE_STEP, int(on))
            e._skip_nonfinite = on
        self.skip_nonfinite = on

    def step(self):
        """One `engine.step()` of the reference's loop.  With `gradient_accumulation_steps = k > 1` (DeepSpeed branch) the call is made
        after every micro-batch like there, and like the DS engine only every k-th call updates the weights: the others add the
        micro-batch's (exchanged) gradient to an fp32 sum and return None; at the boundary the sum goes back into the gradient
        array and AdamW runs with 1 / (world * k) - the mean over the k * world micro-batches - clip and LR schedule once per update."""
        e = self.module._engine
        if self._pending:
            for w in self._pending:
                if w is not None:
                    w.wait()  # makes the current stream wait for the comm stream's collectives
            self._pending = []
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        o = self.optim
        k = getattr(o, "gradient_accumulation_steps", 1)
        if k > 1:
            g = e.grad_bf16
            if self._grad_acc is None or self._grad_acc.shape != g.shape:
                self._grad_acc = torch.zeros(g.shape, dtype=torch.float32, device=g.device)
                self.micro_steps = 0
            self._grad_acc += g
            self.micro_steps += 1
            if self.micro_steps % k != 0:
                return None
            g.copy_(self._grad_acc)
            self._grad_acc.zero_()
        lr = o.lr_at(self.global_steps)
        if self.skip_nonfinite and getattr(e, "_skip_nonfinite", None) is not True:
            self.set_skip_nonfinite(True)       # (the model re-created its engine: the option lives on the engine instance)
        gn = e.adamw_step(lr, o.betas[0], o.betas[1], o.eps, o.weight_decay, o.max_grad_norm, 1.0 / (self.world * k))
        self.global_steps += 1
        self.last_lr, self.last_grad_norm = lr, gn
        if self.skip_nonfinite and not bool(torch.isfinite(gn)):   # (GradScaler.step reads found_inf back as well: one sync per step on this branch)
            e.step_count -= 1                   # optimizer.step() did not run; lr_scheduler.step() did (global_steps stays advanced)
            self.skipped_steps += 1
        return gn

    # -- checkpoint = reference DDP layout (misc_utils.py:105-121): model.pt / optimizer.pt keyed by state-dict names
    def save_checkpoint(self, save_dir: str, tag: Optional[str] = None):
        d = os.path.join(save_dir, tag) if tag else save_dir
        os.makedirs(d, exist_ok=True)
        e = self.module._engine
        # the MODULE's state dict: reference shapes (emb_mask_token is [1,1,embed_dim] there, flat in the engine arena)
        torch.save({k: v.detach().cpu().clone() for k, v in self.module.state_dict().items()}, os.path.join(d, "model.pt"))
        torch.save({"m": {k: e.view(k, "m").cpu() for k in e.params}, "v": {k: e.view(k, "v").cpu() for k in e.params},
                    "step": e.step_count, "global_steps": self.global_steps}, os.path.join(d, "optimizer.pt"))
        self.module.config.save_pretrained(d)

    def load_checkpoint(self, load_dir: str, tag: Optional[str] = None):
        d = os.path.join(load_dir, tag) if tag else load_dir
        e = self.module._engine
        if e is None:
            e = self.module._ensure_engine(1, 8)
        e.load_state_dict(torch.load(os.path.join(d, "model.pt"), map_location="cpu"))
        opt_path = os.path.join(d, "optimizer.pt")
        if os.path.exists(opt_path):
            st = torch.load(opt_path, map_location="cpu")
            for k in e.params:
                e.view(k, "m").copy_(st["m"][k].to(e.device))
                e.view(k, "v").copy_(st["v"][k].to(e.device))
            e.step_count = int(st["step"])
            self.global_steps = int(st["global_steps"])
        return d, {}


def initialize(model: _GgetModel, optim: Optional[OptimConfig] = None, process_group=None) -> GgetEngine:
    return GgetEngine(model, optim, process_group)


# ----------------------------------------------------------------------------- one optimisation step
def _reference_step_args(model, train_cfg, train_stats):
    """The reference's positional form `batch_training(data, model, train_cfg, train_stats, opt_stats)` (training_utils.py:7-13):
    `model` is the engine `deepspeed.initialize` returned, `train_stats` carries `.device`, `.has_embeds_input`, `.use_deepspeed`."""
    return getattr(train_stats, "device", None) or model.device, bool(getattr(train_stats, "has_embeds_input", False))


def _reference_optimizer_step(model, train_cfg, train_stats, loss):
    """backward + optimizer step of the reference's two branches on the engine.
    DeepSpeed branch (training_utils.py:44-45): model.backward(loss); model.step().
    DDP branch (`train_stats.use_deepspeed` False, :46-86): zero_grad; fp16 autocast forward; scaler.scale(loss).backward();
    scaler.unscale_; clip_grad_norm_(max_grad_norm); scaler.step (SKIPPED when a gradient is inf / NaN); scaler.update; lr_scheduler.step().
    On this engine the forward / backward arithmetic is bf16 with fp32 accumulation whatever the branch (fp16 autocast is not reproduced:
    bf16 has fp32's exponent range, so the loss scale is 1 and never changes); what the branch changes is the step rule - GradScaler's
    skip of a non-finite step (GGET_OPT_SKIP_NONFINITE_STEP: weights and Adam state untouched, Adam's step count not advanced) while the
    LR schedule still advances, and `optimizer.gradient_accumulation_steps` must be 1 (the reference asserts it, :47-49)."""
    ddp = not getattr(train_stats, "use_deepspeed", True)
    if model.skip_nonfinite != ddp:
        model.set_skip_nonfinite(ddp)       # (the rule itself - Adam's step count taken back, the skipped-steps counter - lives in GgetEngine.step)
    if ddp:
        oc = getattr(train_cfg, "optimizer", None)
        assert oc is None or getattr(oc, "gradient_accumulation_steps", 1) == 1, \
            "https://pytorch.org/docs/stable/notes/amp_examples.html#gradient-accumulation"
    model.backward(loss)
    return model.step()


def batch_training(data: Dict[str, torch.Tensor], engine: GgetEngine, train_cfg=None, train_stats=None, opt_stats=None):
    """reference training_utils.batch_training DeepSpeed branch (:30-45): loss = head1 (+head2); backward; step.
    position_ids are NOT passed in pre-training (reference comments them out at :35).

    Two call forms.  `batch_training(data, engine)`: host tensors go to the engine as they are; `data["num_tokens"]` (optional host
    int = sum of the attention mask) or the host-side mask itself gives the var-len layout its row count for free.
    `batch_training(data, model, train_cfg, train_stats, opt_stats)` - the reference's own signature (training_utils.py:7-13): every
    tensor is moved to `train_stats.device` first, exactly like :17-26, the model sees device tensors only (the engine then counts
    the mask on the device), and the losses / shapes are recorded on `train_stats` like :87-95."""
    if train_stats is None:
        out = engine(input_ids=data["input_ids"], attention_mask=data["attention_mask"], labels=data["labels"],
                     inputs_raw_embeds=data.get("embed"), sample_wgt=data.get("wgt"), num_tokens=data.get("num_tokens"))
        loss = out.head1_loss
        if out.head2_loss is not None:
            loss = loss + out.head2_loss
        engine.backward(loss)
        engine.step()
        return loss
    model = engine
    device, has_embeds = _reference_step_args(model, train_cfg, train_stats)
    input_ids = data["input_ids"].to(device)
    attention_mask = data["attention_mask"].to(device)
    labels = data["labels"].to(device)
    inputs_raw_embeds = data["embed"].to(device) if has_embeds else None
    sample_wgt = data["wgt"].to(device) if "wgt" in data else None
    output = model(input_ids=input_ids, attention_mask=attention_mask, labels=labels, inputs_raw_embeds=inputs_raw_embeds,
                   sample_wgt=sample_wgt)
    main_loss, aux_loss = output.head1_loss, output.head2_loss
    loss = main_loss + aux_loss if aux_loss is not None else main_loss
    _reference_optimizer_step(model, train_cfg, train_stats, loss)
    train_stats.loss, train_stats.main_loss, train_stats.aux_loss = loss, main_loss, aux_loss
    train_stats.inputs_shape = input_ids.shape
    train_stats.sliced_raw_embeds = inputs_raw_embeds[:2, :8] if inputs_raw_embeds is not None else None
    return loss


def ft_batch_training(data: Dict[str, torch.Tensor], engine: GgetEngine, *ref_args, label_key: str = "task_labels"):
    """reference training_utils.ft_batch_training (:98-205): passes position_ids, task labels, sample weights.
    `ft_batch_training(data, engine[, label_key=])` or the reference's positional form
    `ft_batch_training(data, model, fthead_cfg, train_cfg, train_stats, opt_stats)` (tensors moved to the device first, :114-133;
    `fthead_cfg.task_type` names the label key, multi-label targets become float, the losses are recorded on `train_stats`)."""
    if not ref_args:
        out = engine(input_ids=data["input_ids"], attention_mask=data["attention_mask"], position_ids=data.get("position_ids"),
                     task_labels=data[label_key], cls_idx=data.get("cls_idx"), inputs_raw_embeds=data.get("embed"),
                     sample_wgt=data.get("wgt"), num_tokens=data.get("num_tokens"))
        loss = out.task_loss
        engine.backward(loss)
        engine.step()
        return loss, out.task_logits
    if len(ref_args) != 4:
        raise TypeError("ft_batch_training(data, model, fthead_cfg, train_cfg, train_stats, opt_stats)")
    fthead_cfg, train_cfg, train_stats, _opt_stats = ref_args
    model = engine
    device, has_embeds = _reference_step_args(model, train_cfg, train_stats)
    if getattr(getattr(train_cfg, "finetune", None), "use_aux", False):
        raise NotImplementedError("finetune.use_aux (the double-heads fine-tune model) is outside the hot-path scope")
    task_labels = data[f"{fthead_cfg.task_type}_labels"].to(device)
    if fthead_cfg.problem_type == "multi_label_classification":
        task_labels = task_labels.float()
    output = model(input_ids=data["input_ids"].to(device), attention_mask=data["attention_mask"].to(device), pretrain_labels=None,
                   task_labels=task_labels, cls_idx=data["cls_idx"].to(device) if "cls_idx" in data else None,
                   inputs_raw_embeds=data["embed"].to(device) if has_embeds else None,
                   sample_wgt=data["wgt"].to(device) if "wgt" in data else None, position_ids=data["position_ids"].to(device))
    task_loss = output.task_loss
    loss = task_loss.float()
    _reference_optimizer_step(model, train_cfg, train_stats, loss)
    train_stats.loss, train_stats.main_loss, train_stats.aux_loss = loss, task_loss, None
    return loss, output.task_logits


def _host_token_count(data) -> Optional[int]:
    """Real tokens of a collated batch when that is free to know: `data["num_tokens"]`, or the sum of a HOST-side 2-D attention mask (the
    evaluation loops receive CPU batches and move them: counting before the move costs nothing) - lets the engine run the batch on the
    padding-free token layout (modeling._GgetModel._token_count).  None = unknown (device-side mask): padded layout."""
    if data.get("num_tokens") is not None:
        return int(data["num_tokens"])
    am = data.get("attention_mask")
    if am is not None and am.dim() == 2 and am.device.type == "cpu":
        return int((am != 0).sum())
    return None


def _layout_kw(model, data) -> Dict[str, Any]:
    """`num_tokens=` for the engine-backed model classes only (the reference's call keeps its exact keyword set for any other model)."""
    n = _host_token_count(data) if isinstance(getattr(model, "module", model), _GgetModel) else None
    return {} if n is None else {"num_tokens": n}


# ----------------------------------------------------------------------------- batch hand-over (host -> device)
class DevicePrefetcher:
    """The reference's step moves every tensor of the batch to the device at its top (training_utils.py:17-26, `.to(device)` on pageable
    DataLoader output: a synchronous staging copy per tensor in front of the forward).  Here the host half of the hand-over of batch
    t + 1 runs DURING step t: the collated batch is copied into pinned staging buffers one batch ahead (three sets, used in turn), and
    the device half is one copy kernel per tensor from the pinned buffer into persistent device buffers, enqueued on the COMPUTE stream
    right in front of the step that reads them (1.8 MB for a C1 batch: ~35 us of a 6.7 ms step) - stream order alone keeps a device set
    from being overwritten before the step that read it has run; one event per pinned set tells the host when it may refill it.
    What comes out is the same dict with device tensors plus `num_tokens` = the host-side sum of a 2-D attention mask (free here, and
    what lets the engine run the padding-free layout without reading a count back).  `batch_training(data, model, train_cfg,
    train_stats, ...)` - the reference's form - sees tensors that are already on `train_stats.device`: its `.to(device)` calls return
    them as they are.

        for data in DevicePrefetcher(loader, device):
            batch_training(data, engine)

    The device tensors of a batch are reused for the batch three steps later: consume them inside the step (as the training step does).
    Measured (tools/prefetch_probe.py, C1 step, one box): device-resident batches 6.80 ms, this class 6.87 ms (+0.9 %), the reference's
    synchronous `.cuda()` per tensor 6.95 ms (+2.2 %).  Two traps on the way there: (1) `pinned.copy_(pageable)` - torch's copy_ INTO a
    pinned tensor synchronises with the device (3 ms on average, up to 94 ms under load): the staging buffers are filled through their
    numpy views (14 us); (2) the device half as hipMemcpyAsync (`dst.copy_(pinned, non_blocking=True)`) is a copy-engine job, 6.92 ms: the
    kernel form (gget_op_copy_from_host reads the pinned buffer over the host link) stays in the compute queue.
    Non-tensor entries pass through; tensors already on the device are left alone."""

    SETS = 3

    def __init__(self, batches: Iterable, device=None, count_tokens: bool = True):
        self.batches = batches
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.count_tokens = count_tokens
        self._pinned = [dict() for _ in range(self.SETS)]        # key -> pinned host tensor
        self._dev = [dict() for _ in range(self.SETS)]           # key -> device tensor
        self._copied = [None] * self.SETS                        # compute-stream event behind the set's last pinned -> device copies

    def __len__(self):
        return len(self.batches)

    def _stage(self, data, slot):
        """host half: the batch into pinned set `slot` (+ the token count)"""
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()     # (the device copy that last read this pinned set: three steps old)
            self._copied[slot] = None
        out, pend = {}, []
        for k, v in data.items():
            if not torch.is_tensor(v) or v.device == self.device:
                out[k] = v
                continue
            buf = self._pinned[slot].get(k)
            if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                self._pinned[slot][k] = buf
                self._dev[slot][k] = torch.empty(v.shape, dtype=v.dtype, device=self.device)
            # (NOT buf.copy_(v): torch's copy_ into a pinned tensor synchronises with the device - 3 ms on average, up to 94 ms with the
            #  GPU busy, against 14 us for the same bytes through the buffers' numpy views; that alone took the C1 step from 6.7 to 10 - 20 ms)
            np.copyto(buf.view(torch.uint8).numpy(), v.contiguous().view(torch.uint8).numpy())
            pend.append(k)
        if self.count_tokens and "num_tokens" not in out:
            am = data.get("attention_mask")
            if torch.is_tensor(am) and am.dim() == 2 and am.device.type == "cpu":
                out["num_tokens"] = int((am != 0).sum())
        return out, pend, slot

    def _to_device(self, staged):
        """device half, on the current (compute) stream, in front of the step"""
        out, pend, slot = staged
        if pend:
            import ctypes as C
            from . import _lib as L
            lib, st = L.load(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for k in pend:
            src, dst = self._pinned[slot][k], self._dev[slot][k]
            if os.environ.get("GGET_PF_COPY", "kernel") == "kernel":      # (memcpy: the copy-engine form, kept for tools/prefetch_probe.py)
                L.check(lib.gget_op_copy_from_host(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel() * src.element_size(), st))
            else:
                dst.copy_(src, non_blocking=True)
            out[k] = dst
        if pend:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._copied[slot] = ev
        return out

    def __iter__(self):
        it = iter(self.batches)
        try:
            cur = self._stage(next(it), 0)
        except StopIteration:
            return
        slot = 1
        data = self._to_device(cur)
        while cur is not None:
            try:
                nxt = self._stage(next(it), slot)       # the host half of the NEXT batch, before the consumer's step on this one is enqueued
            except StopIteration:
                nxt = None
            slot = (slot + 1) % self.SETS
            yield data
            # (the device half on a side stream, joined by an event, so that it runs beside the step: 6.88 ms against 6.74 - the fork / join
            #  packets cost more than the 35 us of copy they hide, as with the attention launches, profiles/r06_step_experiments.txt)
            data = self._to_device(nxt) if nxt is not None else None
            cur = nxt


# ----------------------------------------------------------------------------- evaluation pass
@torch.no_grad()
def evaluate(model, loader, eval_name: str = "valid", do_eval: bool = True):
    """Pre-train evaluation pass, reference log_eval_dump_utils.evaluate (:242-304): eval mode, one forward per batch with
    labels and sample weights but WITHOUT position_ids (the reference comments them out), head1 loss sum-reduced to rank 0
    and divided by the world size there, mean over the batches; returns (loss, None) and puts the model back in train mode.
    `do_eval=False` returns (None, None)."""
    if not do_eval:
        return None, None
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    model.eval()
    device = model.device
    losses, aux_losses = [], []
    for data in loader:
        out = model(input_ids=data["input_ids"].to(device), attention_mask=data["attention_mask"].to(device),
                    labels=data["labels"].to(device),
                    inputs_raw_embeds=data["embed"].to(device) if "embed" in data else None,     # log_eval_dump_utils.py:53-59
                    sample_wgt=data["wgt"].to(device) if "wgt" in data else None, **_layout_kw(model, data))
        loss, aux = out.head1_loss.clone(), out.head2_loss
        if world > 1:
            dist.reduce(loss, 0)
            if aux is not None:
                aux = aux.clone()
                dist.reduce(aux, 0)
            if rank == 0:
                loss = loss / world
                aux = aux / world if aux is not None else None
        losses.append(loss)
        if aux is not None and not torch.isnan(aux).item():
            aux_losses.append(aux)
    model.train()
    _check_deferred_all_ranks(model)
    if not losses:
        raise ValueError(f"evaluate: the {eval_name} loader yielded no batch")
    return sum(losses) / len(losses), None


def _check_deferred_all_ranks(model):
    """`model.check_deferred()` so that EVERY rank raises when any rank's device-side input guard fired: a single rank raising in
    front of a collective would leave the others blocked in it.  The flag is max-reduced over the default group."""
    m = getattr(model, "module", model)
    if not hasattr(m, "check_deferred"):
        return
    err = None
    try:
        m.check_deferred()
    except (IndexError, ValueError) as ex:
        err = ex
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = m.device if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if err is not None else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if err is None and int(flag.item()):
            err = RuntimeError("another rank's device-side input guard fired (position_ids outside the RoPE table or a var-len token "
                               "count mismatch): see that rank's error")
    if err is not None:
        raise err


def all_gather_varlen(q: torch.Tensor) -> torch.Tensor:
    """reference misc_utils.all_gather (:472-504): concatenates per-rank tensors of different lengths along dim 0."""
    ws = dist.get_world_size()
    local = torch.tensor(q.shape[0], device=q.device)
    sizes = [torch.zeros_like(local) for _ in range(ws)]
    dist.all_gather(sizes, local)
    mx = int(max(sizes).item())
    if mx > q.shape[0]:
        q = torch.cat([q, torch.zeros([mx - q.shape[0]] + list(q.shape[1:]), device=q.device, dtype=q.dtype)], dim=0)
    out = [torch.zeros_like(q) for _ in range(ws)]
    dist.all_gather(out, q)
    return torch.cat([o[: int(n)] for o, n in zip(out, sizes)])


@torch.no_grad()
def ft_evaluate(model, loader, *, problem_type: str = "single_label_classification", num_labels: int = 2,
                task_level: str = "task", metric_type: Optional[str] = None, dataset_name: str = "", eval_name: str = "valid"):
    """Fine-tune evaluation pass, reference log_eval_dump_utils.ft_evaluate (:77-163): eval mode; per batch one forward WITH
    task labels, sample weights and position_ids; the task loss is averaged over the batches, logits feed the metric object
    (`update(task_logits, labels, idx)`); with several ranks every entry of the metric's tensor dict is gathered from all
    ranks (variable length); the dataset's OGB-style evaluator runs on the gathered dict when there is one, otherwise the
    metric object's own results are returned.  Returns (loss, metrics, eval_result, input_dict) like the reference; the
    keyword arguments replace the fields the reference reads from its Hydra config."""
    from . import metrics as M
    model.eval()
    device = model.device
    cls_metrics = M.get_metrics(metric_type or problem_type, device, num_labels=num_labels)
    test_loss, j = 0, 0
    for j, data in enumerate(loader, 1):
        labels = data[f"{task_level}_labels"].to(device)
        labels = labels.float() if problem_type == "multi_label_classification" else labels
        res = model(input_ids=data["input_ids"].to(device), attention_mask=data["attention_mask"].to(device),
                    task_labels=labels, cls_idx=data["cls_idx"].to(device) if "cls_idx" in data else None,
                    inputs_raw_embeds=data["embed"].to(device) if "embed" in data else None,     # log_eval_dump_utils.py:108-110
                    sample_wgt=data["wgt"].to(device) if "wgt" in data else None, **_layout_kw(model, data),
                    position_ids=data["position_ids"].to(device) if "position_ids" in data else None)
        test_loss = test_loss + res.task_loss.detach()
        idx = data["idx"].to(device) if "idx" in data else torch.arange(labels.shape[0], device=device) + (j - 1) * labels.shape[0]
        cls_metrics.update(res.task_logits, labels, idx)
    model.train()
    _check_deferred_all_ranks(model)      # (a collective itself: every rank raises together, none is left inside the gathers below)
    if j == 0:
        raise ValueError(f"ft_evaluate: the {eval_name} loader yielded no batch")
    test_loss = test_loss / j
    input_dict = cls_metrics.to_dict()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        gdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        input_dict = {k: all_gather_varlen(v.to(gdev)).cpu() for k, v in input_dict.items()}
        # the reference's torchmetrics objects synchronise across ranks inside compute(): feed ours the gathered lists
        cls_metrics.compute({k: all_gather_varlen(v.to(gdev)).cpu().numpy() for k, v in cls_metrics.sync_dict().items()})
    else:
        cls_metrics.compute()
    res = M.evaluate_ogb(dataset_name, {k: v.numpy() for k, v in input_dict.items()})
    if res is None:
        res = cls_metrics.results_in_dict()
    return test_loss, cls_metrics, res, input_dict


# ----------------------------------------------------------------------------- distributed env

def dp_env_defaults() -> int:
    """Environment a multi-process job wants BEFORE its process group / communicator exists: the collective library is held to as many
    channels (= workgroups) as the GEMM launches leave CUs free - GGET_DP_RESERVE_CUS = R, OFF by default (0: tools/dp_standin.py measured that
    16 free CUs do not protect the exact-fit launches and that 32 cost more than the collisions they prevent at the 8-GPU residency of the
    collectives); an NCCL_MAX_NCHANNELS the user set wins.  Returns R (GgetEngine applies the GEMM side: gget_debug_set(15, R))."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    r = max(0, int(os.environ.get("GGET_DP_RESERVE_CUS", "0"))) if world > 1 else 0
    if r:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(r))
    return r

def set_dist_env(backend: Optional[str] = None):
    """reference misc_utils.set_dist_env (:507-539): env:// rendezvous, one process per GPU, barrier."""
    dp_env_defaults()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, init_method="env://", **kw)
        dist.barrier()
    return rank, local, world


# ----------------------------------------------------------------------------- pipeline
class TrainingMode(abc.ABC):
    """Strategy interface of the reference (src/training/mode.py:46-89).  The dataset / tokenizer / sampler halves of
    `prepare_data` stay on the host: a mode is constructed with what they would have produced -
    `batches` (an iterable of collated batches), and for a reference-shaped `Config` the figures the schedule arithmetic
    needs (`tokens_per_sample`, `samples_per_gpu`, `vocab_size` ...)."""

    model_cls = GraphGPTPretrainBase
    skip_keys = False                      # (mode.py: fine-tuning drops the `score` head of a pre-trained checkpoint)
    finetune = False

    def __init__(self, batches: Optional[Iterable] = None, tokens_per_sample: Optional[float] = None,
                 samples_per_gpu: Optional[int] = None, vocab_size: Optional[int] = None, bos_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, mask_inside_model: bool = False):
        self.batches, self.tokens_per_sample, self.samples_per_gpu = batches, tokens_per_sample, samples_per_gpu
        self.vocab_size, self.bos_token_id, self.eos_token_id = vocab_size, bos_token_id, eos_token_id
        # the tokenizer configuration's `pretrain_mlm.method == "inside_model"` (pretrain_mode.py:121-128): the SMTP masking runs in the
        # model's forward; the tokenizer half of the reference lives on the host, so the caller says which it is
        self.mask_inside_model = bool(mask_inside_model)

    @abc.abstractmethod
    def train_step(self, engine: GgetEngine, batch) -> torch.Tensor: ...

    def update_config(self, pipeline) -> None:
        return None

    def prepare_data(self, pipeline) -> None:
        return None

    def _set_model_config(self, pipeline):
        """modules_utils.set_model_config (:57-81): causal flag by task type, next_n_token = stacked_feat, tokenizer-derived ids."""
        from . import conf as CF
        mc, task = pipeline.model_cfg, CF._get(pipeline.train_cfg, "task_type")
        if CF._get(mc, "intermediate_size") == 0 and CF._get(mc, "num_attention_heads") == 0:      # modules_utils.py:37-42, :63-70
            hidden = CF._get(mc, "hidden_size")
            assert hidden % 64 == 0
            CF._set(mc, "intermediate_size", hidden * 4)
            CF._set(mc, "num_attention_heads", hidden // 64)
            CF._set(mc, "head_dim", 64)
        CF._set(mc, "causal_attention", bool(0 if task == "pretrain-mlm" else CF._get(mc, "causal_attention")))
        CF._set(CF._get(mc, "pt_head"), "next_n_token", CF._get(CF._get(mc, "graph_input"), "stacked_feat"))
        for name in ("vocab_size", "bos_token_id", "eos_token_id"):
            if getattr(self, name) is not None:
                CF._set(mc, name, getattr(self, name))
        return mc

    def post_model_setup(self, pipeline) -> bool:
        return False

    def allow_resume(self) -> bool:
        return True

    def allow_save_config(self) -> bool:
        return True

    def setup_optimizer(self, pipeline) -> None:
        if pipeline.reference_cfg:
            from . import conf as CF
            pipeline.optim = CF.optim_from_training(pipeline.train_cfg, pipeline.use_deepspeed, self.finetune)
        pipeline.engine = initialize(pipeline.model, pipeline.optim)
        pipeline.device = pipeline.model.device
        if pipeline.reference_cfg and not pipeline.use_deepspeed:
            # an empty `deepspeed_conf_file` selects the reference's DDP / AMP branch (training_utils.py:46-86): its optimizer step is
            # GradScaler.step, which SKIPS a step whose gradients are inf / NaN - the modes' train_step goes through engine.step(), so
            # the rule is set on the engine here (the lean `batch_training(batch, engine)` form never saw `use_deepspeed`)
            pipeline.engine.skip_nonfinite = True

    def setup_training(self, pipeline) -> None:
        return None

    def run_training(self, pipeline) -> None:
        t0 = time.time()
        tokens = None   # accumulated where the mask lives (no device->host read per step); read at log time only
        # a resumed run (pipeline._resume_checkpoint restored engine.global_steps) does the REMAINDER of the schedule, like the reference
        # continuing from its checkpoint's epoch / j_init (pipeline.py:178-202) - not max_steps further steps
        done0 = int(pipeline.engine.global_steps)
        if pipeline.max_steps and done0 >= pipeline.max_steps:
            pipeline.model.check_deferred()
            return
        # host batches are handed over one step ahead (pinned staging + a side stream: DevicePrefetcher); GGET_PREFETCH=0 = as they come
        source = pipeline.batches
        if bool(int(os.environ.get("GGET_PREFETCH", "1"))) and torch.cuda.is_available() and pipeline.batches is not None:
            source = DevicePrefetcher(pipeline.batches, pipeline.model.device)
        for step, batch in enumerate(source):
            loss = self.train_step(pipeline.engine, batch)
            pipeline.last_loss = loss
            am = batch["attention_mask"]
            n = batch["num_tokens"] if batch.get("num_tokens") is not None else \
                (am.sum() if am.dim() == 2 else am.diagonal(dim1=1, dim2=2).sum())   # packed rows: [B,S,S] block-diagonal
            tokens = n if tokens is None else tokens + n
            if pipeline.log_every and (step + 1) % pipeline.log_every == 0:
                torch.cuda.synchronize()
                dt = time.time() - t0
                pipeline.log(f"step {pipeline.engine.global_steps} loss {float(loss):.5f} lr {pipeline.engine.last_lr:.3e} "
                             f"tokens/s/gpu {int(tokens) / dt:.0f}")
                pipeline.log_lines.append(f"{pipeline.engine.global_steps},{float(loss):.6f},{pipeline.engine.last_lr:.6e}\n")
                pipeline.model.check_deferred()      # device-side input guards (position ids), read where the loop syncs anyway
            if pipeline.max_steps and pipeline.engine.global_steps >= pipeline.max_steps:
                break
        pipeline.model.check_deferred()


class PretrainMode(TrainingMode):
    model_cls = GraphGPTPretrainBase

    def prepare_data(self, pipeline):
        """The schedule / model-config half of reference pretrain_mode.py:96-230 (the dataset half is the constructor's arguments):
        optimizer.min_lr rule (:108), steps_per_saving (:114-116), smtp_inside off unless the tokenizer masks inside the model
        (:121-128), token-budget -> total / warm-up steps (:205-207), epochs (:213-215), nested model config -> flat (:219-220)."""
        if not pipeline.reference_cfg:
            return
        from . import conf as CF
        from .modeling import convert_to_legacy_config
        oc, sc, tc = pipeline.optim_cfg, pipeline.sched_cfg, pipeline.train_cfg
        CF._set(oc, "min_lr", CF._get(oc, "lr") * 0.1 if pipeline.use_deepspeed else 0)
        bs = CF._get(tc, "batch_size")
        if CF._get(sc, "samples_per_saving"):
            CF._set(sc, "steps_per_saving", CF._get(sc, "samples_per_saving") // (pipeline.world_size * bs))
        if self.tokens_per_sample is None:
            raise ValueError("PretrainMode(tokens_per_sample=...): the mean un-padded length the reference estimates from its "
                             "tokenizer (misc_utils.estimate_tokens_per_sample) is needed to turn the token budget into steps")
        # pretrain_mode.py:121-128: smtp_inside follows the tokenizer's masking method, whatever the model config said
        pt_head = CF._get(pipeline.model_cfg, "pt_head")
        if pt_head is not None:
            CF._set(pt_head, "smtp_inside", self.mask_inside_model)
        if self.mask_inside_model:
            assert CF._get(tc, "task_type") in ("pretrain-mlm", "pretrain-smtp")
            CF._set(tc, "task_type", "pretrain-smtp")
        tps = CF._get(pipeline.model_cfg, "max_position_embeddings") if CF._get(tc, "pack_tokens", 0) > 0 else self.tokens_per_sample
        if CF._get(tc, "task_type") == "pretrain-euler":        # pretrain_mode.py:191-195
            tps = tps // 2
        CF.update_num_steps(sc, tps, bs, pipeline.world_size)
        if self.samples_per_gpu:
            CF.update_epochs(sc, tps, self.samples_per_gpu, pipeline.world_size)
        pipeline.model_cfg = self._set_model_config(pipeline)
        pipeline.config = convert_to_legacy_config(pipeline.model_cfg)

    def train_step(self, engine, batch):
        return batch_training(batch, engine)


class FinetuneMode(TrainingMode):
    model_cls = GraphGPTTaskModel
    skip_keys = True
    finetune = True

    def prepare_data(self, pipeline):
        """reference finetune_mode.py:181-192: epoch budget -> steps, set_ft_model_config (modules_utils.py:84-92), nested -> flat."""
        if not pipeline.reference_cfg:
            return
        from . import conf as CF
        from .modeling import convert_to_legacy_config
        if self.samples_per_gpu is None:
            raise ValueError("FinetuneMode(samples_per_gpu=...): train samples per rank (len(sampler) // world) are needed for the schedule")
        CF.update_ft_num_steps(pipeline.train_cfg, self.samples_per_gpu)
        mc = self._set_model_config(pipeline)
        if len(CF._get(pipeline.train_cfg, "pretrain_cpt", "") or "") == 0:
            CF._set(mc, "num_key_value_heads", CF._get(mc, "num_attention_heads"))
        CF._set(mc, "tie_word_embeddings", False)
        CF._set(CF._get(mc, "pt_head"), "next_n_token", 1)
        pipeline.model_cfg = mc
        pipeline.config = convert_to_legacy_config(mc)

    def setup_optimizer(self, pipeline):
        if pipeline.reference_cfg:
            from . import conf as CF
            CF.set_finetune_cfg(CF._get(pipeline.train_cfg, "finetune"))       # finetune_mode.py:224
        super().setup_optimizer(pipeline)

    def train_step(self, engine, batch):
        return ft_batch_training(batch, engine)[0]


class TrainingPipeline:
    """`TrainingPipeline(cfg, mode).run()` (reference src/training/pipeline.py:15-216), the same phases in the same order.

    `cfg` is either the reference's `Config` shape - `cfg.model` = nested GraphGPTModelConfig, `cfg.training` with `.batch_size`,
    `.deepspeed_conf_file`, `.pretrain_cpt`, `.output_dir`, `.schedule`, `.optimizer`, `.distributed` (graph-gpt_amd/conf.py names the
    fields read; any attribute tree works: the reference's dataclasses, OmegaConf, SimpleNamespace) - or the lean form of earlier
    rounds: `model` (GraphGPTConfig or kwargs), `optim` (OptimConfig or kwargs), `batches`, `max_steps`, `log_every`, `output_dir`,
    `resume_from`.  With the reference shape the batch source is the mode's (`PretrainMode(batches=..., tokens_per_sample=...)`);
    the run is `training.schedule.total_num_steps` optimizer steps long."""

    def __init__(self, cfg: Any, mode: TrainingMode):
        from . import conf as CF
        self.cfg, self.mode = cfg, mode
        self.reference_cfg = CF.is_reference_config(cfg)
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        self.token_cfg = self.model_cfg = self.train_cfg = self.data_cfg = self.sched_cfg = self.optim_cfg = None
        self.use_deepspeed, self.pretrain_cpt, self.world_size, self.rank = False, "", 1, 0
        self.config = self.model = self.device = None
        self.engine: Optional[GgetEngine] = None
        self.last_loss = None
        self.log_lines = []        # "step,loss,lr" lines of this run -> <output_dir>/log.csv (the file whose presence means "resume", pipeline.py:129)
        self.max_steps, self.log_every = get("max_steps", 0) or 0, get("log_every", 0) or 0
        if self.reference_cfg:
            self.optim, self.batches = None, get("batches") if get("batches") is not None else mode.batches
            self.output_dir, self.resume_from = None, None
        else:
            from .modeling import GraphGPTConfig
            mc = get("model")
            self.config = mc if isinstance(mc, GraphGPTConfig) else GraphGPTConfig(**mc)
            oc = get("optim", {})
            self.optim = oc if isinstance(oc, OptimConfig) else OptimConfig(**(oc or {}))
            self.batches = get("batches") if get("batches") is not None else mode.batches
            self.output_dir, self.resume_from = get("output_dir"), get("resume_from")

    @property
    def model_config(self):                 # (name of earlier rounds)
        return self.config

    def log(self, msg: str):
        if self.rank == 0:
            print(msg, flush=True)

    # -- shared phases, reference-shaped config only (pipeline.py:97-139)
    def _extract_config(self):
        from . import conf as CF
        g = CF._get
        self.token_cfg, self.model_cfg, self.train_cfg = g(self.cfg, "tokenization"), g(self.cfg, "model"), g(self.cfg, "training")
        self.data_cfg = g(self.token_cfg, "data") if self.token_cfg is not None else None
        self.sched_cfg, self.optim_cfg = g(self.train_cfg, "schedule"), g(self.train_cfg, "optimizer")

    def _setup_deepspeed_flag(self):
        from . import conf as CF
        tc = self.train_cfg
        self.pretrain_cpt, self.output_dir = CF._get(tc, "pretrain_cpt", "") or "", CF._get(tc, "output_dir", None)
        self.use_deepspeed = len(CF._get(tc, "deepspeed_conf_file", "") or "") > 0
        CF._set(tc, "use_deepspeed", self.use_deepspeed)
        if self.output_dir and os.path.exists(os.path.join(self.output_dir, "log.csv")):
            self.log(f"log file {os.path.join(self.output_dir, 'log.csv')} exists, resume training from {self.output_dir} instead of "
                     f"initializing from pre-train ckp {self.pretrain_cpt}!")
            self.pretrain_cpt = self.output_dir

    def _setup_distributed(self):
        from . import conf as CF
        self.rank, _, self.world_size = set_dist_env()
        dc = CF._get(self.train_cfg, "distributed", None) if self.reference_cfg else None
        if dc is not None:
            CF._set(dc, "world_size", self.world_size)
            CF._set(dc, "rank", self.rank)

    def _create_model(self):
        self.model = self.mode.model_cls(self.config)
        self.model.gradient_checkpointing_enable()
        self.model.config.use_cache = False

    def _load_initial_ckp(self):
        """pipeline.py:165-176: a pre-trained checkpoint that is not the run's own output directory is loaded by name."""
        from . import checkpoint as CK
        self.model = CK.load_from_ckp(self.pretrain_cpt, self.output_dir or "", self.model, self.config, skip_keys=self.mode.skip_keys)

    def _resume_checkpoint(self):
        """pipeline.py:178-202: weights + optimizer state of the run's own latest checkpoint."""
        from . import checkpoint as CK
        if not (len(self.pretrain_cpt) > 0 and self.pretrain_cpt == self.output_dir and self.mode.allow_resume()):
            return
        ckp, _ = CK.get_latest_ckp(self.pretrain_cpt)
        if os.path.exists(os.path.join(ckp, "model.pt")):
            self.engine.load_checkpoint(ckp)

    def _save_model_config(self):
        if self.rank != 0 or not self.mode.allow_save_config() or not self.output_dir:
            return
        os.makedirs(self.output_dir, exist_ok=True)
        self.model.config.save_pretrained(self.output_dir)

    def run(self):
        if self.reference_cfg:
            self._extract_config()
            self.mode.update_config(self)
            self._setup_deepspeed_flag()
            self._setup_distributed()
            self.mode.prepare_data(self)
            if self.config is None:
                raise ValueError("the mode's prepare_data did not set pipeline.config (convert_to_legacy_config(pipeline.model_cfg))")
            from . import conf as CF
            if not self.max_steps:
                self.max_steps = int(CF._get(self.sched_cfg, "total_num_steps") or 0)
            if not self.log_every:
                self.log_every = 0
            self._create_model()
            if self.mode.post_model_setup(self):
                return self
            self._load_initial_ckp()
            self.model.cuda()
            self.mode.setup_optimizer(self)
            self._resume_checkpoint()
            self._save_model_config()
            self.mode.setup_training(self)
            self.mode.run_training(self)
            if self.output_dir and self.rank == 0 and self.mode.allow_save_config():
                self.engine.save_checkpoint(self.output_dir)
                # the reference's save_all writes log.csv next to its checkpoints (misc_utils.py:150-154); its presence in output_dir is
                # what makes the next run with this output_dir a RESUME (_setup_deepspeed_flag above / pipeline.py:129)
                with open(os.path.join(self.output_dir, "log.csv"), "a") as fh:
                    fh.writelines(self.log_lines or [f"{self.engine.global_steps},{float(self.last_loss) if self.last_loss is not None else float('nan'):.6f},"
                                                     f"{getattr(self.engine, 'last_lr', float('nan')):.6e}\n"])
            return self
        self._setup_distributed()
        self.mode.update_config(self)
        self.mode.prepare_data(self)
        self._create_model()
        if self.mode.post_model_setup(self):
            return self
        self.model.cuda()
        self.mode.setup_optimizer(self)
        if self.resume_from:
            self.engine.load_checkpoint(self.resume_from)
        self.mode.setup_training(self)
        self.mode.run_training(self)
        if self.output_dir and self.rank == 0:
            self.engine.save_checkpoint(self.output_dir)
        return self


def launch(fn: Callable, *args, **kwargs):
    """Entry-point wrapper with the reference's name (`launch(train)`, pipeline.py:229-257).  Ranks are created by
    `python -m torch.distributed.run`, one per GPU; the only thing the shim has to do is drop the launcher-injected `--local_rank*`
    arguments before the entry point reads sys.argv.  The reference's Hydra override rewriting is CLI-side and out of scope
    (DESIGN.md section 7): pass overrides as `key=value`."""
    import sys
    sys.argv = [a for a in sys.argv if not a.startswith("--local_rank")]
    return fn(*args, **kwargs)
