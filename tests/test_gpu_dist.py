"""Data-parallel step on the device with two ranks (both on cuda:0, gloo transport because one box has one GPU and RCCL
refuses two ranks on one device): the staged backward + per-bucket all-reduce + 1/world scaling + clip + AdamW of
training.GgetEngine must give both ranks identical parameters, equal to one process that averages the two ranks' gradients
itself.  The clip threshold is below the gradient norm, so the (deterministically reduced) norm feeds every update."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(modeling):
    return modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                   num_attention_heads=2, max_position_embeddings=1024, causal_attention=False,
                                   stacked_feat=13, next_n_token=13)


def _batch(synth, rank, layout="padded"):
    b = synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=700 + rank)
    d = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
    if layout == "varlen":      # the collator's token count: the step runs on the rank's compacted real tokens (another row count per rank)
        d["num_tokens"] = int(b["attention_mask"].sum())
    return d


def _worker(rank, world, port, q, overlap="1", layout="padded"):
    os.environ["GGET_DP_OVERLAP"] = overlap
    os.environ["GGET_VARLEN"] = "0" if layout == "padded" else ""      # (a device-side mask alone selects the var-len layout since round 4)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="env://")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
    eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
    assert eng.world == 2
    data = _batch(synth, rank, layout)
    losses = []
    for _ in range(2):
        losses.append(float(tr.batch_training(data, eng)))
    torch.cuda.synchronize()
    e = model._engine
    assert e.varlen_status()[0] == (layout == "varlen")
    q.put((rank, losses, e.master.detach().cpu().numpy(), float(eng.last_grad_norm)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,layout", [("1", "padded"), ("0", "padded"), ("1", "varlen")])
def test_two_rank_step_matches_manual_gradient_average(overlap, layout, monkeypatch):
    """overlap=1: bucketed all-reduce on a side stream behind the staged backward; overlap=0: one all-reduce after it; layout "varlen":
    every rank runs its step on its own compacted real tokens (the gradient buckets are the same on every rank whatever its row count)."""
    monkeypatch.setenv("GGET_VARLEN", "0" if layout == "padded" else "")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap, layout)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][2], res[1][2])          # replicas stay bit-identical
    assert res[0][1][0] != res[1][1][0]                            # ... on different data
    assert abs(res[0][3] - res[1][3]) == 0.0                       # same clipped global norm on both ranks

    # one process, two replicas' gradients averaged by hand (fp32 sum of the bf16 buckets, then the same 1/world scale) - with the kernel
    # menu of a data-parallel rank = the single-GPU menu since round 5 (GGET_DP_LDS_HEADROOM=1 would add (2, 2), GGET_DP_RESERVE_CUS=R
    # (15, R), (13, 0): both opt-in)
    L_ = importlib.import_module("graph-gpt_amd._lib")
    for key, val in ((15, 0), (13, 1), (2, 1)):
        L_.check(L_.load().gget_debug_set(key, val))
    models = [modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1) for _ in range(2)]
    engs = [tr.initialize(m, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05)) for m in models]
    datas = [_batch(synth, r, layout) for r in range(2)]
    for _ in range(2):
        for m, en, d in zip(models, engs, datas):
            out = en(input_ids=d["input_ids"], attention_mask=d["attention_mask"], labels=d["labels"], num_tokens=d.get("num_tokens"))
            en.backward(out.head1_loss)
        torch.cuda.synchronize()
        g = [m._engine.grad_bf16 for m in models]
        tot = (g[0].float() + g[1].float()).to(torch.bfloat16)      # what a bf16 sum all-reduce of two ranks produces
        for m, en in zip(models, engs):
            m._engine.grad_bf16.copy_(tot)
            en.world = 2                                            # the step divides by the world size
            en.step()
            en.world = 1
        torch.cuda.synchronize()
    for key, val in ((15, 0), (13, 1), (2, 1)):
        L_.check(L_.load().gget_debug_set(key, val))
    ref = models[0]._engine.master.detach().cpu().numpy()
    np.testing.assert_allclose(res[0][2], ref, rtol=0, atol=1e-6)


def test_rccl_one_rank_communicator_through_staged_backward(monkeypatch):
    """The RCCL path of the C ABI (gget_comm_unique_id / gget_comm_init / gget_allreduce_grads_async) executed for real:
    a one-rank communicator, the staged backward with one all-reduce per bucket on a HIP side stream, in bf16 and with the
    fp32-accumulate option.  At world 1 the exchange is the identity, so gradients, updated parameters and losses must
    reproduce the monolithic single-process step (not bitwise: the fp32-atomic reductions of the norm-weight / embedding
    gradients and of the loss sum are order-dependent between any two runs)."""
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    data = _batch(synth, 0)

    def run(abi, fp32):
        monkeypatch.setenv("GGET_DP_BACKEND", "abi" if abi else "torch")
        monkeypatch.setenv("GGET_FORCE_STAGED", "1" if abi else "0")
        monkeypatch.setenv("GGET_DP_FP32_REDUCE", "1" if fp32 else "0")
        model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
        losses = [float(tr.batch_training(data, eng)) for _ in range(2)]
        torch.cuda.synchronize()
        e = model._engine
        out = (losses, e.grad_bf16.detach().float().cpu().numpy().copy(), e.master.detach().cpu().numpy().copy())
        if abi:
            assert e.comm_world == 1
            # a larger batch re-creates the engine handle: the communicator must MOVE to the new handle (no collective re-init,
            # which would hang the ranks whose batch did not grow - ADVICE r2) and the next exchange must still work
            big = synth.make_pretrain_batch(B=16, S=40, F=13, V=756, seed=900)
            big = {k: torch.from_numpy(v).cuda() for k, v in big.items() if k != "lengths"}
            l3 = float(tr.batch_training(big, eng))
            torch.cuda.synchronize()
            e2 = model._engine
            assert e2 is not e and e2.comm_world == 1 and e.comm_world == 0 and np.isfinite(l3)
            assert e2.step_count == 3
            e2.comm_destroy()
        return out

    ref0_master = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1).cuda()._engine.master.detach().cpu().numpy().copy()
    ref = run(False, False)
    for fp32 in (False, True):
        got = run(True, fp32)
        np.testing.assert_allclose(got[0], ref[0], rtol=2e-6)
        assert float(np.linalg.norm(got[1] - ref[1]) / np.linalg.norm(ref[1])) < 2e-3
        upd = np.linalg.norm(ref[2] - ref0_master)
        assert float(np.linalg.norm(got[2] - ref[2])) < 0.02 * upd


_RCCL_ONE_RANK = r"""
import importlib, json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GGET_REPO"])
modeling = importlib.import_module("graph-gpt_amd.modeling")
tr = importlib.import_module("graph-gpt_amd.training")
synth = importlib.import_module("graph-gpt_amd.synth")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                              num_attention_heads=2, max_position_embeddings=1024, causal_attention=False, stacked_feat=13, next_n_token=13)
b = synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=40)
data = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
out = {}
for staged in (0, 1):
    os.environ["GGET_FORCE_STAGED"] = str(staged)
    for fp32 in ((0,) if not staged else (0, 1)):
        os.environ["GGET_DP_FP32_REDUCE"] = str(fp32)
        model = modeling.GraphGPTPretrainBase(cfg, seed=1)
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
        losses = [float(tr.batch_training(data, eng)) for _ in range(3)]
        torch.cuda.synchronize()
        e = model._engine
        out[f"{staged}{fp32}"] = {"losses": losses, "master_sum": float(e.master.double().abs().sum()), 
                                  "dp": eng.describe_dp() if staged else None, "world": eng.world, "forced": eng.force_staged}
        np.save(os.path.join(os.environ["GGET_OUT"], f"master_{staged}{fp32}.npy"), e.master.detach().cpu().numpy())
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_torch_rccl_one_rank_group_through_staged_backward(tmp_path):
    """The DEFAULT data-parallel path (torch.distributed all-reduce of the flat bf16 gradient array, one collective per bucket on the side
    stream behind per-bucket events) with the REAL backend: a one-rank NCCL (= RCCL) process group on cuda:0 and GGET_FORCE_STAGED=1, in
    bf16 and with the fp32-accumulate option.  A one-rank all-reduce is the identity, so three steps must reproduce the monolithic
    single-process steps (losses to 2e-6, master weights to 2 % of the update: fp32-atomic reductions differ between any two runs).
    Own process: the process group must not leak into the other tests."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               GGET_REPO=ROOT, GGET_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GGET_DP_BACKEND", None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    ref = out["00"]
    m0 = np.load(tmp_path / "master_00.npy")
    for key in ("10", "11"):
        got = out[key]
        assert got["forced"] and got["world"] == 1
        assert "nccl" in got["dp"]["backend"] or "rccl" in got["dp"]["backend"], got["dp"]
        np.testing.assert_allclose(got["losses"], ref["losses"], rtol=2e-6)
        m = np.load(tmp_path / f"master_{key}.npy")
        assert float(np.linalg.norm(m - m0)) <= 2e-2 * 3 * 1e-3 * np.sqrt(m0.size), key     # (3 steps of lr 1e-3: |update| <= 3e-3 per element)


def test_bf16_vs_fp32_bucket_reduction_drift_world8():
    """Bounds what a bf16 SUM all-reduce over 8 ranks costs against the fp32-accumulated reduction (GGET_DP_FP32_REDUCE=1):
    eight replicas' gradient sets (same weights, eight different batches) are produced on one GPU and summed (a) in a bf16
    ring order - rounding after every hop, what RCCL's ring does with ncclBfloat16 - and (b) in fp32 with one final rounding.
    The two averaged gradients, and the parameters after the clip + AdamW step they feed, must agree to bf16 resolution."""
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
    eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1.0))
    e = model._ensure_engine(8, 32)
    grads = []
    for r in range(8):
        d = _batch(synth, r)
        out = eng(input_ids=d["input_ids"], attention_mask=d["attention_mask"], labels=d["labels"])
        eng.backward(out.head1_loss)
        grads.append(e.grad_bf16.detach().clone())
    ring = grads[0].clone()
    for g in grads[1:]:
        ring = (ring.float() + g.float()).to(torch.bfloat16)           # one rounding per hop
    wide = torch.stack([g.float() for g in grads]).sum(0)
    exact = wide.to(torch.bfloat16)                                      # fp32 accumulate, one rounding
    rel = float((ring.float() - wide).norm() / wide.norm())
    rel_exact = float((exact.float() - wide).norm() / wide.norm())
    assert rel_exact < 3e-3 and rel < 3 * rel_exact + 1e-3, (rel, rel_exact)
    master0 = e.master.detach().clone()
    m0, v0 = e.adam_m.detach().clone(), e.adam_v.detach().clone()
    res = []
    for summed in (ring, exact):
        e.master.copy_(master0); e.adam_m.copy_(m0); e.adam_v.copy_(v0); e.step_count = 0
        e.sync_params()
        e.grad_bf16.copy_(summed)
        eng.world = 8
        eng.step()
        eng.world = 1
        res.append(e.master.detach().clone())
    upd = (res[1] - master0).norm()
    drift = float((res[0] - res[1]).norm() / upd)
    assert drift < 0.05, drift     # Adam normalises the update, so sign flips of tiny gradients dominate: a few % of the step
    import json
    os.makedirs(os.path.join(os.path.dirname(__file__), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "dp_reduce_drift.json"), "w") as fh:
        json.dump({"world": 8, "grad_rel_l2_bf16_ring_vs_fp32": rel, "grad_rel_l2_single_rounding": rel_exact,
                   "param_update_rel_l2_drift_after_one_adamw_step": drift}, fh)


def test_bench_two_ranks_gloo_smoke():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), on ONE GPU over gloo
    (GGET_BENCH_BACKEND=gloo: RCCL refuses two ranks on a device): the N > 1 line must carry the whole-job value, weak scaling and the
    `dp` diagnostics (bucket layout, step time without the exchange, exposed communication)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GGET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "toy-tiny", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    dp = d["dp"]
    assert dp["world"] == 2 and dp["n_buckets"] == 2 + 2 and dp["backend"].endswith("gloo")
    assert dp["menu_probe"]["probed"] and dp["overlap_with_backward"] == dp["menu_probe"]["chosen"]["overlap"]
    assert dp["ms_per_step_without_exchange"] > 0 and abs(dp["exposed_comm_ms"] - (d["ms_per_step"] - dp["ms_per_step_without_exchange"])) < 1e-9
    assert np.isfinite(d["smtp_loss"])


def test_c_abi_exchange_schedule_loopback_world2_and_coalesced_buckets(monkeypatch):
    """The C-ABI exchange beyond one rank on a one-GPU box: a loopback communicator (gget_comm_init_loopback) makes this
    process rank 0 of TWO ranks that hold the same gradients - every collective doubles its range on the side stream, AdamW's grad_scale
    is 1/2 - so the staged backward + per-bucket exchange + step must reproduce the single-rank step (doubling / halving is exact in
    bf16; the runs still differ by the order of the fp32-atomic reductions).  A collective that ran before its bucket was final, a
    missing stream wait in front of AdamW, a wrong range or a forgotten 1/world would all break the match.  Repeated with the buckets
    coalesced into >= 1 MiB collectives (GGET_DP_BUCKET_MB) - fewer collectives, same result."""
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    data = _batch(synth, 0)

    def run(loop_world, bucket_mb="0"):
        monkeypatch.setenv("GGET_DP_BACKEND", "abi" if loop_world else "torch")
        monkeypatch.setenv("GGET_DP_LOOPBACK_WORLD", str(loop_world))
        monkeypatch.setenv("GGET_FORCE_STAGED", "1")
        monkeypatch.setenv("GGET_DP_BUCKET_MB", bucket_mb)
        model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
        eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
        losses = [float(tr.batch_training(data, eng)) for _ in range(3)]
        torch.cuda.synchronize()
        e = model._engine
        info = eng.describe_dp()
        if loop_world:
            assert eng.world == loop_world and e.comm_world == loop_world
            e.comm_destroy()
        return losses, e.master.detach().cpu().numpy().copy(), float(eng.last_grad_norm), info

    master0 = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1).cuda()._engine.master.detach().cpu().numpy().copy()
    ref = run(0)
    upd = np.linalg.norm(ref[1] - master0)
    for world, mb in ((2, "0"), (2, "1"), (4, "1000")):
        got = run(world, mb)
        np.testing.assert_allclose(got[0], ref[0], rtol=2e-6)
        assert float(np.linalg.norm(got[1] - ref[1])) < 0.02 * upd, (world, mb)
        # the clipped global norm is the norm of the AVERAGED gradient: world x the gradients, scaled by 1/world
        assert abs(got[2] - ref[2]) <= 2e-3 * ref[2]
        nb = got[3]["n_buckets"]
        assert got[3]["collectives_per_step"] == (nb if mb == "0" else (1 if mb == "1000" else got[3]["collectives_per_step"]))
        assert got[3]["collectives_per_step"] <= nb and abs(sum(got[3]["collective_mb"]) - sum(got[3]["bucket_mb"])) < 0.2
    # without the 1/world the loopback run must NOT match (the test has teeth): world 2 with grad_scale forced to 1
    monkeypatch.setenv("GGET_DP_BACKEND", "abi")
    monkeypatch.setenv("GGET_DP_LOOPBACK_WORLD", "2")
    model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
    eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=1e9))     # no clipping: the raw scale reaches AdamW's moments
    out = eng(input_ids=data["input_ids"], attention_mask=data["attention_mask"], labels=data["labels"])
    eng.backward(out.head1_loss)
    eng.world = 1
    gn_unscaled = float(eng.step())
    eng.world = 2
    out = eng(input_ids=data["input_ids"], attention_mask=data["attention_mask"], labels=data["labels"])
    eng.backward(out.head1_loss)
    gn_scaled = float(eng.step())
    model._engine.comm_destroy()
    assert gn_unscaled > 1.5 * gn_scaled


def test_bench_eight_ranks_gloo_smoke():
    """`bench.py --gpus 8` as the driver launches it, eight ranks sharing the one GPU over gloo (no 8-GPU node is
    available to this round - this is the readiness check, not a scaling number): the line must carry the L + 2 bucket layout, the
    replicas must be bit-identical after the timed steps, the exposed-communication diagnostic must be finite."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GGET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GGET_DP_BUCKET_MB="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--workload", "toy-tiny", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp8"
    assert d["config"]["global_batch"] == 8 * d["config"]["per_gpu_batch"]
    dp = d["dp"]
    assert dp["world"] == 8 and dp["n_buckets"] == 2 + 2 and dp["collectives_per_step"] == 4
    assert dp["replicas_bit_identical"] is True
    assert np.isfinite(dp["exposed_comm_ms"]) and np.isfinite(d["smtp_loss"])
    assert len(dp["rank_devices"]) == 8 and dp["distinct_devices"] >= 1 and dp["rank_devices"][3]["rank"] == 3
    # the start-up probe (round 6): three exchange arrangements timed on this machine, the fastest one kept by EVERY rank
    mp = dp["menu_probe"]
    assert mp["probed"] is True and len(mp["menus"]) == 3 and all(m["ms_per_step"] > 0 for m in mp["menus"])
    best = min(mp["menus"], key=lambda m: m["ms_per_step"])
    assert (mp["chosen"]["overlap"], mp["chosen"]["reserve_cus"]) == (best["overlap"], best["reserve_cus"])
    assert dp["overlap_with_backward"] == mp["chosen"]["overlap"] and dp["reserved_cus"] == mp["chosen"]["reserve_cus"]


@pytest.mark.parametrize("backend", ["torch", "abi"])
def test_bench_two_gpus_over_rccl(backend):
    """The first real multi-GPU evidence, whenever a box has >= 2 GPUs (auto-skips on the one-GPU boxes of this pool): `bench.py
    --gpus 2` launched exactly as the driver's SCALE run does, over RCCL, with the gradient exchange through torch.distributed
    (default) and through the C ABI's own communicator (GGET_DP_BACKEND=abi: ncclAllReduce on the engine's side stream).  The
    replicas must be bit-identical after the timed steps, the backend must be RCCL, the ranks must sit on distinct devices, the
    exposed-communication diagnostic must be finite.  reference: src/utils/opt_utils.py:13 (DDP), src/utils/misc_utils.py:519-526."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs, this box has {torch.cuda.device_count()}")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GGET_DP_BACKEND=backend)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GGET_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    dp = d["dp"]
    assert d["n_gpus"] == 2 and dp["world"] == 2 and d["config"]["parallelism"] == "dp2"
    assert dp["replicas_bit_identical"] is True
    assert ("nccl" in dp["backend"] or "rccl" in dp["backend"]), dp["backend"]
    assert (backend == "abi") == (dp["backend"] == "rccl-via-c-abi")
    assert dp["distinct_devices"] == 2 and len(dp["rank_devices"]) == 2
    assert np.isfinite(dp["exposed_comm_ms"]) and np.isfinite(d["smtp_loss"]) and dp["rccl_version"]
    assert d["value"] > 0 and d["scaling"] == "weak"
