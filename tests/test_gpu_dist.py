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
    return modeling.GraphGPTConfig(vocab_size=756, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                   num_attention_heads=2, max_position_embeddings=1024, causal_attention=False,
                                   stacked_feat=13, next_n_token=13)


def _batch(synth, rank):
    b = synth.make_pretrain_batch(B=8, S=32, F=13, V=756, seed=700 + rank)
    return {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}


def _worker(rank, world, port, q, overlap="1"):
    os.environ["GGET_DP_OVERLAP"] = overlap
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="env://")
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    model = modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1)
    eng = tr.initialize(model, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05))
    assert eng.world == 2
    data = _batch(synth, rank)
    losses = []
    for _ in range(2):
        losses.append(float(tr.batch_training(data, eng)))
    torch.cuda.synchronize()
    e = model._engine
    q.put((rank, losses, e.master.detach().cpu().numpy(), float(eng.last_grad_norm)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_rank_step_matches_manual_gradient_average(overlap):
    """overlap=1: bucketed all-reduce on a side stream behind the staged backward; overlap=0: one all-reduce after it."""
    modeling = importlib.import_module("graph-gpt_amd.modeling")
    tr = importlib.import_module("graph-gpt_amd.training")
    synth = importlib.import_module("graph-gpt_amd.synth")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][2], res[1][2])          # replicas stay bit-identical
    assert res[0][1][0] != res[1][1][0]                            # ... on different data
    assert abs(res[0][3] - res[1][3]) == 0.0                       # same clipped global norm on both ranks

    # one process, two replicas' gradients averaged by hand (fp32 sum of the bf16 buckets, then the same 1/world scale)
    models = [modeling.GraphGPTPretrainBase(_cfg(modeling), seed=1) for _ in range(2)]
    engs = [tr.initialize(m, tr.OptimConfig(lr=1e-3, max_grad_norm=0.05)) for m in models]
    datas = [_batch(synth, r) for r in range(2)]
    for _ in range(2):
        for m, en, d in zip(models, engs, datas):
            out = en(input_ids=d["input_ids"], attention_mask=d["attention_mask"], labels=d["labels"])
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
    ref = models[0]._engine.master.detach().cpu().numpy()
    np.testing.assert_allclose(res[0][2], ref, rtol=0, atol=1e-6)
