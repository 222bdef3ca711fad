"""world_size-2 gloo tests of the N>1 host path (runs on CPU): env:// rendezvous, per-rank data sharding, and the
bucketed gradient exchange in the order backward completes the buckets, followed by the 1/world scaling."""
import importlib
import os
import socket
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

tr = importlib.import_module("graph-gpt_amd.training")
synth = importlib.import_module("graph-gpt_amd.synth")
spec_mod = importlib.import_module("graph-gpt_amd.spec")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = tr.set_dist_env(backend="gloo")
    assert (r, w) == (rank, world)
    # bucket layout of a tiny model: embeddings | layers | final norm + heads, contiguous, completion order
    spec = spec_mod.spec_from_size("tiny", vocab_size=300, stacked_feat=1, next_n_token=1)
    sizes = []
    for name, shp in spec.param_table().items():
        sizes.append(int(np.prod(shp)))
    n = sum(sizes)
    cut = [0, sizes[0], n - sizes[-1] - sizes[-2], n]
    buckets = [(cut[2], cut[3] - cut[2]), (cut[1], cut[2] - cut[1]), (cut[0], cut[1] - cut[0])]  # heads, layers, embed
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    works = [tr.all_reduce_bucket(flat, b, None, async_op=True) for b in buckets]
    for wk in works:
        wk.wait()
    other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ok = torch.allclose(flat, mine + other, atol=1e-6)
    # data sharding: distinct seeds => distinct batches, same shapes
    b = synth.make_pretrain_batch(B=4, S=16, F=1, V=300, seed=tr.shard_seed(1234, rank))
    q.put((rank, bool(ok), int(b["input_ids"].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "bucketed sum all-reduce mismatch"
    assert res[0][2] != res[1][2], "ranks must draw different batches"


class _FakeOut:
    def __init__(self, loss):
        self.head1_loss, self.head2_loss = loss, None


class _FakeModel:
    """Stands in for the device model on CPU: loss = mean of the labels it is shown (the test is about the host loop)."""
    device = torch.device("cpu")

    def __init__(self):
        self.mode, self.seen_kwargs = "train", None

    def eval(self):
        self.mode = "eval"

    def train(self):
        self.mode = "train"

    def __call__(self, **kw):
        assert self.mode == "eval"
        self.seen_kwargs = sorted(kw)
        return _FakeOut(kw["labels"].float().mean())


def _eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    tr.set_dist_env(backend="gloo")
    loader = [{"input_ids": torch.zeros(2, 3, 1, dtype=torch.int64), "attention_mask": torch.ones(2, 3, dtype=torch.int64),
               "labels": torch.full((2, 3, 1), float(10 * rank + i))} for i in range(3)]
    m = _FakeModel()
    loss, aux = tr.evaluate(m, loader, "valid")
    q.put((rank, float(loss), aux, m.mode, m.seen_kwargs))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_reduces_loss_to_rank0_gloo_world2():
    """reference log_eval_dump_utils.evaluate: per batch the rank losses are summed onto rank 0 and divided by the world size,
    then averaged over the batches; position_ids are not forwarded; the model ends in train mode."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    # rank 0: mean_i ((i) + (10 + i)) / 2 = mean_i (5 + i) = 6
    assert abs(res[0][1] - 6.0) < 1e-6 and res[0][2] is None
    assert res[0][3] == "train" and res[0][4] == ["attention_mask", "input_ids", "inputs_raw_embeds", "labels", "sample_wgt"]


def test_evaluate_single_process():
    m = _FakeModel()
    loader = [{"input_ids": torch.zeros(1, 2, 1, dtype=torch.int64), "attention_mask": torch.ones(1, 2, dtype=torch.int64),
               "labels": torch.full((1, 2, 1), float(v))} for v in (1, 3)]
    loss, aux = tr.evaluate(m, loader)
    assert float(loss) == 2.0 and aux is None and m.mode == "train"
    assert tr.evaluate(m, loader, do_eval=False) == (None, None)


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method="env://")
    tr = importlib.import_module("graph-gpt_amd.training")
    idx = list(range(100, 100 + 1003))
    ft0 = tr.finetune_rank_sampler(idx, world, rank, seed=42, epoch=0)
    ft1 = tr.finetune_rank_sampler(idx, world, rank, seed=42, epoch=1)
    ev = tr.eval_rank_sampler(idx, world, rank, shuffle_seed=7)
    pt = tr.pretrain_rank_sampler(idx[:50], epochs=2, seed=1000, rank=rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (ft0, ft1, ev, pt))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_sharding_rules_gloo_world2():
    """The three partition rules of the reference: fine-tune = disjoint equal shards of one per-epoch
    permutation (loader_utils.py:78-90, :622-627); eval = sorted positions modulo world, every sample once (:70-75); pre-train
    = every rank the full list, own shuffle (loader_utils.py:328-333, misc_utils.py:536-538)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (ft0a, ft1a, eva, pta), (ft0b, ft1b, evb, ptb) = got
    n = 1003
    # fine-tune: equal, disjoint, truncated to a multiple of the world size, a new permutation every epoch
    assert len(ft0a) == len(ft0b) == n // 2 and not set(ft0a) & set(ft0b)
    assert len(set(ft0a) | set(ft0b)) == (n // 2) * 2
    assert ft0a != ft1a and not set(ft1a) & set(ft1b)
    # the exact reference arithmetic, re-stated: randperm(seed + epoch)[rank:total:world]
    g = torch.Generator(); g.manual_seed(42)
    perm = torch.randperm(n, generator=g).tolist()
    assert ft0a == [100 + i for i in perm[0:1002:2]] and ft0b == [100 + i for i in perm[1:1002:2]]
    # eval: a partition of ALL samples, sizes differ by at most one, rank r holds sorted positions r mod world
    assert sorted(eva + evb) == list(range(100, 100 + n)) and abs(len(eva) - len(evb)) <= 1
    assert sorted(eva) == list(range(100, 100 + n, 2))
    # pre-train: both ranks hold the full multi-epoch list, in different orders
    assert sorted(pta) == sorted(ptb) == sorted(list(range(100, 150)) * 2) and pta != ptb
    tr = importlib.import_module("graph-gpt_amd.training")
    assert tr.schedule_steps(1e9, 20.0, 256, 8) == int(1e9 // (20.0 * 256 * 8))


class _FakeTaskModel:
    """Stands in for GraphGPTTaskModel in the fine-tune evaluation pass: logits are a fixed function of the sample index."""
    device = torch.device("cpu")

    def __init__(self):
        self.mode = "train"

    def eval(self):
        self.mode = "eval"

    def train(self):
        self.mode = "train"

    def __call__(self, **kw):
        assert self.mode == "eval" and kw["position_ids"] is not None and kw["task_labels"] is not None
        s = kw["input_ids"][:, 0, 0].float()                      # the sample's id doubles as its score
        lg = torch.stack([torch.zeros_like(s), s], dim=1)
        out = types.SimpleNamespace(task_loss=s.mean(), task_logits=lg)
        return out


def _ft_eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    tr.set_dist_env(backend="gloo")
    # eval sharding rule: sorted positions modulo world; rank 0 gets 6 samples, rank 1 gets 5 (variable-length gather)
    mine = tr.eval_rank_sampler(list(range(11)), world, rank)
    loader = []
    for a in range(0, len(mine), 4):
        ids = torch.tensor(mine[a:a + 4])
        loader.append({"input_ids": ids.view(-1, 1, 1).repeat(1, 3, 2), "attention_mask": torch.ones(len(ids), 3, dtype=torch.int64),
                       "position_ids": torch.arange(3)[None].repeat(len(ids), 1), "task_labels": (ids % 2), "idx": ids})
    m = _FakeTaskModel()
    loss, met, res, d = tr.ft_evaluate(m, loader, problem_type="single_label_classification", num_labels=2, dataset_name="ogbl-ppa")
    q.put((rank, float(loss), res, {k: v.tolist() for k, v in d.items()}, m.mode))
    dist.barrier()
    dist.destroy_process_group()


def test_ft_evaluate_gathers_all_ranks_gloo_world2():
    """reference log_eval_dump_utils.ft_evaluate (:77-163): every rank evaluates its eval shard, the metric tensors of all ranks
    are gathered (different lengths), the dataset evaluator sees every sample exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ft_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, loss, ogb, d, mode in res:
        assert mode == "train"
        assert sorted(d["idx"]) == list(range(11)) and len(d["y_pred"]) == 11          # gathered: every sample once
        assert sorted(zip(d["idx"], d["y_pred"])) == [(i, float(i)) for i in range(11)]
        # positives = odd ids {1,3,5,7,9}, negatives = even ids: fewer than 100 negatives -> Hits@100 = 1 (OGB convention)
        assert ogb == {"hits@100": 1.0}
    # per-rank mean of per-batch mean scores
    assert abs(res[0][1] - np.mean([np.mean([0, 2, 4, 6]), np.mean([8, 10])])) < 1e-6


def test_exchange_groups_coalesce_buckets_in_completion_order():
    """GGET_DP_BUCKET_MB: consecutive buckets of the completion order become ONE collective once they reach the size;
    the groups tile the flat gradient array exactly, every group is exchanged when its LAST bucket completes, the default is one
    collective per bucket."""
    sizes = [8_000_000] + [9_437_184] * 12 + [600_000]          # head, layers L-1..0, embeddings (the base model's layout)
    offs, o = [], sum(sizes)
    for s_ in sizes:
        o -= s_
        offs.append(o)
    e = types.SimpleNamespace(buckets=list(zip(offs, sizes)))
    eng = tr.GgetEngine.__new__(tr.GgetEngine)
    for mb, want in ((0, 14), (40, 5), (60, 4), (10 ** 6, 1)):
        eng.bucket_mb, eng._groups = mb, None
        groups = eng.exchange_groups(e)
        assert len(groups) == want
        covered = sorted((off, off + cnt) for off, cnt in groups.values())
        assert covered[0][0] == 0 and covered[-1][1] == sum(sizes)
        assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))       # an exact tiling, no overlap
        for last, (off, cnt) in groups.items():                                     # a group ends with the bucket that closes it
            assert off == e.buckets[last][0] and (mb == 0 or cnt * 2 >= mb * 2 ** 20 or last == len(sizes) - 1)
    # neighbours that are NOT adjacent in memory are never merged
    e2 = types.SimpleNamespace(buckets=[(100, 10), (50, 10), (40, 10)])
    eng.bucket_mb, eng._groups = 10 ** 6, None
    assert eng.exchange_groups(e2) == {0: (100, 10), 2: (40, 20)}
