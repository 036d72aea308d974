"""CPU, world_size=2, gloo: the data-parallel exchange (fastspeech2_amd/ddp.py) that replaces the reference's
nn.DataParallel (train.py:42).  Checks, with two real processes:
  * bucketed prefix-ready all-reduce == mean of the per-rank flat gradients, for any ready() call pattern;
  * global_counts-normalised per-rank losses, averaged over ranks, == the reference's global-batch masked mean
    (train.py:82-86 computes the loss on the gathered batch);
  * shard_by_length deals disjoint, length-homogeneous batches to the ranks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastspeech2_amd import ddp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        # ---- 1. gradient exchange with uneven ready() prefixes and a small bucket size
        n = 10007
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g)
        every = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        want = sum(every) / world
        ex = ddp.GradExchange(flat, world, bucket_bytes=4 * 1000)
        for end in (10, 999, 1000, 4500, 4501, 9000):
            ex.ready(end)
        ex.finish()
        res["exchange_err"] = (flat - want).abs().max().item()
        res["reset"] = (ex.sent, len(ex.handles))
        # second step reuses the object
        flat.copy_(every[rank])
        ex.ready(n)
        ex.finish()
        res["exchange_err2"] = (flat - want).abs().max().item()

        # ---- 2. loss normalisation: per-rank masked sums / (global count / world), averaged == global masked mean
        gg = torch.Generator().manual_seed(7)
        pred = torch.randn(world, 6, 11, generator=gg)
        tgt = torch.randn(world, 6, 11, generator=gg)
        lens = torch.tensor([[11, 9, 7, 5, 3, 1], [4, 4, 4, 2, 2, 1]])[:world]
        valid = torch.arange(11)[None, None, :] < lens[:, :, None]
        global_mean = ((pred - tgt).abs() * valid).sum() / valid.sum()
        cnt = ddp.global_counts(valid[rank].sum().float().view(1))
        local = ((pred[rank] - tgt[rank]).abs() * valid[rank]).sum() / cnt[0]
        tot = local.clone()
        dist.all_reduce(tot)
        res["loss_err"] = abs((tot / world - global_mean).item())

        # ---- 3. sharding
        lengths = [int(x) for x in torch.randint(10, 200, (64,), generator=torch.Generator().manual_seed(3))]
        mine = ddp.shard_by_length(lengths, world, rank, 8)
        res["shard"] = mine
        res["lengths"] = lengths
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo_exchange_loss_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r in range(world):
        assert out[r]["exchange_err"] < 1e-6
        assert out[r]["exchange_err2"] < 1e-6
        assert out[r]["reset"] == (0, 0)
        assert out[r]["loss_err"] < 1e-6
    s0, s1 = out[0]["shard"], out[1]["shard"]
    lengths = out[0]["lengths"]
    assert len(s0) == len(s1) == 64 // 16
    flat0 = [i for b in s0 for i in b]
    flat1 = [i for b in s1 for i in b]
    assert not set(flat0) & set(flat1) and len(set(flat0) | set(flat1)) == 64
    for b0, b1 in zip(s0, s1):      # same step -> similar lengths on both ranks (adjacent chunks of one sorted group)
        assert min(lengths[i] for i in b0) >= max(lengths[i] for i in b1)


def test_single_process_exchange_is_identity():
    flat = torch.arange(100, dtype=torch.float32)
    ex = ddp.GradExchange(flat.clone(), world_size=1)
    ex.ready(50)
    ex.finish()
    assert torch.equal(ex.flat, flat)
    assert torch.equal(ddp.global_counts(torch.tensor([3.0])), torch.tensor([3.0]))
