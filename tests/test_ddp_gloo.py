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
        # the same counts exchanged AHEAD of the forward pass (CountExchange.start ... __call__) and the blocking fallback
        cx = ddp.CountExchange()
        src_l = lens[rank]
        mel_l = lens[rank] * 7
        cx.start(src_l, mel_l, 9, 40)                                    # clamped to the padded lengths L = 9, T = 40
        got = cx(torch.zeros(2))
        want_c = torch.stack([lens.clamp(max=9).sum(), (lens * 7).clamp(max=40).sum()]).float() / world
        res["count_err"] = (got - want_c).abs().max().item()
        res["count_fallback_err"] = (cx(torch.stack([src_l.clamp(max=9).sum(), mel_l.clamp(max=40).sum()]).float()) - want_c).abs().max().item()
        # graduated buckets: 1000-element buckets, the last 2500 elements in 250-element pieces
        flat.copy_(every[rank])
        ex2 = ddp.GradExchange(flat, world, bucket_bytes=4000, tail_bytes=10000, overlap=False)
        sizes = []
        orig_launch = ex2._launch
        ex2._launch = lambda lo, hi, producers=(): (sizes.append(hi - lo), orig_launch(lo, hi, producers))[1]
        for end in (3000, 7000, 8300, n):
            ex2.ready(end)
        ex2.finish()
        res["exchange_err3"] = (flat - want).abs().max().item()
        res["bucket_sizes"] = sizes

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
        assert out[r]["count_err"] == 0 and out[r]["count_fallback_err"] == 0
        assert out[r]["exchange_err3"] < 1e-6
        bs = out[r]["bucket_sizes"]
        assert sum(bs) == 10007 and bs[:7] == [1000] * 7 and all(b == 250 for b in bs[7:-1]) and bs[-1] == 7, bs
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
