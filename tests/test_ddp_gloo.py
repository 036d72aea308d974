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
    """WITHOUT a process group there is no collective to call: the exchange is a no-op (plain single-GPU training)."""
    flat = torch.arange(100, dtype=torch.float32)
    ex = ddp.GradExchange(flat.clone(), world_size=1)
    assert ex.active is False
    ex.ready(50)
    ex.finish()
    assert torch.equal(ex.flat, flat) and ex.n_buckets == 0
    assert torch.equal(ddp.global_counts(torch.tensor([3.0])), torch.tensor([3.0]))


def _one_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = [0]
        real = dist.all_reduce

        def counted(*a, **k):
            calls[0] += 1
            return real(*a, **k)
        dist.all_reduce = counted
        flat = torch.arange(10007, dtype=torch.float32)
        ex = ddp.GradExchange(flat.clone(), bucket_bytes=4000, tail_bytes=10000)      # world size from the group: 1
        ex.ready(5000)
        early = ex.n_buckets
        ex.ready(10007)
        ex.finish()
        ce = ddp.CountExchange()
        ce.start(torch.tensor([3, 5]), torch.tensor([7, 9]), 4, 8)
        c = ce(torch.tensor([0.0, 0.0]))
        g = ddp.global_counts(torch.tensor([2.0, 3.0]))
        q.put(dict(active=ex.active, world=ex.world, early=early, n_buckets=ex.n_buckets, last_step=ex.last_step, calls=calls[0],
                   same=bool(torch.equal(ex.flat, flat)), counts=c.tolist(), g=g.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_one_rank_group_issues_every_collective():
    """with a process group of ONE rank the exchange used to return early everywhere, so a "one-rank RCCL"
    run never called all_reduce.  Now a group - of any size - means real collectives: every bucket, the tail, the counts."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=100)
    p.join(30)
    assert p.exitcode == 0
    assert r["active"] is True and r["world"] == 1 and r["same"]
    # 10007 floats: tail region = last 2500 -> 1000-float buckets up to 7000 (7), 250-float pieces after (12), 7 floats in finish()
    assert r["early"] == 5 and r["n_buckets"] == 7 + 12 + 1 and r["last_step"] == (19, 1)
    assert r["calls"] == r["n_buckets"] + 2                      # + CountExchange.start + global_counts
    assert r["counts"] == [7.0, 15.0] and r["g"] == [2.0, 3.0]


def test_bucket_schedule_at_the_real_parameter_layout():
    """The exchange's bucket boundaries against the engine's "prefix final" offsets at the REAL layout (4 + 4 layers, 28.87 M
    trainable parameters = 115.5 MB of fp32 gradients; reference train.py:42 moves them with nn.DataParallel's gather).
    Engine._backward reports a final prefix after every decoder layer, after the variance adaptor, after every encoder layer and
    at the end (engine.py `_ready`); GradExchange.ready launches every whole bucket inside it.  Checked without a process group
    (the launches are recorded, not executed): every byte travels exactly once and in order, no bucket leaves before its prefix
    is final, the 32 MB / 8 MB bucket sizes hold, and what is left for finish() - the part exposed in front of clip + Adam - is at
    most one 8 MB tail bucket plus the prefix granularity of the last layer."""
    from fastspeech2_amd.ddp import GradExchange
    from fastspeech2_amd.model import FastSpeech2
    from tests.golden import configs
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4)
    m = FastSpeech2(pcfg, mcfg)
    offsets, total = {}, 0
    for n, p in m._trainable_in_backward_order():          # the same walk as FastSpeech2._ensure_flat
        offsets[n] = total
        total += (p.numel() + 7) // 8 * 8
    n_param = sum(p.numel() for _, p in m._trainable_in_backward_order())
    assert 28.8e6 < n_param < 28.95e6 and total * 4 < 116.5e6
    # the offsets Engine._backward passes to the hook, in order (engine.py: decoder layers 3..0 -> variance adaptor -> encoder 3..0)
    names = [f"decoder.layer_stack.{i}.pos_ffn.layer_norm.weight" for i in (3, 2, 1, 0)]
    names += ["variance_adaptor.energy_predictor.linear_layer.weight"]
    names += [f"encoder.layer_stack.{i}.pos_ffn.layer_norm.weight" for i in (3, 2, 1, 0)] + ["encoder.src_word_emb.weight"]
    ends = [offsets[n] for n in names] + [total]
    assert ends == sorted(ends) and ends[0] > 0            # the PostNet + mel_linear block is final first
    flat = torch.zeros(total)
    ex = GradExchange(flat, world_size=2, collectives=True)      # schedule inspection only: _launch is replaced below
    launched = []
    ex._launch = lambda lo, hi, producers=(): launched.append((lo, hi, cur[0]))
    cur = [0]
    for e in ends:
        cur[0] = e
        ex.ready(e)
    before_finish = len(launched)
    cur[0] = total
    if ex.sent < ex.n:
        launched.append((ex.sent, ex.n, total))
    # exactly once, in order
    assert launched[0][0] == 0 and launched[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(launched, launched[1:]))
    # never ahead of the final prefix
    assert all(hi <= final for lo, hi, final in launched)
    sizes = [(hi - lo) * 4 for lo, hi, _ in launched]
    # the size rule: 32 MiB while a whole one fits in front of the last 48 MiB of the buffer, 8 MiB pieces from there on (at
    # this layout - 110.1 MiB - that is ONE 32 MiB bucket: PostNet + mel_linear + most of the decoder, then 8 MiB pieces)
    for (lo, hi, _), sz in zip(launched[:before_finish], sizes):
        assert sz == ((32 << 20) if (lo + (32 << 20) // 4) <= ex.tail_start else (8 << 20)), (lo, hi, sz)
    assert sizes[0] == 32 << 20
    exposed = sizes[-1] if len(launched) > before_finish else 0
    # the embedding table (361 x 256) and the first encoder layer's attention block are the last gradients to become final
    assert exposed <= (8 << 20), exposed
    # overlap budget: bytes that can travel while backward still runs vs the whole buffer
    early = sum(s for (lo, hi, final), s in zip(launched, sizes) if final < total)
    print(f"{len(launched)} buckets: {[round(s / 2 ** 20, 1) for s in sizes]} MiB; {early / (total * 4):.0%} of the bytes launched before "
          f"backward ends, {exposed / 2 ** 20:.1f} MiB left for finish()")
    assert early >= 0.85 * total * 4
