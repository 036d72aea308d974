"""CPU, world_size = 8, gloo: the N = 8 job of BASELINE configs[2] / [3] as far as a box without GPUs can run it - eight real
processes, the REAL parameter layout (4 + 4 layers, 28.87 M trainable parameters = 115.5 MB of fp32 gradients), the real
sampler.  Replaces the reference's nn.DataParallel scatter / gather (train.py:42) and its per-process sort window
(train.py:30-37, dataset.py:127-146).  Checks
  * data.BucketedBatchSampler at world 8: every step's 8 x 48 items are disjoint across the ranks, together they are exactly one
    contiguous slice of the step's length-sorted window, dealt card-wise - so the ranks' longest items (what sets each rank's
    padded T, i.e. its step time) differ by no more than neighbouring items of one sorted list do;
  * ddp.GradExchange at the real layout, driven by the engine's own "prefix final" offsets: the bucket schedule of every rank is
    the same (1 x 32 MiB, then 8 MiB pieces; what is left for finish() <= 8 MiB), the exchanged gradient equals the mean of the
    eight ranks' gradients, and after 2 optimiser steps the eight replicas hold BIT-IDENTICAL parameters;
  * ddp.CountExchange: the global valid-position counts of the eight ranks' batches.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD, BATCH = 8, 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grad(step, rank, n):
    return torch.randn(n, generator=torch.Generator().manual_seed(1000 * step + rank))


def _pool():
    g = torch.Generator().manual_seed(99)
    return torch.clamp(torch.exp(torch.randn(8192, generator=g) * 0.78 + 3.89), 5, 250).long().numpy()     # bench.py's LibriTTS-like pool


def _worker(rank, world, port, ends, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastspeech2_amd import ddp
        from fastspeech2_amd.data import BucketedBatchSampler
        res = {}
        # ---- sampler
        pool = _pool()
        steps = list(iter(BucketedBatchSampler(pool, BATCH, world_size=world, rank=rank, group_size=4, shuffle=True, seed=1234)))
        res["steps"] = steps
        # ---- counts of this rank's first batch (phoneme rows / mel rows ~ 7 frames per phoneme)
        src = torch.from_numpy(pool[steps[0]])
        mel = src * 7
        ce = ddp.CountExchange()
        ce.start(src, mel, int(src.max()), min(int(mel.max()), 1000))
        c = ce(torch.zeros(2))
        res["counts"] = c.tolist()
        res["my_counts"] = [float(src.sum()), float(mel.clamp(max=1000).sum())]
        # ---- two optimiser steps at the real layout
        params = torch.randn(total, generator=torch.Generator().manual_seed(7))          # the same initial replica on every rank
        flat = torch.zeros(total)
        ex = ddp.GradExchange(flat, world)                                                # default 32 MiB / 8 MiB schedule
        sched = []
        real_launch = ex._launch

        def launch(lo, hi, producers=()):
            sched.append((lo, hi))
            return real_launch(lo, hi, producers)
        ex._launch = launch
        for step in (1, 2):
            flat.copy_(_grad(step, rank, total))
            for e in ends:                                # the engine's prefix hooks, in backward order
                ex.ready(e)
            early = len(sched)
            ex.finish()
            if step == 1:
                res["sched"] = list(sched)
                res["early"] = early
                res["last_step"] = ex.last_step
                if rank == 0:
                    # against the mean of all eight ranks' gradients, on three windows of the buffer (start, a bucket seam, the end)
                    want = sum(_grad(1, r, total) for r in range(world)) / world
                    seam = sched[0][1]
                    res["exchange_err"] = max((flat[a:b] - want[a:b]).abs().max().item()
                                              for a, b in ((0, 4096), (seam - 2048, seam + 2048), (total - 4096, total)))
                    res["exchange_err_all"] = (flat - want).abs().max().item()
                    del want
            params.add_(flat, alpha=-0.01)
            sched.clear()
        chk = torch.stack([params.double().sum(), params.double().abs().sum(), params[::4099].double().sum()])
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        res["replicas_bit_identical"] = all(torch.equal(c0, allchk[0]) for c0 in allchk)
        res["n_buckets"] = ex.n_buckets
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _layout():
    """the flat gradient buffer's layout and the offsets Engine._backward reports as final, from the model itself"""
    from fastspeech2_amd.model import FastSpeech2
    from tests.golden import configs
    pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4)
    m = FastSpeech2(pcfg, mcfg)
    offsets, total = {}, 0
    for n, p in m._trainable_in_backward_order():
        offsets[n] = total
        total += (p.numel() + 7) // 8 * 8
    names = [f"decoder.layer_stack.{i}.pos_ffn.layer_norm.weight" for i in (3, 2, 1, 0)]
    names += ["variance_adaptor.energy_predictor.linear_layer.weight"]
    names += [f"encoder.layer_stack.{i}.pos_ffn.layer_norm.weight" for i in (3, 2, 1, 0)] + ["encoder.src_word_emb.weight"]
    return [offsets[n] for n in names] + [total], total


@pytest.mark.timeout(600)
def test_world8_sampler_exchange_and_replicas_at_the_real_layout():
    ends, total = _layout()
    assert total * 4 > 110 << 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, ends, total, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=500) for _ in range(WORLD))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # ---- sampler: card-wise dealing of one sorted window slice per step
    pool = _pool()
    n_steps = len(out[0]["steps"])
    assert n_steps == 8192 // (WORLD * BATCH) and all(len(out[r]["steps"]) == n_steps for r in range(WORLD))
    order = np.random.default_rng(1234).permutation(len(pool))
    per_step, window = WORLD * BATCH, 4 * WORLD * BATCH
    seen = set()
    for s in range(n_steps):
        batches = [out[r]["steps"][s] for r in range(WORLD)]
        assert all(len(b) == BATCH for b in batches)
        items = [i for b in batches for i in b]
        assert len(set(items)) == per_step and not (set(items) & seen)                  # disjoint across ranks and across steps
        seen |= set(items)
        w0 = (s * per_step) // window * window
        w = order[w0:w0 + window]
        w = w[: len(w) // per_step * per_step]
        w = w[np.argsort(-pool[w], kind="stable")]
        s0 = s * per_step - w0
        assert sorted(items) == sorted(w[s0:s0 + per_step].tolist())                    # exactly this step's slice of the sorted window
        for r in range(WORLD):
            assert batches[r] == w[s0 + r:s0 + per_step:WORLD].tolist()                  # card-wise: rank r takes items r, r + 8, ...
        # what sets a rank's padded length (its longest item) differs across the ranks by at most the spread of the 8 longest items
        longest = [int(pool[b].max()) for b in batches]
        top8 = np.sort(pool[w[s0:s0 + per_step]])[::-1][:WORLD]
        assert max(longest) - min(longest) <= int(top8[0] - top8[-1])
        tot = [int(pool[b].sum()) for b in batches]
        assert max(tot) <= 1.1 * min(tot) + 64, (s, tot)                                  # and the valid rows per rank are balanced
    # ---- counts
    want_counts = [sum(out[r]["my_counts"][k] for r in range(WORLD)) / WORLD for k in (0, 1)]
    for r in range(WORLD):
        assert out[r]["counts"] == pytest.approx(want_counts, rel=1e-6)
    # ---- exchange
    sched0 = out[0]["sched"]
    for r in range(WORLD):
        assert out[r]["sched"] == sched0 and out[r]["last_step"] == out[0]["last_step"]  # every rank issues the same collectives in order
        assert out[r]["replicas_bit_identical"]
    assert sched0[0] == (0, (32 << 20) // 4) and sched0[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(sched0, sched0[1:]))
    sizes = [(hi - lo) * 4 for lo, hi in sched0]
    assert all(sz == 8 << 20 for sz in sizes[1:-1]) and sizes[-1] <= 8 << 20
    early, late = out[0]["last_step"]
    assert late == 1 and early == out[0]["early"] == len(sched0) - 1                     # all but the last piece under "backward"
    assert out[0]["n_buckets"] == 2 * len(sched0)
    assert out[0]["exchange_err"] < 1e-6 and out[0]["exchange_err_all"] < 1e-6
