"""Data-parallel train step END TO END on one GPU box: two real processes share the MI355X and exchange gradients over
gloo (it carries device tensors through the host; RCCL refuses two ranks on one device).  This drives exactly the code the
8-GPU run uses - engine prefix hooks, side-stream joins, bucketed overlapped all-reduce, globally normalised loss, fused
clip+Adam - and checks it against a single process that sees both ranks' batches:
    averaged DP gradient == mean of the two per-batch gradients computed with the global counts   (fp32, dropout off)
    averaged DP gradient == the ORACLE's gradient of the reference's gathered-batch loss (two replicas, per-replica BatchNorm)
    parameters after the step are identical on both ranks and equal to the single-process update."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _setup(rank_seed, dev):
    from oracle.weights import seeded_state_dict, synthetic_batch
    from tests.golden import configs
    from fastspeech2_amd.model import FastSpeech2
    pcfg, mcfg = configs.make(dropout=False, dec_layers=2, enc_layers=2)
    model = FastSpeech2(pcfg, mcfg, compute_dtype="fp32")
    model.load_state_dict(seeded_state_dict(model.state_dict(), 3))
    model.to(dev).train()
    model.disable_dropout = True
    model._ensure_flat(dev)
    b = synthetic_batch(50 + rank_seed, 3, 20 + 4 * rank_seed)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    batch12 = (None, None, d["speakers"], d["texts"], d["src_lens"], d["max_src_len"], d["mels"], d["mel_lens"], d["max_mel_len"],
               d["pitches"], d["energies"], d["durations"])
    return model, batch12, pcfg, mcfg


def _fwd_bwd(model, batch12, loss_fn):
    out = model(*batch12[2:])
    loss_fn(batch12, out)[0].backward()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastspeech2_amd import ddp
        from fastspeech2_amd.model import FastSpeech2Loss, ScheduledOptim
        from tests.golden import configs
        dev = torch.device("cuda:0")
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
        model, batch12, pcfg, mcfg = _setup(rank, dev)
        ex = ddp.GradExchange(model.flat_gradients(), world, bucket_bytes=8 << 20)
        model._engine.grad_hook = ex.ready
        dist.broadcast(model.flat_parameters(), 0)
        loss_fn = FastSpeech2Loss(pcfg, mcfg, count_reduce=ddp.global_counts)
        opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
        for _ in range(2):
            _fwd_bwd(model, batch12, loss_fn)
            ex.finish()
            if _ == 0:
                g1 = model.flat_gradients().clone()
                named = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
            opt.step_and_update_lr(zero_grad=True)
        torch.cuda.synchronize()
        q.put((rank, g1.cpu().numpy(), model.flat_parameters().detach().cpu().numpy(), named))     # by value: the worker exits
    finally:
        dist.destroy_process_group()


def test_two_rank_step_matches_single_process(dev):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, p, named = q.get(timeout=600)
        res[r] = (torch.from_numpy(g), torch.from_numpy(p), named)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]), "ranks diverged after two data-parallel steps"
    assert torch.allclose(res[0][0], res[1][0], rtol=0, atol=0), "all-reduced gradients differ between ranks"

    # single process: both batches, each normalised by (global valid counts / world), gradients averaged
    from fastspeech2_amd.model import FastSpeech2Loss, ScheduledOptim
    from tests.golden import configs
    model, b0, pcfg, mcfg = _setup(0, dev)
    _, b1, _, _ = _setup(1, dev)
    cnt = sum(torch.stack([b[4].sum(), b[7].sum()]).float() for b in (b0, b1)) / world
    loss_fn = FastSpeech2Loss(pcfg, mcfg, count_reduce=lambda c: cnt.to(c.device))
    opt = ScheduledOptim(model, configs.TRAIN, mcfg, 0)
    gsum = torch.zeros_like(model.flat_gradients())
    for b in (b0, b1):
        model.flat_gradients().zero_()
        _fwd_bwd(model, b, loss_fn)
        gsum += model.flat_gradients()
    gref = (gsum / world).cpu()
    g = res[0][0]
    assert (g - gref).abs().max().item() <= 2e-5 * gref.abs().max().item() + 1e-9

    # ... and against the ORACLE of what the reference's DataParallel step differentiates (train.py:42, 82-86): one loss over the
    # gathered outputs of two replicas, each with its own train-mode BatchNorm statistics (the comparison above
    # is product vs product).  The exchanged gradient is the rank AVERAGE of gradients of losses normalised by (global count /
    # world) = the gradient of the gathered-batch loss itself.
    from oracle.weights import seeded_state_dict, synthetic_batch
    from tests.helpers import oracle_gathered_case
    sd = seeded_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, 3)
    batches = [synthetic_batch(50 + r, 3, 20 + 4 * r) for r in range(world)]
    _, ograds = oracle_gathered_case(pcfg, mcfg, sd, batches)
    named = res[0][2]
    assert set(ograds) <= set(named) | {n for n in ograds if n.endswith("w_ks.bias")}
    checked = 0
    for n, og in ograds.items():
        if n.endswith("w_ks.bias"):                       # exactly zero in exact arithmetic (softmax shift invariance)
            continue
        gp = torch.from_numpy(named[n]).double()
        scale = og.abs().max().item()
        err = (gp - og).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (n, err, scale)
        checked += 1
    assert checked >= 100
