"""Dev tool (2 ranks, gloo, one GPU): after ONE exchanged backward, which parameters' gradients differ between the ranks?
    FS2_BENCH_BACKEND=gloo FS2_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dbg_ddp.py [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench

args = bench.parse(sys.argv[1:])
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
device = bench.init_rank(world, int(os.environ["LOCAL_RANK"]))
from fastspeech2_amd import ddp
torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=-1))
model, loss_fn, opt, b, _, _ = bench.build(args, device, rank, world)
ex = ddp.GradExchange(model.flat_gradients(), world)
model._engine.grad_hook = ex.ready
dist.broadcast(model.flat_parameters(), 0)
step, fwd_bwd = bench.make_step(model, loss_fn, opt, b, ex)
if os.environ.get("DBG_WGRAD") == "1":                   # which weight-gradient launch first produces a non-finite value, and from what
    from fastspeech2_amd import ops
    _orig = ops.conv_wgrad
    seen = [0]

    def checked(dy, x, dw, S, taps=1, dil=1, pad=0, lens=None, dbias=None, **kw):
        torch.cuda.synchronize()
        before = bool(torch.isfinite(dw).all()) and (dbias is None or bool(torch.isfinite(dbias).all()))
        _orig(dy, x, dw, S, taps=taps, dil=dil, pad=pad, lens=lens, dbias=dbias, **kw)
        torch.cuda.synchronize()
        after = bool(torch.isfinite(dw).all()) and (dbias is None or bool(torch.isfinite(dbias).all()))
        if before and not after and seen[0] < 6:
            seen[0] += 1
            M = dy.shape[0]
            t = torch.arange(M, device=dy.device) % S
            if lens is not None:
                valid = t < lens.repeat_interleave(S)[:M].to(t.dtype) if lens.numel() * S == M else torch.ones_like(t, dtype=torch.bool)
            else:
                valid = torch.ones_like(t, dtype=torch.bool)
            bad_dy = ~torch.isfinite(dy.float()).all(1)
            bad_x = ~torch.isfinite(x.float()).all(1)
            nz_dy_pad = (dy.float().abs().sum(1) != 0) & ~valid
            print(f"[rank {rank}] non-finite gradient from conv_wgrad dy{tuple(dy.shape)} x{tuple(x.shape)} S={S} taps={taps} lens={lens is not None}: "
                  f"dy bad rows {int(bad_dy.sum())} (valid {int((bad_dy & valid).sum())}), x bad rows {int(bad_x.sum())} (valid {int((bad_x & valid).sum())}), "
                  f"nonzero dy rows in padding {int(nz_dy_pad.sum())}; first bad x row {int(torch.nonzero(bad_x)[0]) if bad_x.any() else -1} "
                  f"t={int(t[torch.nonzero(bad_x)[0]]) if bad_x.any() else -1}", flush=True)
    ops.conv_wgrad = checked
for it in range(3):
    fwd_bwd()
    ex.finish()
    torch.cuda.synchronize()
    g = model.flat_gradients().clone()
    allg = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(allg, g)
    if rank == 0:
        bad = []
        for n, p in model._trainable_in_backward_order():
            o = model._flat_offsets[n]
            a, c = allg[0][o:o + p.numel()], allg[1][o:o + p.numel()]
            if not torch.equal(a, c):
                bad.append((n, o, int((a != c).sum()), float((a - c).abs().max()), float(a.abs().max())))
        print(f"iter {it}: L={b['max_src_len']} T={b['max_mel_len']} differing tensors: {len(bad)}", flush=True)
        for x in bad[:12]:
            print("   ", x, flush=True)
    opt.step_and_update_lr(zero_grad=True)
    torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
