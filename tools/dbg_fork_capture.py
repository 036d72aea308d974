"""Dev tool: where does a hipGraph capture with the side stream forked INSIDE it deviate from the eager step?
fp32, dropout off, B = 8, L = 40: ONE step's flat gradient (before the optimiser) from
  eager two-stream | capture single-stream | capture forked | capture forked with every tensor allocated during the capture kept
  alive until the capture ends (separates "a block of the graph's pool is reused by the other branch while still read" from "an
  edge between the branches is missing")
each replayed 3 times (a race shows as replay-to-replay differences), compared per tensor with the eager single-stream gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastspeech2_amd
fastspeech2_amd.configure_hw_queues()
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
DT = sys.argv[1] if len(sys.argv) > 1 else "fp32"
BATCH, PH = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("8", "40")


def build(side):
    args = bench.parse(["--batch", BATCH, "--phonemes", PH, "--dtype", DT, "--side-stream", str(side)])
    torch.manual_seed(1234)
    model, loss_fn, opt, b, _, _ = bench.build(args, dev, 0, 1)
    model.disable_dropout = True
    model._engine.device_seed = True
    v = os.environ.get("FORK_VARIANT")
    if v == "nobranch":         # weight gradients forked, variance predictors on the main stream
        model._engine.concurrent_branches = False
    if v == "noctx":            # branch sections pin the launch stream but do NOT switch torch's current stream (allocations stay on the main stream's pool)
        from fastspeech2_amd import engine as E
        from fastspeech2_amd import ops as O
        def enter(self):
            e = self.eng
            e._side_stream.wait_stream(e._branch_main)
            self.pin = O.pinned_stream(e._side_stream)
            self.pin.__enter__()
            e._on_side = True
            return self
        def exit_(self, *exc):
            self.eng._on_side = False
            self.pin.__exit__(*exc)
        E.Engine._Branch.__enter__, E.Engine._Branch.__exit__ = enter, exit_
    if v in ("trim0", "trim1", "trim2"):
        # the backward predictor-branch block with less and less inside (gradients are wrong; only the read-before-write verdict counts):
        # trim0: nothing launched on the side stream (fork + events only); trim1: rowdot_bwd only; trim2: rowdot_bwd + the first LayerNorm backward
        from fastspeech2_amd import ops as O
        eng = model._engine
        def pred_bwd(W, G, kind, sv, dpred, B, seed_dev, dx_acc):
            pre = f"variance_adaptor.{kind}_predictor."
            cl = pre + "conv_layer."
            out = torch.zeros_like(sv.x) if v == "trim0" else None
            if v == "trim0":
                return out
            dn2 = O.rowdot_bwd(sv.n2, eng.P[pre + "linear_layer.weight"], dpred, sv.lens, G[pre + "linear_layer.weight"],
                               G[pre + "linear_layer.bias"], B, sv.S)
            if v == "trim1":
                return dn2
            _, dc2 = O.ln_bwd(sv.c2, dn2, eng.P[cl + "layer_norm_2.weight"], None, sv.m2, sv.r2, G[cl + "layer_norm_2.weight"],
                              G[cl + "layer_norm_2.bias"], B, sv.S, want_d1=False, want_d2=True, relu_bwd=True)
            return dc2
        eng._pred_bwd = pred_bwd
    if v == "nowait":           # _wgrad inside a branch section (already on the side stream) does not make the side stream wait for main again
        from fastspeech2_amd import ops as O
        eng = model._engine
        real_wgrad = eng._wgrad
        def wgrad(gw, gb, dy, x, S, taps=1, pad=0, lens=None):
            if not eng._on_side:
                return real_wgrad(gw, gb, dy, x, S, taps=taps, pad=pad, lens=lens)
            if gw.dim() == 3:
                gw = gw.permute(0, 2, 1)
            eng._ln_flush()
            O.conv_wgrad(dy, x, gw, S, taps=taps, pad=pad, lens=lens, dbias=gb, ws_owner=eng._wgrad_ws)
            eng._side_keep.append((dy, x))
        eng._wgrad = wgrad
    if v == "norecord":         # no Tensor.record_stream anywhere
        torch.Tensor.record_stream = lambda self, s: None
    if v in ("fwdonly", "bwdonly"):   # the predictor branch forked only in the forward / only in the backward pass
        from fastspeech2_amd import engine as E
        eng = model._engine
        real_bwd, real_fwd = eng._backward, eng._forward
        if v == "fwdonly":
            def bwd(sv, *a, **k):
                sv.branch = False
                return real_bwd(sv, *a, **k)
            eng._backward = bwd
        else:
            def fwd(*a, **k):
                eng.concurrent_branches = False
                try:
                    return real_fwd(*a, **k)
                finally:
                    eng.concurrent_branches = True
            def bwd(sv, *a, **k):
                sv.branch = True
                return real_bwd(sv, *a, **k)
            eng._forward, eng._backward = fwd, bwd
    step, fwd_bwd = bench.make_step(model, loss_fn, opt, b, None)
    return model, opt, fwd_bwd


def eager(side):
    model, opt, fwd_bwd = build(side)
    opt.zero_grad()
    fwd_bwd()
    torch.cuda.synchronize()
    return model, model.flat_gradients().clone()


def captured(fork, keep_all):
    model, opt, fwd_bwd = build(1)
    for _ in range(2):                       # warm-up (allocator, lazy buffers); gradients cleared afterwards
        fwd_bwd()
    torch.cuda.synchronize()
    model._engine.fork_in_capture = fork
    keep = []
    real_empty, real_like = torch.empty, torch.empty_like
    if keep_all:
        def e(*a, **k):
            t = real_empty(*a, **k); keep.append(t); return t
        def el(*a, **k):
            t = real_like(*a, **k); keep.append(t); return t
        torch.empty, torch.empty_like = e, el
    g = torch.cuda.CUDAGraph()
    opt.zero_grad()
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            fwd_bwd()
    finally:
        torch.empty, torch.empty_like = real_empty, real_like
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        model.flat_gradients().zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        outs.append(model.flat_gradients().clone())
    return model, outs, keep


def report(tag, model, ref, got):
    rel = float((got - ref).norm() / ref.norm())
    worst = []
    for name, prm in model._trainable_in_backward_order():
        off, n = model._flat_offsets[name], prm.numel()
        a, b = ref[off:off + n], got[off:off + n]
        d = float((a - b).norm() / (a.norm() + 1e-30))
        worst.append((d, name, float(a.norm())))
    worst.sort(reverse=True)
    print(f"{tag}: rel {rel:.3e}; worst tensors: " + ", ".join(f"{n} {d:.2e}" for d, n, _ in worst[:6]), flush=True)


def compare():
    m0, ref = eager(0)
    _, e1 = eager(1)
    report("eager two-stream vs eager one-stream", m0, ref, e1)
    for fork, keep_all, tag in ((False, False, "capture one-stream"), (True, False, "capture FORKED"), (True, True, "capture FORKED, every tensor kept alive")):
        try:
            m, outs, keep = captured(fork, keep_all)
        except Exception as ex:
            print(f"{tag}: FAILED {type(ex).__name__}: {str(ex)[:300]}", flush=True)
            continue
        for i, o in enumerate(outs):
            report(f"{tag} (replay {i})", m, ref, o)
        print(f"   replay-to-replay: {float((outs[1] - outs[0]).norm() / outs[0].norm()):.3e}, {float((outs[2] - outs[0]).norm() / outs[0].norm()):.3e}", flush=True)
        del keep



def poison_search(dt="fp32"):
    """Read-before-write finder: forked capture with every capture-time allocation kept (so every tensor has its own address for the
    life of the graph) and its ALLOCATION SITE recorded; before a replay the kept tensors are filled with NaN - a kernel that
    reads a buffer before the replay's own producer has written it (a missing edge between the branches; masked in later replays
    of an unpoisoned graph by the previous replay's identical values) turns the gradient NaN.  Bisects to the single buffer."""
    import traceback
    global DT
    DT = dt
    model, opt, fwd_bwd = build(1)
    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    model._engine.fork_in_capture = True
    keep, sites = [], []
    real_empty, real_like, real_zeros = torch.empty, torch.empty_like, torch.zeros

    def site():
        fr = [f for f in traceback.extract_stack()[:-2] if "fastspeech2_amd" in f.filename or "bench.py" in f.filename]
        return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in fr[-3:][::-1])

    def e(*a, **k):
        t = real_empty(*a, **k); keep.append(t); sites.append(site()); return t

    def el(*a, **k):
        t = real_like(*a, **k); keep.append(t); sites.append(site()); return t
    torch.empty, torch.empty_like = e, el
    g = torch.cuda.CUDAGraph()
    opt.zero_grad()
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            fwd_bwd()
    finally:
        torch.empty, torch.empty_like = real_empty, real_like
    torch.cuda.synchronize()
    fl = [i for i, t in enumerate(keep) if t.is_floating_point() and t.is_cuda]
    print(f"{len(keep)} capture-time allocations, {len(fl)} floating-point device tensors", flush=True)

    def run(idx):
        """poison keep[i] for i in idx, replay, return whether the flat gradient has non-finite entries"""
        for i in idx:
            keep[i].fill_(float("nan"))
        model.flat_gradients().zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        return not bool(torch.isfinite(model.flat_gradients()).all())

    run([])                                   # a clean replay first (replay 0 is the one that deviates when unpoisoned)
    if not run(fl):
        print("no NaN with EVERY buffer poisoned: no read-before-write among the capture-time allocations "
              "(the replay-0 deviation comes from a buffer allocated BEFORE the capture)", flush=True)
        return
    bad = []
    cand = list(fl)
    # several independent offenders are possible: peel them off one at a time by bisection
    for _ in range(6):
        if not run(cand):
            break
        lo, hi = 0, len(cand)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if run(cand[lo:mid]):
                hi = mid
            else:
                lo = mid
        i = cand[lo]
        if not run([i]):
            print(f"bisection ended on #{i} but it alone does not reproduce (interaction); stopping", flush=True)
            break
        bad.append(i)
        print(f"READ BEFORE WRITE: allocation #{i} of {len(keep)}  shape {tuple(keep[i].shape)} {keep[i].dtype}  allocated at {sites[i]}", flush=True)
        cand = [c for c in cand if c != i]
    print("offenders:", bad, flush=True)
    # who consumed an offender too early?  poison it alone, replay, and list the capture-time tensors that hold NaN afterwards, in
    # allocation order (forward allocations first): the first ones are the outputs of the kernel that read it before its writer ran
    for i in bad[:3]:
        for t in keep:
            if t.is_floating_point():
                t.zero_()
        run([])
        run([i])
        hit = [j for j in fl if j != i and not bool(torch.isfinite(keep[j]).all())]
        self_nan = not bool(torch.isfinite(keep[i]).all())
        print(f"poison #{i} alone -> after the replay #{i} itself {'STILL holds NaN (its writer never ran in the graph)' if self_nan else 'was overwritten (its writer ran - later than a reader)'}; "
              f"{len(hit)} other tensors hold NaN; first: " + "; ".join(f"#{j} {tuple(keep[j].shape)} {sites[j].split(' <- ')[0]}" for j in hit[:8]), flush=True)


if len(sys.argv) > 4 and sys.argv[4] == "poison":
    poison_search(DT)
else:
    compare()
