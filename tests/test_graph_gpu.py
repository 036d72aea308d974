"""hipGraph replay of the train step computes what the eager step computes: `bench.py --graph 1` captures
forward + loss + backward + clip/Adam once and replays it.  Two host-side state machines used to be baked into the capture -
the dropout position (every replay drew the SAME masks) and the BatchNorm backward's ping-pong workspace parity (the C = 80
layer reduced into a never-cleared workspace from the second replay on).  Here 3 eager warm-up steps + 4 replays are compared
with 7 eager steps of an identically seeded model: per-step losses, Adam's first moments (linear in every gradient of every
step), BatchNorm buffers.  side_stream = 1: the EAGER steps run with the weight-gradient side stream (the default of train.py
and bench.py); bench.capture_graph captures on one stream by default.  fork=True forks the side stream INSIDE the capture: rounds
3-5 measured that 2.8e-3 off (3.5e-2 on the first replay), round 6 found the missing edge (Engine._wgrad) and
test_forked_capture_* below holds the forked capture to the eager gradient and to a NaN-poisoned replay."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(dev, use_graph, side_stream):
    import bench
    args = bench.parse(["--batch", "8", "--phonemes", "40", "--dtype", "fp32", "--side-stream", str(side_stream)])
    torch.manual_seed(1234)
    model, loss_fn, opt, b, _, _ = bench.build(args, dev, 0, 1)
    model._engine.device_seed = True                      # what bench.py sets for --graph 1, before its warm-up
    model._engine.reseed(seed=4242, rank=0)
    step, fwd_bwd = bench.make_step(model, loss_fn, opt, b, None)
    losses = []
    for _ in range(3):
        losses.append(step())
    torch.cuda.synchronize()
    if use_graph:
        graph, static_loss, replay = bench.capture_graph(model, opt, fwd_bwd, fork=(use_graph == "fork"))
        for _ in range(4):
            replay()
            losses.append(static_loss.clone())
    else:
        for _ in range(4):
            losses.append(step())
    torch.cuda.synchronize()
    bn = {k: v.detach().clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    return torch.stack([l.float() for l in losses]).cpu(), opt._m.clone(), model.flat_parameters().clone(), bn


@pytest.mark.parametrize("side_stream,fork", [(0, False), (1, False), (1, True)])
def test_graph_replay_matches_eager_steps(dev, side_stream, fork):
    le, me, pe, bne = _run(dev, False, side_stream)
    le2, me2, pe2, _ = _run(dev, False, side_stream)
    lg, mg, pg, bng = _run(dev, "fork" if fork else True, side_stream)
    assert torch.allclose(le, lg, rtol=2e-5, atol=1e-5), (le, lg)       # fresh dropout masks on every replay, same sequence as eager
    # Adam's first moments = every gradient of every step.  Two EAGER runs of the same seeded steps already differ (fp32 weight
    # gradients are summed with atomics; observed 2e-4 .. 1e-3 after 7 steps, but on some boxes two eager runs schedule
    # identically and agree to 7e-7 while the captured single-stream run still sits 7e-4 away): the graph run must sit within a
    # few times the largest run-to-run distance seen - the two bugs this test was written for (identical masks on every replay,
    # BatchNorm sums accumulating across replays) move it by O(0.1 - 1).
    noise = float((me2 - me).norm() / me.norm())
    rel = float((mg - me).norm() / me.norm())
    print(f"side_stream={side_stream}: eager-vs-eager {noise:.2e}, graph-vs-eager {rel:.2e}")
    # round 4: with bit-reproducible BatchNorm sums two eager runs agree to 4e-7 .. 5e-5 and the captured run sits 1e-6 .. 7e-4 from
    # them (fp32 atomics of the one-tap weight gradients, amplified by Adam's g / sqrt(v) on near-zero gradients over 7 steps:
    # tools/dbg_single_stream.py, profiles/r04s_dbg_single.log - 5e-7 after ONE step in every stream mode).  The bar sits below the
    # 2.8e-3 a forked capture produced in round 3.
    assert rel < 4 * noise + 1.5e-3, (rel, noise)
    for k in bne:
        assert torch.allclose(bne[k].float(), bng[k].float(), rtol=1e-5, atol=1e-6), k
    assert float((pg - pe).abs().max()) < 1e-5                          # 7 warm-up-rate Adam steps: |dp| <= ~2e-6 each
    # the per-step losses really differ from step to step (a replay of identical masks / stale sums would not show above otherwise)
    assert len({round(float(x), 4) for x in le}) >= 5


def _one_step_setup(dev, side):
    import bench
    args = bench.parse(["--batch", "8", "--phonemes", "40", "--dtype", "fp32", "--side-stream", str(side)])
    torch.manual_seed(1234)
    model, loss_fn, opt, b, _, _ = bench.build(args, dev, 0, 1)
    model.disable_dropout = True
    model._engine.device_seed = True
    _, fwd_bwd = bench.make_step(model, loss_fn, opt, b, None)
    return model, opt, fwd_bwd


def test_forked_capture_gradient_equals_eager_and_survives_poison(dev):
    """VERDICT r05 next 7.  ONE step's flat gradient (fp32, dropout off) from a hipGraph whose capture FORKS the weight-gradient /
    variance-predictor side stream, against the eager one-stream step: <= 1e-5 relative on EVERY replay including the first (the
    bug showed 3.5e-2 on replay 0 and looked fine afterwards: the early reader picked up the previous replay's identical values).
    Then the structural check that found it: every tensor allocated during the capture is kept alive and filled with NaN before
    a replay - a kernel that reads a buffer before this replay's producer wrote it (a missing edge between the two captured
    streams) turns the gradient NaN."""
    model, opt, fwd_bwd = _one_step_setup(dev, 0)
    opt.zero_grad()
    fwd_bwd()
    torch.cuda.synchronize()
    ref = model.flat_gradients().clone()

    model, opt, fwd_bwd = _one_step_setup(dev, 1)
    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    keep = []
    real_empty, real_like = torch.empty, torch.empty_like

    def e(*a, **k):
        t = real_empty(*a, **k); keep.append(t); return t

    def el(*a, **k):
        t = real_like(*a, **k); keep.append(t); return t
    g = torch.cuda.CUDAGraph()
    opt.zero_grad()
    torch.cuda.synchronize()
    model._engine.fork_in_capture = True
    torch.empty, torch.empty_like = e, el
    try:
        with torch.cuda.graph(g):
            fwd_bwd()
    finally:
        torch.empty, torch.empty_like = real_empty, real_like
        model._engine.fork_in_capture = False
    torch.cuda.synchronize()
    assert model._engine._side_stream is not None           # the capture really had a second stream in it
    for r in range(3):
        model.flat_gradients().zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        rel = float((model.flat_gradients() - ref).norm() / ref.norm())
        assert rel < 1e-5, (r, rel)
    fl = [t for t in keep if t.is_cuda and t.is_floating_point()]
    assert len(fl) > 100
    for t in fl:
        t.fill_(float("nan"))
    model.flat_gradients().zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    got = model.flat_gradients()
    assert bool(torch.isfinite(got).all()), "a captured kernel read a capture-time buffer before its producer ran"
    assert float((got - ref).norm() / ref.norm()) < 1e-5
