"""Op-level parity: every HIP kernel vs a plain PyTorch fp32/fp64 CPU computation of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from fastspeech2_amd import ops
    return ops


def rel_err(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def conv_ref(x, w, b, S, dil, pad, lens=None):
    """x: [B*S, Cin] rows; w: (Cout, Cin, k) -> [B*S, Cout] with arbitrary left pad (right pad implied by same length)."""
    B = x.shape[0] // S
    k = w.shape[2]
    xx = x.double().view(B, S, -1).transpose(1, 2)
    right = (k - 1) * dil - pad
    xx = F.pad(xx, (pad, max(right, 0)))
    y = F.conv1d(xx, w.double(), None if b is None else b.double(), dilation=dil)[:, :, :S]
    y = y.transpose(1, 2).reshape(B * S, -1)
    if lens is not None:
        t = torch.arange(S).unsqueeze(0)
        mask = (t >= lens.unsqueeze(1)).reshape(-1)
        y[mask] = 0
    return y


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,S,Cin,Cout,k,dil", [(3, 50, 256, 256, 1, 1), (2, 77, 256, 1024, 9, 1), (2, 130, 80, 512, 5, 1),
                                                (1, 300, 512, 80, 5, 1), (2, 64, 128, 128, 3, 3), (1, 200, 32, 8, 7, 1),
                                                (2, 33, 1024, 256, 1, 1), (1, 257, 64, 64, 11, 5)])
def test_conv_gemm_fwd(dev, dtype, tol, B, S, Cin, Cout, k, dil):
    ops = _ops()
    torch.manual_seed(0)
    x = torch.randn(B * S, Cin)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout)
    pad = dil * (k - 1) // 2
    lens = torch.tensor([S - 7 * i for i in range(B)], dtype=torch.int32)
    xd = x.to(dev).to(dtype)
    wf, wd = ops.pack_weight(w.permute(0, 2, 1).contiguous().to(dev), dtype)      # master weights are tap-major
    xr = xd.float().cpu()
    wr = wf.float().cpu().permute(0, 2, 1).contiguous()
    for use_lens in (False, True):
        y = ops.conv_gemm(xd, wf, b.to(dev), S, taps=k, dil=dil, pad=pad, act=ops.ACT_RELU,
                          lens=lens.to(dev) if use_lens else None)
        ref = torch.relu(conv_ref(xr, wr, b, S, dil, pad)).clone()
        if use_lens:
            t = torch.arange(S).unsqueeze(0)
            ref[(t >= lens.unsqueeze(1)).reshape(-1)] = 0
        assert rel_err(y.float(), ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,S,Cin,Cout,k,dil", [(2, 70, 256, 1024, 9, 1), (2, 61, 256, 256, 3, 1), (1, 140, 80, 512, 5, 1),
                                                (3, 40, 256, 768, 1, 1), (1, 300, 512, 80, 5, 1)])
def test_conv_grads(dev, dtype, tol, B, S, Cin, Cout, k, dil):
    ops = _ops()
    torch.manual_seed(1)
    pad = dil * (k - 1) // 2
    x = torch.randn(B * S, Cin)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    dy = torch.randn(B * S, Cout)
    xd, dyd = x.to(dev).to(dtype), dy.to(dev).to(dtype)
    wf, wd = ops.pack_weight(w.permute(0, 2, 1).contiguous().to(dev), dtype)
    xr = xd.float().cpu().double().requires_grad_(True)
    wr = wf.float().cpu().permute(0, 2, 1).contiguous().double().requires_grad_(True)
    y = conv_ref(xr, wr, None, S, dil, pad)
    y.backward(dyd.float().cpu().double())
    # dgrad: conv of dy with tap-flipped transposed weights, pad' = (k-1)*dil - pad
    dx = ops.conv_gemm(dyd, wd, None, S, taps=k, dil=dil, pad=(k - 1) * dil - pad)
    assert rel_err(dx.float(), xr.grad) < tol
    dw = torch.zeros(Cout, k, Cin, device=dev)                                      # tap-major gradient
    dbf = torch.zeros(Cout, device=dev)                                             # bias gradient fused into the same pass
    ops.conv_wgrad(dyd, xd, dw, S, taps=k, dil=dil, pad=pad, dbias=dbf)
    assert rel_err(dw.permute(0, 2, 1), wr.grad) < (tol if dtype == torch.float32 else 2e-2)
    assert rel_err(dbf, dyd.float().cpu().double().sum(0)) < 1e-4
    db = torch.zeros(Cout, device=dev)
    ops.colsum(dyd, db)
    assert rel_err(db, dyd.float().cpu().double().sum(0)) < 1e-4


def _wgrad_ref(dy, x, Bq, S, k, pad):
    xs, dys = x.float().view(Bq, S, -1).double(), dy.float().view(Bq, S, -1).double()
    ref = torch.zeros(dys.shape[-1], k, xs.shape[-1], device=dy.device, dtype=torch.float64)
    for j in range(k):
        sh = j - pad
        lo, hi = max(0, -sh), min(S, S - sh)
        if hi > lo:
            ref[:, j, :] = torch.einsum("bsn,bsc->nc", dys[:, lo:hi], xs[:, lo + sh:hi + sh])
    return ref


@pytest.mark.parametrize("Bq,S,Cin,Cout,k,pad", [
    (3, 200, 128, 128, 5, 2), (5, 64, 256, 128, 9, 4), (4, 65, 128, 256, 9, 4), (2, 37, 128, 128, 3, 1), (6, 130, 80, 200, 5, 2),
    (3, 131, 256, 80, 7, 3), (2, 500, 128, 128, 2, 0), (2, 129, 128, 128, 4, 1), (7, 63, 136, 128, 3, 1), (2, 300, 128, 128, 6, 2),
    (2, 150, 128, 128, 8, 3), (3, 96, 128, 128, 11, 5), (4, 925, 256, 256, 9, 4), (3, 90, 128, 128, 3, 2), (2, 128, 512, 128, 1, 0)])
def test_weight_gradient_workspace_path_edge_cases(dev, Bq, S, Cin, Cout, k, pad):
    """The round-3 split-K weight gradient (fs2_conv_wgrad_ws: LDS-DMA tap-group kernel for k >= 2, slab stores + finalize for all)
    against fp64: sequences shorter / exactly / just longer than one 64-row K-tile, tap groups 2 / 3 / 4 / 5 / 5+4 / 4+3 / 3+3 /
    4+4 / 4+4+3, asymmetric padding, N / Cin that are not multiples of 128 (clamped DMA columns), ragged lens with EMPTY sequences
    (skipped K-tiles, rows zeroed in LDS), += into a non-zero gradient, a NaN-poisoned workspace (every slab word that is read
    was written), bit-reproducibility, and agreement with the atomic path."""
    ops = _ops()
    g = torch.Generator().manual_seed(S * 7 + k)
    M = Bq * S
    x = torch.randn(M, Cin, generator=g).to(dev).to(torch.bfloat16)
    dy_full = torch.randn(M, Cout, generator=g).to(dev)
    lens = torch.randint(0, S + 1, (Bq,), generator=g).to(torch.int32)
    lens[0] = S
    if Bq > 2:
        lens[1] = 0
    for use_lens in (False, True):
        ld = lens.to(dev) if use_lens else None
        valid = (torch.arange(S, device=dev).unsqueeze(0) < lens.to(dev).unsqueeze(1)).reshape(-1, 1) if use_lens else 1.0
        dy = (dy_full * valid).to(torch.bfloat16)                              # contract: gradient rows >= lens are zero
        ref = _wgrad_ref(dy, x, Bq, S, k, pad)
        bref = dy.float().double().sum(0)
        scale, bscale = max(ref.abs().max().item(), 1e-6), max(bref.abs().max().item(), 1e-6)
        init = torch.randn(Cout, k, Cin, generator=g).to(dev)
        binit = torch.randn(Cout, generator=g).to(dev)
        outs = []
        for rep in range(2):
            for key, ws in list(ops._default_wgrad_ws._ws.items()):
                ws.fill_(float("nan"))
            dw, db = init.clone(), binit.clone()
            ops.conv_wgrad(dy, x, dw, S, taps=k, pad=pad, lens=ld, dbias=db)
            if rep == 0:                                                   # (first call may have just created the workspace: poison + redo)
                for key, ws in list(ops._default_wgrad_ws._ws.items()):
                    ws.fill_(float("nan"))
                dw, db = init.clone(), binit.clone()
                ops.conv_wgrad(dy, x, dw, S, taps=k, pad=pad, lens=ld, dbias=db)
            outs.append((dw, db))
            err = (dw.double() - init.double() - ref).abs().max().item()
            assert err <= 2e-5 * scale + 2e-6, (use_lens, err, scale)
            assert ((db.double() - binit.double() - bref).abs().max().item()) <= 2e-5 * bscale + 2e-6
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # no atomics: bit-reproducible
        dwa, dba = init.clone(), binit.clone()
        ops.conv_wgrad(dy, x, dwa, S, taps=k, pad=pad, lens=ld, dbias=dba, use_ws=False)         # round-1/2 atomic kernels
        assert (dwa.double() - init.double() - ref).abs().max().item() <= 2e-5 * scale + 2e-6


def attn_ref(qkv, lens, B, S, H):
    dk = 128
    q, k, v = qkv.view(B, S, 3, H, dk).unbind(2)
    q, k, v = [t.permute(0, 2, 1, 3) for t in (q, k, v)]  # B,H,S,dk
    s = q @ k.transpose(-1, -2) / math.sqrt(dk)
    mask = torch.arange(S).view(1, 1, 1, S) >= lens.view(B, 1, 1, 1)
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * dk)
    return o


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2e-2)])
# (10, 200, 2) / (5, 130, 2): 20 / 10 (sequence, head) pairs = full groups of 8 AND a shorter last group in the XCD-aware block map
@pytest.mark.parametrize("B,S,H", [(2, 100, 2), (3, 257, 2), (1, 31, 1), (2, 640, 2), (10, 200, 2), (5, 130, 2)])
def test_attention(dev, dtype, tol, B, S, H):
    ops = _ops()
    torch.manual_seed(2)
    qkv = torch.randn(B * S, 3 * H * 128)
    lens = torch.tensor([max(S - 13 * i, 7) for i in range(B)], dtype=torch.int32)
    qd = qkv.to(dev).to(dtype)
    qr = qd.float().cpu().double().requires_grad_(True)
    ref = attn_ref(qr, lens, B, S, H)
    ctx, lse = ops.attn_fwd(qd, lens.to(dev), B, S, H)
    valid = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).reshape(-1)
    assert rel_err(ctx.float().cpu()[valid], ref.detach()[valid]) < tol
    dctx = torch.randn(B * S, H * 128)
    dctx[~valid] = 0
    dd = dctx.to(dev).to(dtype)
    ref.backward(dd.float().cpu().double())
    dqkv = ops.attn_bwd(qd, ctx, dd, lse, lens.to(dev), B, S, H)
    g = qr.grad.clone()
    assert rel_err(dqkv.float(), g) < (tol * 3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_layernorm(dev, dtype, tol):
    ops = _ops()
    torch.manual_seed(3)
    B, S, C = 3, 45, 256
    y = torch.randn(B * S, C); res = torch.randn(B * S, C)
    gamma = torch.randn(C); beta = torch.randn(C)
    lens = torch.tensor([45, 30, 17], dtype=torch.int32)
    yd, rd = y.to(dev).to(dtype), res.to(dev).to(dtype)
    yr = yd.float().cpu().double().requires_grad_(True)
    rr = rd.float().cpu().double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    z = yr + rr
    if dtype == torch.bfloat16:
        z = z + (z.detach().float().bfloat16().double() - z.detach())   # kernel normalises the stored (rounded) z
    ref = F.layer_norm(z, (C,), gr, br, 1e-5)
    pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).reshape(-1)
    ref = ref.masked_fill(pad.unsqueeze(1), 0)
    out, mean, rstd = ops.ln_fwd(yd, rd, gamma.to(dev), beta.to(dev), lens.to(dev), B, S)
    assert rel_err(out.float(), ref.detach()) < tol
    dout = torch.randn(B * S, C)
    dd = dout.to(dev).to(dtype)
    ref.backward(dd.float().cpu().double())
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    d1, _ = ops.ln_bwd(yd, dd, gamma.to(dev), lens.to(dev), mean, rstd, dg, db, B, S)
    assert rel_err(d1.float(), yr.grad) < tol * 2
    assert rel_err(dg, gr.grad) < tol * 2
    assert rel_err(db, br.grad) < tol * 2
    # the upstream gradient as a pair dout + dout2 (fs2_ln_bwd_sum): fp32 adds the same two numbers the caller would have added -
    # bit-identical to the summed call; bf16 adds the two STORED values in fp32 (no rounding of the sum)
    half = (0.5 * dout).to(dev).to(dtype)
    dg2 = torch.zeros(C, device=dev); db2 = torch.zeros(C, device=dev)
    p1, _ = ops.ln_bwd(yd, half, gamma.to(dev), lens.to(dev), mean, rstd, dg2, db2, B, S, dout2=half)
    if dtype == torch.float32:
        dg1 = torch.zeros(C, device=dev); db1 = torch.zeros(C, device=dev)
        s1, _ = ops.ln_bwd(yd, half + half, gamma.to(dev), lens.to(dev), mean, rstd, dg1, db1, B, S)
        assert torch.equal(p1, s1)
    zz = z.detach().clone().requires_grad_(True)
    ref2 = F.layer_norm(zz, (C,), gr.detach(), br.detach(), 1e-5).masked_fill(pad.unsqueeze(1), 0)
    ref2.backward(2 * half.float().cpu().double())
    assert rel_err(p1.float(), zz.grad) < tol * 2


def test_layernorm_dropout_consistency(dev):
    """dropout masks are regenerated in backward: zero pattern of d2 must equal the forward keep-mask."""
    ops = _ops()
    torch.manual_seed(4)
    B, S, C = 2, 40, 256
    y = torch.randn(B * S, C, device=dev) + 3.0
    res = torch.zeros(B * S, C, device=dev)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    y0 = y.clone()
    out, mean, rstd = ops.ln_fwd(y, res, g, b, None, B, S, p_pre=0.2, seed_pre=1234)
    keep = (y != 0)                       # y now holds z = drop(y0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.8) < 0.02
    assert torch.allclose(y[keep], y0[keep] / 0.8, rtol=1e-6)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    dout = torch.randn(B * S, C, device=dev)
    d1, d2 = ops.ln_bwd(y, dout, g, None, mean, rstd, dg, db, B, S, want_d2=True, p_pre=0.2, seed_pre=1234)
    assert torch.equal(d2 != 0, keep & (d1 != 0))
    assert torch.allclose(d2[keep], d1[keep] / 0.8, rtol=1e-6)
    # post-LN dropout + relu backward (variance predictor form)
    y = torch.relu(torch.randn(B * S, C, device=dev))
    out, mean, rstd = ops.ln_fwd(y, None, g, b, None, B, S, p_post=0.5, seed_post=77)
    keep = out != 0
    assert abs(keep.float().mean().item() - 0.5) < 0.03
    ref = F.layer_norm(y, (C,)) * 2.0
    assert torch.allclose(out[keep], ref[keep], rtol=1e-4, atol=1e-5)


def test_batchnorm_workspace_size_is_checked(dev):
    """the reducing BatchNorm entry points grew their workspace from 2C to fs2_bn_ws_floats(C) floats in round 4 under
    unchanged names; they now take the size and refuse a small buffer instead of writing past it."""
    from fastspeech2_amd import _lib, ops
    C, M = 80, 512
    x = torch.randn(M, C, device=dev)
    small = torch.zeros(2 * C, device=dev)
    with pytest.raises(ValueError, match="workspace"):
        _lib.call("fs2_bn_stats", x.data_ptr(), small.data_ptr(), small.numel(), M, C, ops.F32, ops._stream())
    ok = ops.bn_workspace(C, dev)
    assert ok.numel() == _lib.load().fs2_bn_ws_floats(C) > 2 * C
    _lib.call("fs2_bn_stats", x.data_ptr(), ok.data_ptr(), ok.numel(), M, C, ops.F32, ops._stream())
    torch.cuda.synchronize()
    assert torch.allclose(ok[:C].cpu(), x.sum(0).cpu(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_batchnorm(dev, dtype, tol):
    ops = _ops()
    torch.manual_seed(5)
    M, C = 700, 512
    x = torch.randn(M, C) * 2 + 0.5
    gamma = torch.rand(C) + 0.5; beta = torch.randn(C) * 0.1
    xd = x.to(dev).to(dtype)
    xr = xd.float().cpu().double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    rm = torch.zeros(C, dtype=torch.double); rv = torch.ones(C, dtype=torch.double)
    ref = torch.tanh(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    rmd = torch.zeros(C, device=dev); rvd = torch.ones(C, device=dev)
    out, mean_rstd = ops.bn_train_fwd(xd, gamma.to(dev), beta.to(dev), rmd, rvd, ops.ACT_TANH, 0.0, 0)
    assert rel_err(out.float(), ref.detach()) < tol
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 1e-4
    dout = torch.randn(M, C)
    dd = dout.to(dev).to(dtype)
    ref.backward(dd.float().cpu().double())
    dx, dgam, dbet = ops.bn_bwd(xd, dd, mean_rstd, gamma.to(dev), beta.to(dev), ops.ACT_TANH, 0.0, 0)
    assert rel_err(dx.float(), xr.grad) < tol * 3
    assert rel_err(dgam, gr.grad) < tol * 3
    assert rel_err(dbet, br.grad) < tol * 3


def lr_ref(x, dur, max_len):
    """Restatement of reference LengthRegulator (model/modules.py:167-194) in plain python."""
    outs, lens = [], []
    for xb, db in zip(x, dur):
        rows = []
        for i in range(xb.shape[0]):
            n = max(int(db[i].item()), 0)
            rows.append(xb[i:i + 1].expand(n, -1))
        e = torch.cat(rows, 0)
        lens.append(e.shape[0])
        if e.shape[0] >= max_len:
            e = e[:max_len]
        else:
            e = F.pad(e, (0, 0, 0, max_len - e.shape[0]))
        outs.append(e)
    return torch.stack(outs), torch.tensor(lens, dtype=torch.int64)


@pytest.mark.parametrize("is_float", [False, True])
def test_length_regulator_bit_exact(dev, is_float):
    ops = _ops()
    torch.manual_seed(6)
    B, L, C = 4, 37, 256
    x = torch.randn(B, L, C)
    if is_float:
        dur = (torch.rand(B, L) * 9 - 1.0)           # includes negatives and fractions (trunc toward zero)
        dur[0, 3] = 2.5; dur[0, 4] = -0.5; dur[1, 0] = 0.999
    else:
        dur = torch.randint(0, 10, (B, L))
    dur[2, 20:] = 0
    for T in (int(dur.clamp(min=0).long().sum(1).max().item()), 150, 60):
        ref, ref_len = lr_ref(x, dur, T)
        cum, idx, mel_len = ops.lr_index(dur.to(dev), T)
        out = ops.lr_gather_fwd(x.to(dev).view(B * L, C), idx, None, B, L, T)
        assert torch.equal(mel_len.cpu(), ref_len)
        assert torch.equal(out.cpu().view(B, T, C), ref)          # payload copied verbatim: bit-exact
        # backward = segment sum
        dy = torch.randn(B, T, C)
        xr = x.clone().requires_grad_(True)
        r2, _ = lr_ref(xr, dur, T)
        r2.backward(dy)
        dx = ops.lr_gather_bwd(dy.to(dev).view(B * T, C), cum, B, L, T)
        assert torch.allclose(dx.cpu().view(B, L, C), xr.grad, atol=1e-5)


def test_embed_bucket_rowdot(dev):
    ops = _ops()
    torch.manual_seed(7)
    B, L, C, V = 3, 20, 256, 361
    tok = torch.randint(1, V, (B, L)); tok[1, 15:] = 0
    emb = torch.randn(V, C); emb[0] = 0
    pe = torch.randn(64, C)
    out = ops.embed_pe_fwd(tok.to(dev), emb.to(dev), pe.to(dev), torch.float32)
    ref = emb[tok] + pe[:L].unsqueeze(0)
    assert torch.equal(out.cpu().view(B, L, C), ref)
    dy = torch.randn(B * L, C)
    demb = torch.zeros(V, C, device=dev)
    ops.embed_bwd(tok.to(dev), dy.to(dev), demb)
    r = torch.zeros(V, C).index_add_(0, tok.view(-1), dy); r[0] = 0
    assert torch.allclose(demb.cpu(), r, atol=1e-5)
    # bucketize
    bins = torch.linspace(-2.9, 11.3, 255)
    vals = torch.cat([torch.randn(B * L - 4) * 3, bins[[0, 10, 254]], torch.tensor([100.0])])
    table = torch.randn(256, C)
    x = torch.randn(B * L, C)
    o, idx = ops.bucket_embed_add_fwd(x.to(dev), vals.to(dev), 1.0, bins.to(dev), table.to(dev))
    ridx = torch.bucketize(vals, bins)
    assert torch.equal(idx.cpu().long(), ridx)
    assert torch.equal(o.cpu(), x + table[ridx])
    # rowdot
    w = torch.randn(C); bb = torch.randn(1)
    lens = torch.tensor([20, 11, 5], dtype=torch.int32)
    y = ops.rowdot_fwd(x.to(dev), w.to(dev), bb.to(dev), lens.to(dev), B, L)
    pad = torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)
    ry = (x @ w + bb).view(B, L).masked_fill(pad, 0)
    assert torch.allclose(y.cpu(), ry, atol=1e-4)
    g = torch.randn(B, L)
    dw = torch.zeros(C, device=dev); db = torch.zeros(1, device=dev)
    dx = ops.rowdot_bwd(x.to(dev), w.to(dev), g.to(dev), lens.to(dev), dw, db, B, L)
    gm = g.masked_fill(pad, 0).view(-1)
    assert torch.allclose(dx.cpu(), gm.unsqueeze(1) * w, atol=1e-5)
    assert torch.allclose(dw.cpu(), gm @ x, atol=1e-3)
    assert torch.allclose(db.cpu(), gm.sum().view(1), atol=1e-4)


def test_duration_round(dev):
    ops = _ops()
    logd = torch.log(torch.tensor([1.5, 2.5, 3.5, 4.5, 0.2, 1.0, 7.49, 7.51]) + 1)
    for ctl in (1.0, 0.8, 1.3):
        out = ops.duration_round(logd.to(dev), ctl).cpu()
        ref = torch.clamp(torch.round(torch.exp(logd) - 1) * ctl, min=0)
        assert torch.equal(out, ref)


def test_duration_round_boundary_sweep(dev):
    """the inference-path index contract (model/modules.py:132-135: round(exp(log_d) - 1) * d_control, clamped)
    at every place it can flip - the 33 fp32 values around log(k + 1.5) (where exp(x) - 1 crosses k + 0.5) and around log(k + 1)
    (exact integers) for every k <= 64 - against the reference's own CPU arithmetic (torch.exp / torch.round in fp32)."""
    ops = _ops()
    import numpy as np
    xs = []
    for k in range(0, 65):
        for c in (k + 1.5, k + 1.0):
            x0 = np.float32(np.log(np.float64(c)))
            lo = hi = x0
            xs.append(x0)
            for _ in range(16):
                lo = np.nextafter(lo, np.float32(-np.inf)); hi = np.nextafter(hi, np.float32(np.inf))
                xs += [lo, hi]
    logd = torch.from_numpy(np.array(xs, dtype=np.float32))
    n_diff = 0
    for ctl in (1.0, 0.8, 1.3):
        out = ops.duration_round(logd.to(dev), ctl).cpu()
        ref = torch.clamp(torch.round(torch.exp(logd) - 1) * ctl, min=0)
        n_diff += int((out != ref).sum())
    assert n_diff == 0, (n_diff, logd.numel())


def test_adam_matches_torch(dev):
    ops = _ops()
    torch.manual_seed(8)
    n = 10007 * 4
    p = torch.randn(n); g = torch.randn(n) * 3
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        pr.grad = g.clone() * step
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = (g * step).to(dev)
        nsq = torch.zeros(1, device=dev)
        ops.sumsq(gd, nsq)
        hyper = torch.tensor([1e-3, 1 - 0.9 ** step, 1 - 0.98 ** step], device=dev)
        ops.adam_step(pd, gd, m, v, nsq, 1.0, hyper, 0.9, 0.98, 1e-9, 0.0)
    assert torch.allclose(pd.cpu(), pr.detach(), atol=2e-6)


@pytest.mark.parametrize("B,S,Cin,Cout,k,dil", [(9, 950, 64, 1024, 9, 1), (8, 1024, 128, 1024, 1, 1), (11, 777, 64, 768, 5, 2),
                                                (40, 900, 64, 250, 3, 1)])
def test_conv_gemm_big_tile_bf16(dev, B, S, Cin, Cout, k, dil):
    """shapes large enough to take the 256x256-tile kernel (>= 128 tiles): ragged lens incl. fully padded tiles,
    M / N not multiples of 256, residual, accumulate + out_scale, ReLU-gate epilogue."""
    ops = _ops()
    torch.manual_seed(3)
    dtype, tol = torch.bfloat16, 2e-2
    x = torch.randn(B * S, Cin)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout)
    pad = dil * (k - 1) // 2
    lens = torch.tensor([max(S - 97 * i, 3) for i in range(B)], dtype=torch.int32)
    xd = x.to(dev).to(dtype)
    wf, _ = ops.pack_weight(w.permute(0, 2, 1).contiguous().to(dev), dtype)
    xr = xd.float().cpu()
    wr = wf.float().cpu().permute(0, 2, 1).contiguous()
    base = conv_ref(xr, wr, b, S, dil, pad)
    padmask = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).reshape(-1)
    # plain + relu, with and without lens
    for use_lens in (False, True):
        y = ops.conv_gemm(xd, wf, b.to(dev), S, taps=k, dil=dil, pad=pad, act=ops.ACT_RELU, lens=lens.to(dev) if use_lens else None)
        ref = torch.relu(base).clone()
        if use_lens:
            ref[padmask] = 0
        assert rel_err(y.float(), ref) < tol
    # residual + out_scale + accumulate
    res = torch.randn(B * S, Cout).to(dtype)
    y0 = torch.randn(B * S, Cout).to(dtype)
    out = y0.clone().to(dev)
    ops.conv_gemm(xd, wf, b.to(dev), S, taps=k, dil=dil, pad=pad, res=res.to(dev), out=out, accumulate=True, out_scale=1.0 / 3)
    ref = (base + res.double()) / 3 + y0.double()
    assert rel_err(out.float(), ref) < tol
    # ReLU gate: y = (R > 0) ? conv : 0
    g = ops.conv_gemm(xd, wf, None, S, taps=k, dil=dil, pad=pad, act=ops.ACT_GATE, res=res.to(dev))
    ref = torch.where(res.double() > 0, conv_ref(xr, wr, None, S, dil, pad), torch.zeros(1, dtype=torch.float64))
    assert rel_err(g.float(), ref) < tol


@pytest.mark.parametrize("rows,C,hot", [(60, 256, False), (6144, 256, True), (3000, 512, True)])
def test_bucket_embed_bwd(dev, rows, C, hot):
    """gather-reduce per bin (with a hot bin holding every padded position, as zero pitch targets do) vs index_add_."""
    ops = _ops()
    torch.manual_seed(21)
    nb = 256
    idx = torch.randint(0, nb, (rows,), dtype=torch.int32)
    if hot:
        idx[torch.rand(rows) < 0.3] = 52
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 1e-4)):
        dy = torch.randn(rows, C).to(dtype)
        demb = torch.ones(nb, C, device=dev)                     # accumulates INTO the gradient buffer
        ops.bucket_embed_bwd(idx.to(dev), dy.to(dev), demb)
        ref = torch.ones(nb, C, dtype=torch.double).index_add_(0, idx.long(), dy.double())
        assert rel_err(demb, ref) < tol


def test_pack_dgrad_multi_matches_single(dev):
    ops = _ops()
    torch.manual_seed(22)
    shapes = [(1024, 256, 9), (256, 1024, 1), (80, 256, 1), (512, 80, 5), (768, 256, 1), (100, 72, 3)]
    for dtype in (torch.float32, torch.bfloat16):
        chunks, rows, off, wdo, tile0 = [], [], 0, 0, 0
        for (co, ci, k) in shapes:
            w = torch.randn(co, k, ci)
            chunks.append(w.reshape(-1))
            rows.append([off, wdo, co, ci, k, tile0])
            n = co * ci * k
            off += n; wdo += (n + 7) // 8 * 8
            tile0 += ((co + 63) // 64) * ((ci + 63) // 64) * k
        flat = torch.cat(chunks).to(dev)
        wd_all = torch.zeros(wdo, device=dev, dtype=dtype)
        ops.pack_dgrad_multi(flat, wd_all, torch.tensor(rows, dtype=torch.int64, device=dev), tile0)
        for (co, ci, k), r in zip(shapes, rows):
            w = flat[r[0]:r[0] + co * ci * k].view(co, k, ci)
            _, wd = ops.pack_weight(w, dtype)
            got = wd_all[r[1]:r[1] + co * ci * k].view(ci, k, co)
            assert torch.equal(got, wd), (co, ci, k, dtype)


def test_adam_fused_shadow_and_zero_grad(dev):
    ops = _ops()
    torch.manual_seed(23)
    n = 4096 * 8
    p = torch.randn(n, device=dev); g = torch.randn(n, device=dev)
    p2, g2 = p.clone(), g.clone()
    m, v, m2, v2 = (torch.zeros(n, device=dev) for _ in range(4))
    nsq = torch.zeros(1, device=dev)
    ops.sumsq(g, nsq)
    hyper = torch.tensor([1e-3, 0.1, 0.02, 0.0], device=dev)
    ops.adam_step(p, g, m, v, nsq, 1.0, hyper, 0.9, 0.98, 1e-9, 0.0)
    lp = torch.empty(n, device=dev, dtype=torch.bfloat16)
    ops.adam_step(p2, g2, m2, v2, nsq, 1.0, hyper, 0.9, 0.98, 1e-9, 0.0, p_lowp=lp, zero_grad=True)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert torch.equal(lp, p2.to(torch.bfloat16))
    assert torch.count_nonzero(g2).item() == 0 and torch.count_nonzero(g).item() > 0


@pytest.mark.parametrize("M,C", [(1000, 80), (333, 512), (64, 4)])
def test_batchnorm_shapes(dev, M, C):
    """row-chunk mapping at channel counts that are not a power of two / tiny, large column means (shifted sums)."""
    ops = _ops()
    torch.manual_seed(24)
    x = torch.randn(M, C) * 0.5 + 30.0                          # |mean| >> std: a naive E[x^2]-E[x]^2 would cancel
    gamma = torch.rand(C) + 0.5; beta = torch.randn(C) * 0.1
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    rm = torch.zeros(C, dtype=torch.double); rv = torch.ones(C, dtype=torch.double)
    res = torch.randn(M, C)
    ref = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5) + res.double()
    rmd = torch.zeros(C, device=dev); rvd = torch.ones(C, device=dev)
    out, mean_rstd = ops.bn_train_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rmd, rvd, ops.ACT_NONE, 0.0, 0, res=res.to(dev))
    assert rel_err(out, ref.detach()) < 2e-4
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 2e-3
    dout = torch.randn(M, C)
    ref.backward(dout.double())
    dx, dgam, dbet = ops.bn_bwd(x.to(dev), dout.to(dev), mean_rstd, gamma.to(dev), beta.to(dev), ops.ACT_NONE, 0.0, 0)
    assert rel_err(dx, xr.grad) < 2e-3
    assert rel_err(dgam, gr.grad) < 2e-3 and rel_err(dbet, br.grad) < 1e-4


@pytest.mark.parametrize("B,S,C,N,k,dil", [(48, 925, 128, 128, 11, 1), (12, 4000, 256, 256, 3, 3), (24, 2000, 1024, 256, 1, 1)])
def test_conv_gemm_big_tile_lrelu_prologue(dev, B, S, C, N, k, dil):
    """leaky-ReLU prologue on the 256x128 ring kernel (applied to the fragments after the LDS-DMA) == activation applied
    beforehand in bf16 (HiFi-GAN's pre-activation convolutions at the 128/256-channel stages)."""
    ops = _ops()
    torch.manual_seed(41)
    M = B * S
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, k, C, device=dev) / (C * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    pad = dil * (k - 1) // 2
    got = ops.conv_gemm(x, w, b, S, taps=k, dil=dil, pad=pad, in_act=ops.ACT_LRELU, in_slope=0.1)
    xl = torch.nn.functional.leaky_relu(x.float(), 0.1).to(torch.bfloat16)
    ref = ops.conv_gemm(xl, w, b, S, taps=k, dil=dil, pad=pad)
    # same kernel for both calls (the multi-tap shapes): same accumulation order, same bf16 rounding of the activation -> equal
    # bits.  The one-tap shape takes the persistent kernel WITH the prologue and the wide one-tap kernel (K-steps of 32) without
    # it: same products, different fp32 summation order -> at most one bf16 ulp apart.
    if k > 1:
        assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    else:
        d = (got.float() - ref.float()).abs()
        assert bool((d <= 2.0 ** -7 * ref.float().abs() + 1e-3).all()), d.max().item()
