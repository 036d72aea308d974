"""Tensor-level wrappers over the C ABI (include/fs2hip.h).

PyTorch is used only as plumbing: device memory (torch.empty), the current HIP stream and dtype bookkeeping.
Every function launches hand-written HIP kernels through ctypes; nothing here computes with torch ops.
"""
import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU, ACT_GATE = 0, 1, 2, 3, 4

# bench.py instrumentation: when a dict, every conv_gemm launch is bracketed by HIP events on its own stream and
# recorded as (algorithmic FLOPs, start, end).  None in normal operation (zero overhead).
PROFILE = None


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {d}")


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "fs2 ops need device tensors (no CPU fallback)"
    return t.data_ptr()


_tls = __import__("threading").local()


def _stream():
    """HIP stream handle the next launch goes to.  torch.cuda.current_stream() costs ~8 us per call (r01i host profile:
    ~3 ms per train step for ~400 launches), so the engine pins the handle for the duration of a forward / backward with
    `pinned_stream`; outside such a block the current torch stream is looked up per call."""
    h = getattr(_tls, "handle", None)
    return h if h is not None else torch.cuda.current_stream().cuda_stream


class pinned_stream:
    """with pinned_stream(): ... — launches inside use the torch stream that is current on ENTRY (or `stream` if given),
    without further lookups.  Nestable (the side-stream sections of backward nest inside the main one)."""

    def __init__(self, stream=None):
        self.stream = stream

    def __enter__(self):
        self.prev = getattr(_tls, "handle", None)
        s = self.stream if self.stream is not None else torch.cuda.current_stream()
        _tls.handle = s.cuda_stream
        return self

    def __exit__(self, *exc):
        _tls.handle = self.prev
        return False


def _contig(t):
    assert t.is_contiguous(), "fs2 ops need contiguous tensors"
    return t


# ------------------------------------------------------------------ weights
def pack_weight(w, dtype, want_fwd=True, want_dgrad=True, wf=None, wd=None):
    """w: tap-major master weight (Cout, k, Cin) or (N, K) fp32 -> (Wf [Cout][k][Cin], Wd [Cin][k][Cout]) in `dtype`.
    With dtype == float32 and no explicit output buffer, Wf is w itself (no copy)."""
    w = _contig(w)
    if w.dim() == 2:
        cout, cin, k = w.shape[0], w.shape[1], 1
    else:
        cout, k, cin = w.shape
    if want_fwd and wf is None and dtype == torch.float32:
        wf, want_fwd_copy = w.view(cout, k, cin), False
    else:
        want_fwd_copy = want_fwd
    if want_fwd and wf is None:
        wf = torch.empty(cout, k, cin, device=w.device, dtype=dtype)
    if want_dgrad and wd is None:
        wd = torch.empty(cin, k, cout, device=w.device, dtype=dtype)
    if want_fwd_copy or want_dgrad:
        _lib.call("fs2_pack_weight", _p(w), _p(wf) if want_fwd_copy else None, _p(wd) if want_dgrad else None, cout, cin, k,
                  dt(dtype), _stream())
    return wf, wd


# ------------------------------------------------------------------ contraction
def conv_gemm(x, wpacked, bias, S, taps=1, dil=1, pad=0, act=ACT_NONE, slope=0.0, lens=None, res=None, out=None,
              in_act=ACT_NONE, in_slope=0.0, accumulate=False, out_scale=1.0, ldx=None, M=None, Cin=None, N=None,
              ldy=None, tmap=None, ksplit=1, ws=None, tail_ws=None, ragged=None, res_unlrelu=0.0, post_slope=0.0):
    """x: [M, Cin] rows (or any buffer with row stride ldx); wpacked: [N, taps, Cin]; returns [M, N].
    tmap: tile_map(lens, B, S) of the same lens (optional; lets the persistent kernel skip fully padded M-tiles).
    ragged: bookkeeping only (bench.py's valid-row FLOP count) - the rows hold ragged sequences although the launch itself does
    not carry lens (it computes the padded rows too: see Engine.gemm_lens_*).
    tail_ws: tail_workspace(device) scratch (optional; lets the persistent kernel K-split its last partial round of tiles)."""
    if M is None:
        M = x.shape[0]
    if Cin is None:
        Cin = wpacked.shape[-1]
    if N is None:
        N = wpacked.shape[0]
    if ldx is None:
        ldx = x.stride(0)
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=x.dtype)
    if ldy is None:
        ldy = out.stride(0) if out.dim() == 2 else N
    ldr = res.stride(0) if res is not None else 0
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if res_unlrelu or post_slope:          # stored-leaky-ReLU chains (HiFi-GAN): fs2_conv_gemm_lrelu_io
        assert ksplit == 1 and lens is None and in_act == ACT_NONE
        _lib.call("fs2_conv_gemm_lrelu_io", _p(x), ldx, _p(wpacked), _p(bias), _p(res), ldr, _p(out), ldy, _p(tail_ws), M, N, Cin, S, taps,
                  dil, pad, act, slope, int(accumulate), out_scale, res_unlrelu, post_slope, dt(x), _stream())
    elif ksplit > 1:      # few output tiles, long reduction: K-split workgroups + f32 scratch slabs (ksplit x M x N) + finalize launch
        assert ws is not None and ws.numel() >= ksplit * M * N and ws.dtype == torch.float32 and in_act == ACT_NONE and not accumulate
        _lib.call("fs2_conv_gemm_splitk", _p(x), ldx, _p(wpacked), _p(bias), _p(res), ldr, _p(out), ldy, _p(lens), _p(tmap), _p(ws),
                  ksplit, M, N, Cin, S, taps, dil, pad, act, slope, out_scale, dt(x), _stream())
    elif tail_ws is not None:
        _lib.call("fs2_conv_gemm_tail", _p(x), ldx, _p(wpacked), _p(bias), _p(res), ldr, _p(out), ldy, _p(lens), _p(tmap), _p(tail_ws),
                  M, N, Cin, S, taps, dil, pad, act, slope, in_act, in_slope, int(accumulate), out_scale, dt(x), _stream())
    else:
        _lib.call("fs2_conv_gemm", _p(x), ldx, _p(wpacked), _p(bias), _p(res), ldr, _p(out), ldy, _p(lens), _p(tmap), M, N, Cin, S,
                  taps, dil, pad, act, slope, in_act, in_slope, int(accumulate), out_scale, dt(x), _stream())
    if PROFILE is not None:
        e1.record()
        if res_unlrelu or post_slope:
            var = _lib.load().fs2_conv_gemm_lrelu_io_variant(ldx, ldy, ldr, int(accumulate), M, N, Cin, S, taps, dil, act, res_unlrelu,
                                                             post_slope, dt(x))
        else:
            var = 5 if ksplit > 1 and taps > 1 else (6 if ksplit > 1 else
                  _lib.load().fs2_conv_gemm_variant(ldx, ldy, ldr, int(lens is not None), int(tmap is not None), M, N, Cin, S, taps, dil,
                                                    in_act, in_slope, dt(x)))
        PROFILE.setdefault("conv_gemm", []).append((2.0 * M * N * Cin * taps, e0, e1, var, (lens is not None) or bool(ragged), S))
    return out


def resblock_fwd(x, w1, w2, b1, b2, B, S, k, dilations, xs=None, out_scale=1.0, slope=0.1, post_slope=0.0):
    """a whole HiFi-GAN ResBlock1 (hifigan/models.py:96-103) on rows x [B*S][C] in one launch (fs2_resblock_fwd; C in {32, 64}, bf16):
    xs = (xs if given else 0) + out_scale * block(x).  w1 / w2: [3][C][k][C], b1 / b2: [3][C] f32."""
    acc = xs is not None
    if xs is None:
        xs = torch.empty_like(x)
    d0, d1, d2 = dilations
    _lib.call("fs2_resblock_fwd", _p(x), x.stride(0), _p(w1), _p(w2), _p(b1), _p(b2), _p(xs), xs.stride(0), int(acc), out_scale, slope,
              post_slope, B, S, x.shape[1], k, d0, d1, d2, dt(x), _stream())
    return xs


def resstage_fwd(x, blocks, B, S, dilations, out_scale=1.0 / 3, slope=0.1, post_slope=0.0):
    """the three residual blocks of one up-sampling stage in one launch (fs2_resstage_fwd): blocks = [(w1, w2, b1, b2, k)] x 3;
    returns xs = out_scale * sum of the blocks' outputs."""
    xs = torch.empty_like(x)
    args = []
    for w1, w2, b1, b2, k in blocks:
        args += [_p(w1), _p(w2), _p(b1), _p(b2), k]
    d0, d1, d2 = dilations
    _lib.call("fs2_resstage_fwd", _p(x), x.stride(0), *args, _p(xs), xs.stride(0), out_scale, slope, post_slope, B, S, x.shape[1], d0, d1, d2,
              dt(x), _stream())
    return xs


def tail_workspace(device):
    """scratch for conv_gemm(..., tail_ws=): fs2_conv_gemm_tail_ws_bytes() bytes of f32 (one 256x128 slab per CU).  One per stream
    of contraction launches (launches sharing it must not run concurrently)."""
    return torch.empty(_lib.load().fs2_conv_gemm_tail_ws_bytes() // 4, device=device, dtype=torch.float32)


TILE_ROWS = 256      # M-tile height of the persistent contraction kernel (fs2_gemm_p.hip)


def lens_prep(lens, B, S, count_out):
    """int64 lengths -> (lens32 clamped to S, bool padding mask (B, S), tile map); count_out (1-element float view) = sum of the
    clamped lengths.  One launch (fs2_lens_prep)."""
    dev = lens.device
    lens = lens if (lens.dtype == torch.int64 and lens.is_contiguous()) else lens.to(torch.int64).contiguous()
    lens32 = torch.empty(B, device=dev, dtype=torch.int32)
    mask = torch.empty(B, S, device=dev, dtype=torch.bool)
    tmap = torch.empty(1 + (B * S + TILE_ROWS - 1) // TILE_ROWS, device=dev, dtype=torch.int32)
    _lib.call("fs2_lens_prep", _p(lens), B, S, TILE_ROWS, _p(lens32), _p(mask), _p(count_out), _p(tmap), _stream())
    return lens32, mask, tmap


def tile_map(lens, B, S):
    """[n_real, real 256-row M-tiles ..., fully padded M-tiles ...] of the [B*S] row space (device int32); see fs2_tile_map."""
    ntm = (B * S + TILE_ROWS - 1) // TILE_ROWS
    out = torch.empty(1 + ntm, device=lens.device, dtype=torch.int32)
    _lib.call("fs2_tile_map", _p(lens), B, S, TILE_ROWS, _p(out), _stream())
    return out


class WgradWorkspaces:
    """split-K scratch of the weight-gradient launches, one buffer per stream (launches on a stream run in order, so they share
    it), allocated once at the library's own cap (fs2_conv_wgrad_ws_cap: no shape asks for more).  OWNED: an Engine holds one and
    it dies with the engine; `ops` keeps a module-level one only for direct calls (tests, tools).  Should a larger buffer ever be
    needed, the old one is RETIRED, not freed, until `release_retired` is called after the streams have been joined: it was
    allocated from the pool of torch's current stream but is used by kernels queued on another (the engine's side stream);
    handing it back early let the caching allocator give its memory to the next main-stream tensor while a queued weight-gradient
    kernel still wrote slabs there (round-3 finding: non-finite gradients on the first 2-rank step, tools/dbg_ddp.py)."""

    def __init__(self):
        self._ws = {}
        self._retired = []

    def get(self, device, stream_handle, nbytes):
        key = (device.index, stream_handle)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 4 < nbytes:
            if ws is not None:
                self._retired.append(ws)
            ws = torch.empty(max(nbytes, _lib.load().fs2_conv_wgrad_ws_cap()) // 4 + 4, device=device, dtype=torch.float32)
            self._ws[key] = ws
        return ws

    def release_retired(self):
        """call only when every stream that used the retired buffers has been joined / synchronised"""
        self._retired.clear()


_default_wgrad_ws = WgradWorkspaces()


def wgrad_workspace(device, stream_handle, nbytes):
    return _default_wgrad_ws.get(device, stream_handle, nbytes)


def conv_wgrad(dy, x, dw, S, taps=1, dil=1, pad=0, lens=None, dbias=None, use_ws=True, ws_owner=None):
    """dw: tap-major (Cout, k, Cin) fp32 += dy^T * shifted x;  dbias (Cout,) fp32 += column sums of dy (same pass).
    bf16: split-K through a per-stream workspace (slab stores + one finalize launch, no atomics) unless use_ws is False."""
    M, N = dy.shape
    Cin = x.shape[1]
    if use_ws and dy.dtype == torch.bfloat16 and M > 0:
        need = _lib.load().fs2_conv_wgrad_ws_bytes(M, N, Cin, S, taps, dil, int(lens is not None), BF16)
        if need > 0:
            h = _stream()
            ws = (ws_owner or _default_wgrad_ws).get(dy.device, h, need)
            if PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _lib.call("fs2_conv_wgrad_ws", _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), _p(dbias), _p(lens), M, N, Cin, S, taps,
                      dil, pad, BF16, _p(ws), ws.numel() * 4, h)
            if PROFILE is not None:
                e1.record()
                PROFILE.setdefault("conv_wgrad", []).append((2.0 * M * N * Cin * taps, e0, e1, taps, lens is not None, S))
            return
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("fs2_conv_wgrad", _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), _p(dbias), _p(lens), M, N, Cin, S, taps, dil,
              pad, dt(dy), _stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.setdefault("conv_wgrad", []).append((2.0 * M * N * Cin * taps, e0, e1, taps, lens is not None, S))


def colsum(x, out):
    M, N = x.shape
    _lib.call("fs2_colsum", _p(x), x.stride(0), _p(out), M, N, dt(x), _stream())


# ------------------------------------------------------------------ attention
def attn_fwd(qkv, lens, B, S, H, dk=128):
    ctx = torch.empty(B * S, H * dk, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, H, S, device=qkv.device, dtype=torch.float32)
    _lib.call("fs2_attn_fwd", _p(qkv), _p(ctx), _p(lse), _p(lens), B, S, H, dk, float(dk) ** -0.5, dt(qkv), _stream())
    return ctx, lse


def attn_bwd(qkv, ctx, dctx, lse, lens, B, S, H, dk=128):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    _lib.call("fs2_attn_bwd", _p(qkv), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), _p(lens), B, S, H, dk,
              float(dk) ** -0.5, dt(qkv), _stream())
    return dqkv


# ------------------------------------------------------------------ layer norm
def ln_fwd(y, res, gamma, beta, lens, B, S, eps=1e-5, p_pre=0.0, seed_pre=0, p_post=0.0, seed_post=0, seed_dev=None):
    C = y.shape[-1]
    out = torch.empty_like(y)
    mean = torch.empty(B * S, device=y.device, dtype=torch.float32)
    rstd = torch.empty(B * S, device=y.device, dtype=torch.float32)
    _lib.call("fs2_ln_fwd", _p(y), _p(res), _p(gamma), _p(beta), _p(lens), _p(out), _p(mean), _p(rstd), B, S, C, eps,
              p_pre, seed_pre, p_post, seed_post, _p(seed_dev), dt(y), _stream())
    return out, mean, rstd


def gemm_res_ln(x, wpacked, bias, res, gamma, beta, lens, tmap, B, S, eps=1e-5, p_pre=0.0, seed_pre=0, seed_dev=None, streaming_only=False):
    """Linear (N = 256) + dropout + residual + LayerNorm + pad-row zero in ONE launch (fs2_gemm_res_ln_fwd).  Returns
    (z, out, mean, rstd) exactly like conv_gemm followed by ln_fwd (z is what ln_fwd leaves in its y argument), or None when the
    shape is not supported (the caller then runs the two launches)."""
    M, Cin = x.shape
    N = wpacked.shape[0]
    if x.dtype != torch.bfloat16 or wpacked.shape[1] != 1 or not _lib.load().fs2_gemm_res_ln_supported(M, N, Cin, S, BF16):
        return None
    streams = bool(_lib.load().fs2_gemm_res_ln_streams(M, N, Cin, S, BF16))
    if streaming_only and not streams:                   # (the wide-tile form is slower than two launches at the step's shapes)
        return None
    z = torch.empty(M, N, device=x.device, dtype=x.dtype)
    out = torch.empty(M, N, device=x.device, dtype=x.dtype)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("fs2_gemm_res_ln_fwd", _p(x), x.stride(0), _p(wpacked), _p(bias), _p(res), res.stride(0) if res is not None else 0, _p(z), N,
              _p(out), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(lens), _p(tmap), M, N, Cin, S, eps, p_pre, seed_pre, _p(seed_dev), BF16,
              _stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.setdefault("conv_gemm", []).append((2.0 * M * N * Cin, e0, e1, 9 if streams else 7, lens is not None, S))
    return z, out, mean, rstd


def ln_bwd(z, dout, gamma, lens, mean, rstd, dgamma, dbeta, B, S, want_d1=True, want_d2=False, d1_add=None, p_pre=0.0,
           seed_pre=0, p_post=0.0, seed_post=0, relu_bwd=False, seed_dev=None, defer=False, dout2=None):
    """defer=True: the affine-gradient reduction is NOT launched; returns (d1, d2, ws) and the caller runs
    ln_bwd_reduce(ws, C, dgamma, dbeta) later (any stream ordered after this call; ws must stay alive until then).
    dout2: the upstream gradient is dout + dout2 (fs2_ln_bwd_sum: the term that bypassed the sub-layer above, added here instead
    of in the epilogue of the contraction that made dout)."""
    C = z.shape[-1]
    d1 = torch.empty_like(z) if want_d1 else None
    d2 = torch.empty_like(z) if want_d2 else None
    ws = torch.empty(1024 * 2 * C + 4, device=z.device, dtype=torch.float32)
    _lib.call("fs2_ln_bwd_sum", _p(z), _p(dout), _p(dout2), _p(gamma), _p(lens), _p(mean), _p(rstd), _p(d1_add), _p(d1), _p(d2),
              _p(None if defer else dgamma), _p(None if defer else dbeta), _p(ws), B, S, C, p_pre, seed_pre, p_post, seed_post,
              _p(seed_dev), int(relu_bwd), dt(z), _stream())
    if defer:
        return d1, d2, ws
    return d1, d2


def ln_bwd_reduce(ws, C, dgamma, dbeta):
    _lib.call("fs2_ln_bwd_reduce", _p(ws), C, _p(dgamma), _p(dbeta), _stream())


# ------------------------------------------------------------------ batch norm
def bn_train_fwd(x, gamma, beta, running_mean, running_var, act, p, seed, eps=1e-5, momentum=0.1, res=None, seed_dev=None,
                 ws=None, num_batches_tracked=None):
    """ws: a persistent BN workspace (bn_workspace(C): zeroed once, kept consistent by the kernels): statistics, running-stat
    update and the num_batches_tracked increment then take two launches, with no fill / fix-up / counter launches around them."""
    M, C = x.shape
    mean_rstd = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    if ws is not None:
        _lib.call("fs2_bn_train_stats", _p(x), _p(ws), ws.numel(), _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(mean_rstd),
                  M, C, eps, momentum, dt(x), _stream())
    else:
        stats = bn_workspace(C, x.device)
        _lib.call("fs2_bn_stats", _p(x), _p(stats), stats.numel(), M, C, dt(x), _stream())
        _lib.call("fs2_bn_finalize", _p(stats), _p(running_mean), _p(running_var), _p(mean_rstd), M, C, eps, momentum,
                  _stream())
    out = torch.empty_like(x)
    _lib.call("fs2_bn_apply", _p(x), _p(mean_rstd), _p(gamma), _p(beta), _p(res), _p(out), M, C, act, p, seed, _p(seed_dev),
              dt(x), _stream())
    return out, mean_rstd


def bn_workspace(C, device):
    """workspace of the reducing BatchNorm launches (reduced sums + per-workgroup partial sums, summed in index order)"""
    return torch.zeros(_lib.load().fs2_bn_ws_floats(C), device=device, dtype=torch.float32)


def bn_bwd_acc(x, dout, mean_rstd, gamma, beta, act, p, seed, ws, dgamma_acc, dbeta_acc, seed_dev=None):
    """BatchNorm backward with the affine gradients accumulated straight into dgamma_acc / dbeta_acc; ws: bn_workspace(C)."""
    M, C = x.shape
    dx = torch.empty_like(x)
    _lib.call("fs2_bn_bwd_acc", _p(x), _p(dout), _p(mean_rstd), _p(gamma), _p(beta), _p(ws), ws.numel(), _p(dx), _p(dgamma_acc),
              _p(dbeta_acc), M, C, act, p, seed, _p(seed_dev), dt(x), _stream())
    return dx


def bn_bwd(x, dout, mean_rstd, gamma, beta, act, p, seed, seed_dev=None):
    M, C = x.shape
    sums = bn_workspace(C, x.device)
    dx = torch.empty_like(x)
    _lib.call("fs2_bn_bwd", _p(x), _p(dout), _p(mean_rstd), _p(gamma), _p(beta), _p(sums), sums.numel(), _p(dx), M, C, act, p, seed,
              _p(seed_dev), dt(x), _stream())
    return dx, sums[C:2 * C], sums[:C]  # dx, dgamma, dbeta


# ------------------------------------------------------------------ gathers
def embed_pe_fwd(tokens, emb, pe, dtype):
    B, L = tokens.shape
    V, C = emb.shape
    out = torch.empty(B * L, C, device=emb.device, dtype=dtype)
    _lib.call("fs2_embed_pe_fwd", _p(tokens), _p(emb), _p(pe), _p(out), B, L, C, V, dt(dtype), _stream())
    return out


def embed_bwd(tokens, dy, demb, pad_idx=0):
    V, C = demb.shape
    _lib.call("fs2_embed_bwd", _p(tokens), _p(dy), _p(demb), tokens.numel(), C, V, pad_idx, dt(dy), _stream())


def add_rowvec(x, table, idx, B, S):
    V, C = table.shape
    _lib.call("fs2_add_rowvec", _p(x), _p(table), _p(idx), B, S, C, V, dt(x), _stream())


def rowvec_bwd(dy, dtable, idx, B, S):
    V, C = dtable.shape
    _lib.call("fs2_rowvec_bwd", _p(dy), _p(dtable), _p(idx), B, S, C, V, dt(dy), _stream())


def bucket_embed_add_fwd(x, vals, scale, bins, emb):
    rows, C = x.shape
    out = torch.empty_like(x)
    idx = torch.empty(rows, device=x.device, dtype=torch.int32)
    _lib.call("fs2_bucket_embed_add_fwd", _p(x), _p(vals), float(scale), _p(bins), bins.numel(), _p(emb), _p(out),
              _p(idx), rows, C, dt(x), _stream())
    return out, idx


def bucket_embed_bwd(idx, dy, demb):
    rows, C = dy.shape
    _lib.call("fs2_bucket_embed_bwd", _p(idx), _p(dy), _p(demb), rows, demb.shape[0], C, dt(dy), _stream())


def lr_index(durations, T):
    B, L = durations.shape
    is_float = durations.dtype == torch.float32
    assert is_float or durations.dtype == torch.int64
    dev = durations.device
    cum = torch.empty(B, L + 1, device=dev, dtype=torch.int32)
    idx = torch.empty(B, T, device=dev, dtype=torch.int32)
    mel_len = torch.empty(B, device=dev, dtype=torch.int64)
    _lib.call("fs2_lr_index", _p(_contig(durations)), int(is_float), B, L, T, _p(cum), _p(idx), _p(mel_len), _stream())
    return cum, idx, mel_len


def lr_gather_fwd(x, idx, pe, B, L, T):
    C = x.shape[-1]
    out = torch.empty(B * T, C, device=x.device, dtype=x.dtype)
    _lib.call("fs2_lr_gather_fwd", _p(x), _p(idx), _p(pe), _p(out), B, L, T, C, dt(x), _stream())
    return out


def lr_gather_bwd(dy, cum, B, L, T, dx=None, accumulate=False):
    C = dy.shape[-1]
    if dx is None:
        dx = torch.empty(B * L, C, device=dy.device, dtype=dy.dtype)
    _lib.call("fs2_lr_gather_bwd", _p(dy), _p(cum), _p(dx), B, L, T, C, int(accumulate), dt(dy), _stream())
    return dx


def duration_round(logd, d_control):
    out = torch.empty_like(logd)
    _lib.call("fs2_duration_round", _p(logd), float(d_control), _p(out), logd.numel(), _stream())
    return out


def rowdot_fwd(x, w, bias, lens, B, S):
    out = torch.empty(B, S, device=x.device, dtype=torch.float32)
    _lib.call("fs2_rowdot_fwd", _p(x), _p(w), _p(bias), _p(lens), _p(out), B, S, x.shape[-1], dt(x), _stream())
    return out


def rowdot_bwd(x, w, g, lens, dw, db, B, S):
    dx = torch.empty_like(x)
    _lib.call("fs2_rowdot_bwd", _p(x), _p(w), _p(g), _p(lens), _p(dx), _p(dw), _p(db), B, S, x.shape[-1], dt(x),
              _stream())
    return dx


def mask_rows(x, lens, B, S):
    _lib.call("fs2_mask_rows", _p(x), _p(lens), B, S, x.shape[-1], dt(x), _stream())
    return x


def cast(x, dtype, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    _lib.call("fs2_cast", _p(x), dt(x), _p(out), dt(dtype), x.numel(), _stream())
    return out


# ------------------------------------------------------------------ optimiser
def sumsq(x, out, ws=None):
    """out[0] += ||x||^2, bit-reproducible (fixed summation order).  ws: >= 1024 floats of workspace."""
    if ws is None:
        ws = torch.empty(1024, device=x.device, dtype=torch.float32)
    _lib.call("fs2_sumsq", _p(x), x.numel(), _p(out), _p(ws), _stream())


def adam_step(p, g, m, v, gnorm_sq, max_norm, hyper, b1, b2, eps, wd, p_lowp=None, zero_grad=False):
    """clip + Adam over the flat buffers; optionally writes the bf16 shadow copy of p and clears g in the same pass."""
    _lib.call("fs2_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(gnorm_sq), float(max_norm), _p(hyper), b1, b2,
              eps, wd, _p(p_lowp), BF16 if p_lowp is not None else 0, int(zero_grad), _stream())


def pack_dgrad_multi(flat, wd_all, table, total_tiles):
    _lib.call("fs2_pack_dgrad_multi", _p(flat), _p(wd_all), _p(table), table.shape[0], total_tiles, dt(wd_all), _stream())


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    _lib.call("fs2_add", _p(a), _p(b), _p(out), a.numel(), dt(a), _stream())
    return out


def add_pe(x, pe, B, S):
    _lib.call("fs2_add_pe", _p(x), _p(pe), B, S, x.shape[-1], dt(x), _stream())
    return x


def bump_counter(ctr, inc=1):
    _lib.call("fs2_bump_counter", _p(ctr), inc, _stream())


# ------------------------------------------------------------------ loss
def loss_fwd(mel, post, mel_t, mel_lens, src_lens, p_pred, p_t, e_pred, e_t, logd, dur, cnt, p_frame, e_frame):
    """-> losses[6] (device f32): total, mel, postnet, pitch, energy, duration."""
    B, T, n_mel = mel.shape
    L = logd.shape[1]
    sums = torch.empty(5, device=mel.device, dtype=torch.float32)
    losses = torch.empty(6, device=mel.device, dtype=torch.float32)
    _lib.call("fs2_loss_fwd", _p(mel), _p(post), _p(mel_t), mel_t.stride(0), _p(mel_lens), _p(src_lens), _p(p_pred), _p(p_t),
              p_t.stride(0), _p(e_pred), _p(e_t), e_t.stride(0), _p(logd), _p(dur), dur.stride(0), _p(cnt), B, T, L, n_mel,
              int(p_frame), int(e_frame), _p(sums), _p(losses), _stream())
    return losses


def loss_bwd(mel, post, mel_t, mel_lens, src_lens, p_pred, p_t, e_pred, e_t, logd, dur, cnt, g, p_frame, e_frame):
    B, T, n_mel = mel.shape
    L = logd.shape[1]
    dmel, dpost = torch.empty_like(mel), torch.empty_like(post)
    dp, de, dlogd = torch.empty_like(p_pred), torch.empty_like(e_pred), torch.empty_like(logd)
    _lib.call("fs2_loss_bwd", _p(mel), _p(post), _p(mel_t), mel_t.stride(0), _p(mel_lens), _p(src_lens), _p(p_pred), _p(p_t),
              p_t.stride(0), _p(e_pred), _p(e_t), e_t.stride(0), _p(logd), _p(dur), dur.stride(0), _p(cnt), _p(g), B, T, L, n_mel,
              int(p_frame), int(e_frame), _p(dmel), _p(dpost), _p(dp), _p(de), _p(dlogd), _stream())
    return dmel, dpost, dp, de, dlogd
