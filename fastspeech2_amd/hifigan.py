"""HiFi-GAN generator (reference hifigan/models.py:112-174, hifigan/__init__.py) over the HIP kernel library.

Same surface as the reference: `Generator(h)`, weight-normalised `state_dict` keys (`*.weight_g / *.weight_v / *.bias`,
SURVEY Appendix C) so `generator_*.pth.tar["generator"]` loads unchanged, `remove_weight_norm()`, `forward(x)` with
x (B, 80, T) float32 -> (B, 1, T*prod(upsample_rates)) float32.  Inference only (the reference never trains it).

Underneath every layer is a launch of the implicit-GEMM kernel on time-major rows [B*T][C]:
  * Conv1d (conv_pre, ResBlock convs, dilation 1/3/5): taps = k, leaky-ReLU applied on operand load (prologue) or on
    the producer's accumulators (epilogue), residual add / 3-branch mean (`xs / 3`, models.py:160) fused as
    accumulate + out_scale;
  * ConvTranspose1d(k = 2u, stride u, pad u/2): polyphase decomposition.  Output sample u*q + r only touches inputs
    q-1, q, q+1, so the layer is a 3-tap conv with N = u*Cout output columns, and the row-major output
    [B*T][u*Cout] *is* the up-sampled signal in time-major order [B*T*u][Cout] — no zero-stuffing, no scatter;
  * conv_post + tanh + int16 PCM: one fused HBM pass (`fs2_conv_post_pcm`).
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import ACT_LRELU

LRELU_SLOPE = 0.1          # hifigan/models.py:7


class AttrDict(dict):
    """hifigan/__init__.py / models.py AttrDict: dict with attribute access."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def get_padding(kernel_size, dilation=1):
    """hifigan/models.py:16."""
    return int((kernel_size * dilation - dilation) / 2)


class _WNConv(nn.Module):
    """Parameter holder with torch.nn.utils.weight_norm's key names (weight_g, weight_v) / plain `weight` after removal."""

    def __init__(self, wshape, fan_in, std=None, norm_dim0=None):
        super().__init__()
        v = torch.empty(*wshape)
        if std is None:
            nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        else:
            v.normal_(0.0, std)                                        # init_weights, hifigan/models.py:10-13
        self.weight_v = nn.Parameter(v)
        self.weight_g = nn.Parameter(v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1))
        bound = 1.0 / math.sqrt(fan_in)
        self.bias = nn.Parameter(torch.empty(norm_dim0 or wshape[0]).uniform_(-bound, bound))

    def effective_weight(self):
        if hasattr(self, "weight"):
            return self.weight
        v, g = self.weight_v, self.weight_g
        norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
        return v * (g / norm)

    def remove_weight_norm(self):
        if hasattr(self, "weight"):
            raise ValueError("weight_norm of this layer was already removed")   # same failure as torch's helper
        w = self.effective_weight().detach()
        del self.weight_g
        del self.weight_v
        self.weight = nn.Parameter(w)


class _ResBlock(nn.Module):
    def __init__(self, channels, kernel_size, dilations):
        super().__init__()
        self.kernel_size, self.dilations = kernel_size, tuple(dilations)
        self.convs1 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels * kernel_size, std=0.01)
                                     for _ in dilations])
        self.convs2 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels * kernel_size, std=0.01)
                                     for _ in dilations])


class Generator(nn.Module):
    """reference hifigan/models.py:112-174 (resblock type "1")."""

    def __init__(self, h, compute_dtype="fp32"):
        super().__init__()
        self.h = h
        assert str(h["resblock"]) == "1", "only ResBlock1 (the reference's shipped config) is built"
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        c0 = h["upsample_initial_channel"]
        self.conv_pre = _WNConv((c0, 80, 7), 80 * 7)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
            # ConvTranspose1d weight is (Cin, Cout, k); weight_norm's dim-0 norm is therefore per INPUT channel
            self.ups.append(_WNConv((cin, cout, k), cin * k, std=0.01, norm_dim0=cout))
            for rk, rd in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(_ResBlock(cout, rk, rd))
        self.conv_post = _WNConv((1, c0 // (2 ** self.num_upsamples), 7), c0 // (2 ** self.num_upsamples) * 7)
        self.set_compute_dtype(compute_dtype)
        self._packed = None
        # True: a narrow stage's three residual blocks in one launch; "block": one launch per block; False: one per convolution (A/B, tests)
        self.fuse_resblocks = True
        self.stage_probe = None          # a list: _run appends the output rows of conv_pre and of every up-sampling stage (tests)
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    def set_compute_dtype(self, compute_dtype):
        if isinstance(compute_dtype, str):
            compute_dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                             "bfloat16": torch.bfloat16}[compute_dtype]
        self.compute_dtype = compute_dtype
        self._packed = None

    def _invalidate(self):
        self._packed = None

    def remove_weight_norm(self):
        """hifigan/models.py:167-174."""
        for l in self.ups:
            l.remove_weight_norm()
        for rb in self.resblocks:
            for l in list(rb.convs1) + list(rb.convs2):
                l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
        self._invalidate()

    # ------------------------------------------------------------------ weight packing (once per load / device)
    @staticmethod
    def _pack_conv(layer, dev, cdt):
        w = layer.effective_weight().detach().to(dev, torch.float32)            # (Cout, Cin, k)
        return (w.permute(0, 2, 1).contiguous().to(cdt), layer.bias.detach().to(dev, torch.float32).contiguous())

    @staticmethod
    def _pack_convt(layer, u, k, dev, cdt):
        """Polyphase pack of ConvTranspose1d(Cin, Cout, k, stride=u, padding=(k-u)//2):
        y[u*q + r, co] = sum_d sum_ci x[q + d, ci] * w[ci, co, r + p - u*d]  for the d with 0 <= r + p - u*d < k."""
        w = layer.effective_weight().detach().to(dev, torch.float32)            # (Cin, Cout, k)
        cin, cout, _ = w.shape
        p = (k - u) // 2
        ds = sorted({d for r in range(u) for d in range(-k, k + 1) if 0 <= r + p - u * d < k})
        dmin, dmax = ds[0], ds[-1]
        taps = dmax - dmin + 1
        wp = torch.zeros(u, cout, taps, cin, device=dev, dtype=torch.float32)
        for r in range(u):
            for d in range(dmin, dmax + 1):
                j = r + p - u * d
                if 0 <= j < k:
                    wp[r, :, d - dmin, :] = w[:, :, j].t()
        bias = layer.bias.detach().to(dev, torch.float32).repeat(u).contiguous()
        return wp.view(u * cout, taps, cin).contiguous().to(cdt), bias, taps, -dmin

    def _weights(self, dev):
        key = (dev, self.compute_dtype)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        cdt = self.compute_dtype
        h = self.h
        W = {"pre": self._pack_conv(self.conv_pre, dev, cdt)}
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            W[f"up{i}"] = self._pack_convt(self.ups[i], u, k, dev, cdt)
        for j, rb in enumerate(self.resblocks):
            for m in range(len(rb.dilations)):
                W[f"rb{j}.1.{m}"] = self._pack_conv(rb.convs1[m], dev, cdt)
                W[f"rb{j}.2.{m}"] = self._pack_conv(rb.convs2[m], dev, cdt)
        # the narrow stages' residual blocks run as ONE launch each (fs2_resblock_fwd): convs1 / convs2 stacked [3][C][k][C]
        lib = _lib.load()
        for j, rb in enumerate(self.resblocks):
            C = rb.convs1[0].bias.shape[0]
            nd = len(rb.dilations)
            if nd == 3 and lib.fs2_resblock_supported(C, rb.kernel_size, *rb.dilations, ops.dt(cdt)):
                W[f"rb{j}.fused"] = (torch.stack([W[f"rb{j}.1.{m}"][0] for m in range(nd)]).contiguous(),
                                     torch.stack([W[f"rb{j}.2.{m}"][0] for m in range(nd)]).contiguous(),
                                     torch.stack([W[f"rb{j}.1.{m}"][1] for m in range(nd)]).contiguous(),
                                     torch.stack([W[f"rb{j}.2.{m}"][1] for m in range(nd)]).contiguous())
        wpost = self.conv_post.effective_weight().detach().to(dev, torch.float32)   # (1, C, 7)
        W["post"] = (wpost[0].t().contiguous(), self.conv_post.bias.detach().to(dev, torch.float32).contiguous())
        self._packed = (key, W)
        return W

    def prepare(self, device):
        """pack the kernels' weight images for `device` now, on the current stream (they are otherwise packed by the first forward, on
        whatever stream that runs on: a caller that spreads forwards over several streams - utils.SynthPipeline - orders them behind this)"""
        self._weights(torch.device(device))

    # ------------------------------------------------------------------ forward
    def _run(self, x_rows, B, T, want_wav, want_pcm, max_wav_value):
        """x_rows: [B*T][80] in the compute dtype.  Returns (wav f32 (B, T*hop) | None, pcm int16 (B, T*hop) | None).

        Storage convention (round 5): every convolution of the generator reads leaky_relu(x, 0.1) (models.py:98-100, 152), so the chains of
        single launches STORE leaky_relu(value) and read it without a prologue (the in-kernel prologue cost 27-35 % of every launch that
        had one); the raw value is needed in two places only and undone there on the fly (r > 0 ? r : r / slope): the residual add of a
        block's conv2, and what the tests probe.  A stage whose residual blocks run fused (C <= 64) takes and keeps raw rows - the
        running sum never leaves registers there - and hands leaky_relu(xs) to the next up-sampling convolution.  `lre` below says
        which form the current `x` is in.  The last stage's output stays raw: conv_post applies ITS leaky_relu (slope 0.01, models.py:161)."""
        h = self.h
        W = self._weights(x_rows.device)
        S = T
        SL, INV = LRELU_SLOPE, 1.0 / LRELU_SLOPE
        wp, bp = W["pre"]
        x = ops.conv_gemm(x_rows, wp, bp, S, taps=7, pad=3, post_slope=SL)                # stored leaky-ReLU'd: up0 reads it as it is
        lre = True

        def raw(t):
            """fp32 copy of a stored activation in its RAW form (tests' stage probe only)"""
            t = t.float()
            return torch.where(t > 0, t, t * INV)

        if self.stage_probe is not None:
            self.stage_probe.append((raw(x), S))
        nk = self.num_kernels
        nup = len(h["upsample_rates"])
        for i, u in enumerate(h["upsample_rates"]):
            wu, bu, taps, pad = W[f"up{i}"]
            cout = wu.shape[0] // u
            fz = [W.get(f"rb{i * nk + j}.fused") for j in range(nk)] if self.fuse_resblocks else [None] * nk
            dils = {self.resblocks[i * nk + j].dilations for j in range(nk)}
            fused = all(f is not None for f in fz) and len(dils) == 1 and nk == 3
            out_sl = SL if i + 1 < nup else 0.0             # what the NEXT consumer wants: lrelu'd for an up-sampling conv, raw for conv_post
            assert lre
            # the polyphase transposed convolution: its input is stored lrelu'd (no prologue); its output feeds a fused stage raw,
            # a chain of single launches lrelu'd
            y = ops.conv_gemm(x, wu, bu, S, taps=taps, pad=pad, post_slope=0.0 if fused else SL)
            S = S * u
            x = y.view(B * S, cout)
            ks = [self.resblocks[i * nk + j].kernel_size for j in range(nk)]
            if fused and self.fuse_resblocks != "block":
                # the whole stage's residual blocks in ONE launch: x read once, xs written once
                xs = torch.empty_like(x)
                args = []
                for j in range(nk):
                    args += [fz[j][0].data_ptr(), fz[j][1].data_ptr(), fz[j][2].data_ptr(), fz[j][3].data_ptr(), ks[j]]
                if ops.PROFILE is not None:          # bench.py's roofline replay: HIP events around the launch, algorithmic FLOPs
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.call("fs2_resstage_fwd", x.data_ptr(), x.stride(0), *args, xs.data_ptr(), xs.stride(0), 1.0 / nk, SL, out_sl,
                          B, S, x.shape[1], *next(iter(dils)), ops.dt(x), ops._stream())
                if ops.PROFILE is not None:
                    e1.record()
                    C_ = x.shape[1]
                    ops.PROFILE.setdefault("conv_gemm", []).append((2.0 * B * S * C_ * C_ * sum(ks) * 6, e0, e1, 10, False, S))
            elif fused:
                # one launch per block (A/B form): the last one also applies the output leaky-ReLU
                xs = None
                for j in range(nk):
                    rb = self.resblocks[i * nk + j]
                    xs = ops.resblock_fwd(x, fz[j][0], fz[j][1], fz[j][2], fz[j][3], B, S, ks[j], rb.dilations, xs=xs, out_scale=1.0 / nk,
                                          slope=SL, post_slope=out_sl if j == nk - 1 else 0.0)
            else:
                xs = None
                for j in range(nk):
                    rb = self.resblocks[i * nk + j]
                    rk = rb.kernel_size
                    cur = x                                 # lrelu'd rows
                    nd = len(rb.dilations)
                    for m, d in enumerate(rb.dilations):
                        w1, b1 = W[f"rb{i * nk + j}.1.{m}"]
                        w2, b2 = W[f"rb{i * nk + j}.2.{m}"]
                        # t = lrelu(conv1(lrelu(cur))): the operand is stored lrelu'd, the second leaky-ReLU runs on conv1's accumulators
                        t = ops.conv_gemm(cur, w1, b1, S, taps=rk, dil=d, pad=get_padding(rk, d), act=ACT_LRELU, slope=SL)
                        if m < nd - 1:                      # cur <- lrelu(conv2(t) + raw cur)
                            cur = ops.conv_gemm(t, w2, b2, S, taps=rk, pad=get_padding(rk, 1), res=cur, res_unlrelu=INV, post_slope=SL)
                        else:   # last conv of the branch: xs (+)= (conv + raw cur) / num_kernels (models.py:155-160); the stage's last
                                # launch stores what the next consumer reads
                            xs = ops.conv_gemm(t, w2, b2, S, taps=rk, pad=get_padding(rk, 1), res=cur, res_unlrelu=INV, out=xs,
                                               accumulate=xs is not None, out_scale=1.0 / nk,
                                               post_slope=out_sl if j == nk - 1 else 0.0)
            x = xs
            lre = out_sl > 0
            if self.stage_probe is not None:
                self.stage_probe.append((raw(x) if lre else x, S))
        assert not lre
        wpost, bpost = W["post"]
        M = B * S
        wav = torch.empty(B, S, device=x.device, dtype=torch.float32) if want_wav else None
        pcm = torch.empty(B, S, device=x.device, dtype=torch.int16) if want_pcm else None
        # F.leaky_relu(x) with the DEFAULT slope 0.01 (models.py:161), conv_post, tanh, (x 32768 -> int16)
        _lib.call("fs2_conv_post_pcm", x.data_ptr(), x.stride(0), wpost.data_ptr(), bpost.data_ptr(), 0.01,
                  wav.data_ptr() if want_wav else None, pcm.data_ptr() if want_pcm else None, float(max_wav_value), M, S,
                  x.shape[1], wpost.shape[0], 3, ops.dt(x), ops._stream())
        return wav, pcm

    def _rows(self, x):
        """(B, 80, T) float32 (the reference's layout) -> rows [B*T][80] in the compute dtype."""
        if not x.is_cuda:
            raise RuntimeError("fastspeech2_amd.hifigan.Generator runs on an AMD GPU only (no CPU fallback)")
        B, C, T = x.shape
        xt = x.transpose(1, 2)
        if xt.is_contiguous() and x.dtype == torch.float32:          # the usual case: mel_predictions.transpose(1, 2)
            rows = xt.reshape(B * T, C)
            return (ops.cast(rows, self.compute_dtype) if self.compute_dtype != torch.float32 else rows), B, T
        x = x.contiguous().float()
        rows = torch.empty(B * T, C, device=x.device, dtype=self.compute_dtype)
        _lib.call("fs2_chan_to_rows", x.data_ptr(), rows.data_ptr(), B, C, T, ops.dt(rows), ops._stream())
        return rows, B, T

    def forward(self, x):
        rows, B, T = self._rows(x)
        wav, _ = self._run(rows, B, T, True, False, 32768.0)
        return wav.unsqueeze(1)

    def infer_pcm(self, x, max_wav_value=32768.0):
        """forward + utils/model.py:82-85's `(wav * max_wav_value).astype("int16")` fused on the device.
        Returns int16 (B, T*hop)."""
        rows, B, T = self._rows(x)
        _, pcm = self._run(rows, B, T, False, True, max_wav_value)
        return pcm
