"""CPU oracle for the FastSpeech 2 hot path — TEST INFRASTRUCTURE ONLY.

A functional restatement (plain torch CPU ops on a state_dict, fp32 by default, fp64 on request) of the
reference's algorithm, each function citing the reference file:line it follows.  It is pinned against the live
reference by tests/golden/make_golden.py (run in the build container, where /root/reference exists) and the
committed fixtures under tests/golden/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path (fastspeech2_amd/) never does.

Parity status: acoustic model / loss / optimiser / HiFi-GAN are PINNED by golden vectors generated from the
reference itself.  The STFT mel filterbank follows librosa==0.7.2 `filters.mel` (Slaney) which is absent from
/root/reference -> that boundary is "parity unpinned" (see DESIGN.md).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ storage emulation
# The product's bf16 path STORES every activation between two kernels in bfloat16 (and, in backward, the gradient of that
# activation).  `with storage(round_st_bf16):` makes the functions below apply the given function at exactly those points - the
# bf16 EMULATION the whole-step bf16 bars are derived from (tests/golden/make_bf16_bars.py).  Outside the context nothing is
# applied and every function is the plain fp32 / fp64 restatement pinned by the goldens.
_STORE = None


class storage:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _STORE
        self.prev, _STORE = _STORE, self.fn
        return self

    def __exit__(self, *exc):
        global _STORE
        _STORE = self.prev
        return False


def _st(x):
    return x if _STORE is None else _STORE(x)


class _RoundST(torch.autograd.Function):
    """bf16 rounding of a stored activation with a straight-through backward that rounds the GRADIENT the same way (the product
    keeps the gradient of a bf16 activation in bf16 too)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def round_st_bf16(x):
    return _RoundST.apply(x)


class realisation:
    """A rounding REALISATION of the bf16 emulation: every value is multiplied by (1 + sigma * N(0, 1)) right before it is rounded
    to bf16, with sigma = 2^-19 ~ the relative error of an fp32 accumulation over K = 256 ... 2304 products (sqrt(K) * 2^-24).
    The product's sums differ from the exact fp64 sums by about that much and in a build-dependent pattern (tile shapes, split-K
    order), which flips the ~1e-3 of the bf16 roundings that sit near a tie; downstream the step is chaotic at its own rounding
    level (L1 sign flips), so each realisation is an independent draw of the heavy-tailed per-tensor distance.  The error
    MAGNITUDE of a rounding is unchanged (2^-19 << 2^-9).  `store` is the function for `storage(...)`, `plain` the autograd-free
    twin the attention emulation takes.  tests/golden/make_bf16_bars.py."""

    def __init__(self, seed, sigma=2.0 ** -19):
        self.gen = torch.Generator().manual_seed(seed)
        self.sigma = sigma
        outer = self

        class _R(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return outer.plain(x)

            @staticmethod
            def backward(ctx, g):
                return outer.plain(g)
        self._fn = _R

    def plain(self, x):
        noise = torch.randn(x.shape, generator=self.gen, dtype=torch.float32).to(x.dtype)
        return (x * (1.0 + self.sigma * noise)).to(torch.bfloat16).to(x.dtype)

    def store(self, x):
        return self._fn.apply(x)


class _AttnCoreEmu(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk), key mask) v with the roundings of the product's flash-style kernels (fs2_attn.hip) - used only
    under `storage(...)`.  Forward: scores and softmax in full precision, P ROUNDED as the operand of the second product.
    Backward (attn_bwd_dq / attn_bwd_dkv): dV = P_r^T dO;  dP = dO V^T;  delta = rowsum(dO * O) with the STORED (rounded) O - not
    rowsum(P * dP);  dS = P (dP - delta) ROUNDED as the operand of  dQ = dS_r K / sqrt(dk),  dK = dS_r^T Q / sqrt(dk).
    q, k, v arrive already rounded (stored tensors); `rnd` is the plain rounding function (no autograd)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, rnd):
        dk = q.shape[-1]
        s = torch.matmul(q, k.transpose(-1, -2)) / (dk ** 0.5)
        s = s.masked_fill(mask, float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = rnd(torch.matmul(rnd(p), v))
        ctx.save_for_backward(q, k, v, p, o)
        ctx.rnd = rnd
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        rnd = ctx.rnd
        dk = q.shape[-1]
        do = rnd(do)
        dv = torch.matmul(rnd(p).transpose(-1, -2), do)
        dp = torch.matmul(do, v.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True)
        ds = rnd(p * (dp - delta))
        dq = torch.matmul(ds, k) / (dk ** 0.5)
        dkk = torch.matmul(ds.transpose(-1, -2), q) / (dk ** 0.5)
        return rnd(dq), rnd(dkk), rnd(dv), None, None


def _plain_bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


# ------------------------------------------------------------------------------------------------ helpers
def sinusoid_table(n_position, d_hid):
    """transformer/Models.py:10-30 — float64 numpy angles, sin on even / cos on odd dims, cast to float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)
    table = np.empty_like(angle)
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.from_numpy(table).float()


def mask_from_lengths(lengths, max_len=None):
    """utils/tools.py:91-99 — True = padding."""
    if max_len is None:
        max_len = int(lengths.max().item())
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0)
    return ids >= lengths.unsqueeze(1)


def _conv1d_rows(x, w, b, padding, dilation=1):
    """nn.Conv1d applied to a (B, S, C) tensor (the reference transposes around every conv:
    model/modules.py:291-296, transformer/SubLayers.py:87-89)."""
    return F.conv1d(x.transpose(1, 2), w, b, padding=padding, dilation=dilation).transpose(1, 2)


# ------------------------------------------------------------------------------------------------ FFT block
def multi_head_attention(sd, pre, x, key_pad_mask, n_head, dropout_p, training):
    """transformer/SubLayers.py:29-57 + transformer/Modules.py:14-25 (post-LN, key-padding mask, no attn dropout)."""
    B, S, D = x.shape
    dk = D // n_head
    q = _st(F.linear(x, sd[pre + "w_qs.weight"], sd[pre + "w_qs.bias"])).view(B, S, n_head, dk).permute(0, 2, 1, 3)
    k = _st(F.linear(x, sd[pre + "w_ks.weight"], sd[pre + "w_ks.bias"])).view(B, S, n_head, dk).permute(0, 2, 1, 3)
    v = _st(F.linear(x, sd[pre + "w_vs.weight"], sd[pre + "w_vs.bias"])).view(B, S, n_head, dk).permute(0, 2, 1, 3)
    if _STORE is None:
        attn = torch.matmul(q, k.transpose(-1, -2)) / (dk ** 0.5)
        attn = attn.masked_fill(key_pad_mask.view(B, 1, 1, S), float("-inf"))
        attn = torch.softmax(attn, dim=-1)
        out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, S, D)
    else:                                                  # the product's kernel-internal operand roundings made explicit
        owner = getattr(_STORE, "__self__", None)          # storage(realisation(seed).store): its autograd-free twin
        rnd = owner.plain if isinstance(owner, realisation) else _plain_bf16
        out = _AttnCoreEmu.apply(q, k, v, key_pad_mask.view(B, 1, 1, S), rnd).permute(0, 2, 1, 3).reshape(B, S, D)
    out = F.dropout(_st(F.linear(out, sd[pre + "fc.weight"], sd[pre + "fc.bias"])), dropout_p, training)
    return _st(F.layer_norm(_st(out + x), (D,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5))


def positionwise_ffn(sd, pre, x, kernel_size, dropout_p, training):
    """transformer/SubLayers.py:85-93."""
    h = _st(F.relu(_conv1d_rows(x, sd[pre + "w_1.weight"], sd[pre + "w_1.bias"], (kernel_size[0] - 1) // 2)))
    h = _st(_conv1d_rows(h, sd[pre + "w_2.weight"], sd[pre + "w_2.bias"], (kernel_size[1] - 1) // 2))
    h = F.dropout(h, dropout_p, training)
    D = x.shape[-1]
    return _st(F.layer_norm(_st(h + x), (D,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5))


def fft_block(sd, pre, x, pad_mask, n_head, kernel_size, dropout_p, training):
    """transformer/Layers.py:21-30 — masked_fill(mask, 0) after each sub-layer."""
    x = multi_head_attention(sd, pre + "slf_attn.", x, pad_mask, n_head, dropout_p, training)
    x = x.masked_fill(pad_mask.unsqueeze(-1), 0)
    x = positionwise_ffn(sd, pre + "pos_ffn.", x, kernel_size, dropout_p, training)
    return x.masked_fill(pad_mask.unsqueeze(-1), 0)


def encoder(sd, cfg, texts, pad_mask, training, dropout):
    """transformer/Models.py:73-100."""
    tc = cfg["transformer"]
    L = texts.shape[1]
    emb = sd["encoder.src_word_emb.weight"]
    if (not training) and L > cfg["max_seq_len"]:
        pe = sinusoid_table(L, tc["encoder_hidden"]).to(emb.dtype)
    else:
        pe = sd["encoder.position_enc"][0, :L]
    x = _st(F.embedding(texts, emb, padding_idx=0) + pe.unsqueeze(0))
    p = tc["encoder_dropout"] if dropout else 0.0
    for i in range(tc["encoder_layer"]):
        x = fft_block(sd, f"encoder.layer_stack.{i}.", x, pad_mask, tc["encoder_head"], tc["conv_kernel_size"], p, training)
    return x


def decoder(sd, cfg, x, pad_mask, training, dropout):
    """transformer/Models.py:139-171 — truncation to max_seq_len in train (or when it fits)."""
    tc = cfg["transformer"]
    T = x.shape[1]
    if (not training) and T > cfg["max_seq_len"]:
        x = _st(x + sinusoid_table(T, tc["decoder_hidden"]).to(x.dtype).unsqueeze(0))
    else:
        T = min(T, cfg["max_seq_len"])
        x = _st(x[:, :T] + sd["decoder.position_enc"][0, :T].unsqueeze(0))
        pad_mask = pad_mask[:, :T]
    p = tc["decoder_dropout"] if dropout else 0.0
    for i in range(tc["decoder_layer"]):
        x = fft_block(sd, f"decoder.layer_stack.{i}.", x, pad_mask, tc["decoder_head"], tc["conv_kernel_size"], p, training)
    return x, pad_mask


# ------------------------------------------------------------------------------------------------ variance adaptor
def variance_predictor(sd, pre, cfg, x, pad_mask, training, dropout):
    """model/modules.py:242-250, conv stack :209-240 (second conv has padding=1 regardless of kernel size)."""
    vc = cfg["variance_predictor"]
    k = vc["kernel_size"]
    p = vc["dropout"] if dropout else 0.0
    C = vc["filter_size"]
    h = _conv1d_rows(x, sd[pre + "conv_layer.conv1d_1.conv.weight"], sd[pre + "conv_layer.conv1d_1.conv.bias"], (k - 1) // 2)
    h = _st(F.layer_norm(_st(F.relu(h)), (C,), sd[pre + "conv_layer.layer_norm_1.weight"], sd[pre + "conv_layer.layer_norm_1.bias"], 1e-5))
    h = F.dropout(h, p, training)
    h = _conv1d_rows(h, sd[pre + "conv_layer.conv1d_2.conv.weight"], sd[pre + "conv_layer.conv1d_2.conv.bias"], 1)
    h = _st(F.layer_norm(_st(F.relu(h)), (C,), sd[pre + "conv_layer.layer_norm_2.weight"], sd[pre + "conv_layer.layer_norm_2.bias"], 1e-5))
    h = F.dropout(h, p, training)
    out = F.linear(h, sd[pre + "linear_layer.weight"], sd[pre + "linear_layer.bias"]).squeeze(-1)
    if pad_mask is not None:
        out = out.masked_fill(pad_mask, 0.0)
    return out


def length_regulate(x, durations, max_len):
    """model/modules.py:167-194 + utils/tools.py:299-317: repeat row i max(int(d_i),0) times, pad/crop to max_len,
    mel_len = un-cropped length."""
    outs, lens = [], []
    for xb, db in zip(x, durations):
        reps = torch.clamp(db.to(torch.float64).trunc().to(torch.int64), min=0)  # int(): truncation toward zero
        e = torch.repeat_interleave(xb, reps, dim=0)
        lens.append(e.shape[0])
        outs.append(e)
    if max_len is None:
        max_len = max(lens)
    padded = []
    for e in outs:
        if e.shape[0] >= max_len:
            padded.append(e[:max_len])
        else:
            padded.append(F.pad(e, (0, 0, 0, max_len - e.shape[0])))
    return torch.stack(padded), torch.tensor(lens, dtype=torch.int64)


def variance_adaptor(sd, cfg, pcfg, x, src_mask, mel_mask, max_len, p_target, e_target, d_target, p_control,
                     e_control, d_control, training, dropout):
    """model/modules.py:102-158.  NB the energy embedding is called with p_control (modules.py:124,146)."""
    pitch_level = pcfg["preprocessing"]["pitch"]["feature"]
    energy_level = pcfg["preprocessing"]["energy"]["feature"]
    pre = "variance_adaptor."

    def embed(kind, h, target, mask, control):
        pred = variance_predictor(sd, f"{pre}{kind}_predictor.", cfg, h, mask, training, dropout)
        bins = sd[f"{pre}{kind}_bins"]
        if target is not None:
            idx = torch.bucketize(target, bins)
        else:
            pred = pred * control
            idx = torch.bucketize(pred.detach() if False else pred, bins)
        return pred, F.embedding(idx, sd[f"{pre}{kind}_embedding.weight"])

    log_d = variance_predictor(sd, pre + "duration_predictor.", cfg, x, src_mask, training, dropout)
    p_pred = e_pred = None
    if pitch_level == "phoneme_level":
        p_pred, emb = embed("pitch", x, p_target, src_mask, p_control)
        x = _st(x + emb)
    if energy_level == "phoneme_level":
        e_pred, emb = embed("energy", x, e_target, src_mask, p_control)
        x = _st(x + emb)
    if d_target is not None:
        x, mel_len = length_regulate(x, d_target, max_len)
        d_rounded = d_target
    else:
        d_rounded = torch.clamp(torch.round(torch.exp(log_d) - 1) * d_control, min=0)
        x, mel_len = length_regulate(x, d_rounded, max_len)
        mel_mask = mask_from_lengths(mel_len)
    if pitch_level == "frame_level":
        p_pred, emb = embed("pitch", x, p_target, mel_mask, p_control)
        x = _st(x + emb)
    if energy_level == "frame_level":
        e_pred, emb = embed("energy", x, e_target, mel_mask, p_control)
        x = _st(x + emb)
    return x, p_pred, e_pred, log_d, d_rounded, mel_len, mel_mask


# ------------------------------------------------------------------------------------------------ postnet + model
def postnet(sd, x, training, dropout, bn_buffers=None, n_layers=5):
    """transformer/Layers.py:129-137 — BatchNorm uses batch statistics over ALL B*T positions in training."""
    p = 0.5 if dropout else 0.0
    h = x.transpose(1, 2)
    for i in range(n_layers):
        pre = f"postnet.convolutions.{i}."
        w = sd[pre + "0.conv.weight"]
        h = _st(F.conv1d(h, w, sd[pre + "0.conv.bias"], padding=(w.shape[2] - 1) // 2))
        rm, rv = sd[pre + "1.running_mean"], sd[pre + "1.running_var"]
        if training and bn_buffers is not None:
            rm, rv = bn_buffers[pre + "1.running_mean"], bn_buffers[pre + "1.running_var"]
        elif training:
            rm, rv = rm.clone(), rv.clone()
        h = F.batch_norm(h, rm, rv, sd[pre + "1.weight"], sd[pre + "1.bias"], training, 0.1, 1e-5)
        if i < n_layers - 1:
            h = _st(torch.tanh(h))
        h = F.dropout(h, p, training)
    return h.transpose(1, 2)


def fastspeech2_forward(sd, cfg, pcfg, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None,
                        max_mel_len=None, p_targets=None, e_targets=None, d_targets=None, p_control=1.0, e_control=1.0,
                        d_control=1.0, training=False, dropout=False, bn_buffers=None):
    """model/fastspeech2.py:43-110.  `dropout=False` with training=True gives the deterministic train-mode path
    (BatchNorm batch statistics live, dropout neutralised) used for gradient parity."""
    src_masks = mask_from_lengths(src_lens, max_src_len)
    mel_masks = mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
    x = encoder(sd, cfg, texts, src_masks, training, dropout)
    if cfg["multi_speaker"]:
        x = x + F.embedding(speakers, sd["speaker_emb.weight"]).unsqueeze(1)
    x, p_pred, e_pred, log_d, d_rounded, mel_lens, mel_masks = variance_adaptor(
        sd, cfg, pcfg, x, src_masks, mel_masks, max_mel_len, p_targets, e_targets, d_targets, p_control, e_control,
        d_control, training, dropout)
    x, mel_masks = decoder(sd, cfg, x, mel_masks, training, dropout)
    mel = _st(F.linear(x, sd["mel_linear.weight"], sd["mel_linear.bias"]))
    post = _st(postnet(sd, mel, training, dropout, bn_buffers) + mel)
    return mel, post, p_pred, e_pred, log_d, d_rounded, src_masks, mel_masks, src_lens, mel_lens


def fastspeech2_loss(pcfg, batch_targets, predictions):
    """model/loss.py:19-92.  batch_targets = (mels, pitches, energies, durations)."""
    mel_t, p_t, e_t, d_t = batch_targets
    mel, post, p_pred, e_pred, log_d, _, src_masks, mel_masks, _, _ = predictions
    src_v, mel_v = ~src_masks, ~mel_masks
    log_d_t = torch.log(d_t.float() + 1).to(log_d.dtype)
    mel_t = mel_t[:, : mel_v.shape[1], :]
    pm = src_v if pcfg["preprocessing"]["pitch"]["feature"] == "phoneme_level" else mel_v
    em = src_v if pcfg["preprocessing"]["energy"]["feature"] == "phoneme_level" else mel_v
    pitch_loss = F.mse_loss(p_pred.masked_select(pm), p_t.masked_select(pm))
    energy_loss = F.mse_loss(e_pred.masked_select(em), e_t.masked_select(em))
    dur_loss = F.mse_loss(log_d.masked_select(src_v), log_d_t.masked_select(src_v))
    mv = mel_v.unsqueeze(-1)
    mel_loss = F.l1_loss(mel.masked_select(mv), mel_t.masked_select(mv))
    post_loss = F.l1_loss(post.masked_select(mv), mel_t.masked_select(mv))
    total = mel_loss + post_loss + dur_loss + pitch_loss + energy_loss
    return total, mel_loss, post_loss, pitch_loss, energy_loss, dur_loss


def lr_at_step(step, d_model, warmup, anneal_steps, anneal_rate):
    """model/optimizer.py:33-51 (step is the already-incremented current_step)."""
    lr = min(step ** -0.5, warmup ** -1.5 * step)
    for s in anneal_steps:
        if step > s:
            lr *= anneal_rate
    return d_model ** -0.5 * lr


# ------------------------------------------------------------------------------------------------ HiFi-GAN
def hifigan_forward(sd, h, mel):
    """hifigan/models.py:149-165 with weight norm already removed (plain `weight` keys).
    mel: (B, 80, T) -> (B, 1, 256 T).  Last leaky_relu uses the DEFAULT slope 0.01 (models.py:161)."""
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            pre = f"resblocks.{i * nk + j}."
            y = x
            for m, d in enumerate(rd):
                t = F.leaky_relu(y, 0.1)
                t = F.conv1d(t, sd[f"{pre}convs1.{m}.weight"], sd[f"{pre}convs1.{m}.bias"], dilation=d, padding=(rk * d - d) // 2)
                t = F.leaky_relu(t, 0.1)
                t = F.conv1d(t, sd[f"{pre}convs2.{m}.weight"], sd[f"{pre}convs2.{m}.bias"], padding=(rk - 1) // 2)
                y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def bf16_store(t):
    """round to bfloat16 (nearest even) and back: what a product tensor STORED in bf16 holds."""
    return t.to(torch.bfloat16).to(t.dtype)


def hifigan_forward_stored(sd, h, mel, store=None, stages=None, stage_inputs=None, weight_store=None, fused_max_channels=64):
    """hifigan_forward with the PRODUCT's storage points made explicit (fastspeech2_amd/hifigan.py `_run`, round-5 convention): `store`
    is applied wherever the product writes an activation to HBM (or, inside a fused residual block, to LDS) in its compute dtype,
    `weight_store` to the packed weights (conv_post's stay fp32).
      * every convolution of the generator reads leaky_relu(x, 0.1), so the chains of single launches store leaky_relu(value) ONCE
        (no separate stored x + re-rounded prologue operand as in rounds 1-4) and undo it where the raw value is needed: the
        residual add of a block's conv2 (`unl`: a > 0 ? a : a / 0.1);
      * stages whose channel count is <= fused_max_channels run their residual blocks fused (fs2_resblock.hip): the running sum y is
        NOT rounded between the three pairs, only the convolutions' operands (leaky_relu(y), t) are, and the stage's running xs is
        rounded after every block; the stage hands leaky_relu(xs) to the next up-sampling convolution, raw xs to conv_post.
    With store = weight_store = None this is hifigan_forward up to summation order (tests/test_oracle_golden.py); with
    bf16_store it is the bf16 EMULATION the bf16 bars are derived from (the product differs from it by accumulation order only).
    stages: list that receives the RAW output of conv_pre and of every up-sampling stage; stage_inputs: RAW tensors to use as the
    INPUT of stage i instead (i = 0: conv_pre <- mel ... len-1: conv_post) - per-stage comparisons that do not compound."""
    st = store if store is not None else (lambda t: t)
    wst = weight_store if weight_store is not None else (lambda t: t)
    lre = lambda t: F.leaky_relu(t, 0.1)
    unl = lambda a: torch.where(a > 0, a, a * 10.0)

    def inp(i, x):
        return stage_inputs[i] if (stage_inputs is not None and stage_inputs[i] is not None) else x

    x = F.conv1d(st(inp(0, mel)), wst(sd["conv_pre.weight"]), sd["conv_pre.bias"], padding=3)
    xl = st(lre(x))                                         # what is stored; unl(xl) is the raw value the product can still recover
    if stages is not None:
        stages.append(unl(xl))
    nk = len(h["resblock_kernel_sizes"])
    nup = len(h["upsample_rates"])
    x = None
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        if stage_inputs is not None and stage_inputs[1 + i] is not None:
            xl = st(lre(stage_inputs[1 + i]))               # (re-rounds to exactly the stored value when the input is the product's probe)
        up = F.conv_transpose1d(xl, wst(sd[f"ups.{i}.weight"]), sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        fused = up.shape[1] <= fused_max_channels
        last = i + 1 == nup
        xs = None
        x0 = st(up) if fused else st(lre(up))               # fused stages keep raw rows, the chains lrelu'd ones
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            pre = f"resblocks.{i * nk + j}."
            if fused:
                y = x0
                for m, d in enumerate(rd):
                    t = F.conv1d(st(lre(y)), wst(sd[f"{pre}convs1.{m}.weight"]), sd[f"{pre}convs1.{m}.bias"], dilation=d, padding=(rk * d - d) // 2)
                    y = F.conv1d(st(lre(t)), wst(sd[f"{pre}convs2.{m}.weight"]), sd[f"{pre}convs2.{m}.bias"], padding=(rk - 1) // 2) + y
                xs = st(y / nk) if xs is None else st(xs + y / nk)
            else:
                cur = x0                                    # lrelu'd
                for m, d in enumerate(rd):
                    t = F.conv1d(cur, wst(sd[f"{pre}convs1.{m}.weight"]), sd[f"{pre}convs1.{m}.bias"], dilation=d, padding=(rk * d - d) // 2)
                    t = st(lre(t))
                    ynew = F.conv1d(t, wst(sd[f"{pre}convs2.{m}.weight"]), sd[f"{pre}convs2.{m}.bias"], padding=(rk - 1) // 2) + unl(cur)
                    if m < len(rd) - 1:
                        cur = st(lre(ynew))
                    else:
                        v = ynew / nk if xs is None else xs + ynew / nk
                        # the stage's last launch stores what the next consumer reads: lrelu'd for an up-sampling conv, raw for conv_post
                        xs = st(lre(v)) if (j == nk - 1 and not last) else st(v)
        if fused and not last:
            xl = st(lre(xs))
            x = unl(xl)
        elif not last:
            xl = xs
            x = unl(xl)
        else:
            x = xs
        if stages is not None:
            stages.append(x)
    x = inp(1 + nup, x)
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def remove_weight_norm_sd(sd):
    """weight_g / weight_v -> weight = g * v / ||v|| (norm over all dims but 0), as torch.nn.utils.weight_norm."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_v"):
            g = sd[k[:-1] + "g"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            out[k[:-2]] = v * (g / norm)
        elif k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out


def pcm16(wav, max_wav_value=32768.0):
    """utils/model.py:82-85 — numpy astype('int16') after scaling: truncation toward zero, no clipping."""
    return (wav.detach().cpu().numpy() * max_wav_value).astype("int16")


# ------------------------------------------------------------------------------------------------ STFT / mel
def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa==0.7.2 filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm=1 -> Slaney area norm).
    Third-party algorithm restated from its published definition (librosa is not vendored in the reference;
    pinned at requirements.txt:3, called at audio/stft.py:145-147)."""
    if fmax is None:
        fmax = sr / 2.0

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)

    n_freq = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2.0, n_freq)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    # librosa builds the table in its OUTPUT dtype (float32): the float64 triangles are rounded to float32 when they are
    # assigned, the area normalisation then multiplies the float32 table by the float64 `enorm` (numpy computes float32 *
    # float64 in float64) and the in-place result is rounded to float32 again - two roundings, reproduced here in that order
    weights = np.zeros((n_mels, n_freq), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights


def stft_basis(filter_length, win_length):
    """audio/stft.py:26-50: [Re; Im] of FFT(I)[:cutoff] times a periodic hann window, float32."""
    from scipy.signal import get_window

    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    win = get_window("hann", win_length, fftbins=True)
    if win_length < filter_length:  # librosa.util.pad_center
        lpad = (filter_length - win_length) // 2
        win = np.pad(win, (lpad, filter_length - win_length - lpad))
    basis = torch.FloatTensor(fb) * torch.from_numpy(win).float()
    return basis  # (2*cutoff, filter_length)


def mel_spectrogram(y, filter_length=1024, hop_length=256, win_length=1024, n_mel=80, sr=22050, fmin=0, fmax=8000,
                    mel_basis=None):
    """audio/stft.py:159-178 (+ :52-81 transform, audio_processing.py:85-91): y (B, N) in [-1,1] ->
    (mel (B, n_mel, frames), energy (B, frames))."""
    assert y.min() >= -1 and y.max() <= 1
    basis = stft_basis(filter_length, win_length)
    x = F.pad(y.unsqueeze(1), (filter_length // 2, filter_length // 2), mode="reflect")
    ft = F.conv1d(x, basis.unsqueeze(1), stride=hop_length)
    cutoff = filter_length // 2 + 1
    mag = torch.sqrt(ft[:, :cutoff] ** 2 + ft[:, cutoff:] ** 2)
    if mel_basis is None:
        mel_basis = torch.from_numpy(slaney_mel_filterbank(sr, filter_length, n_mel, fmin, fmax))
    mel = torch.log(torch.clamp(torch.matmul(mel_basis, mag), min=1e-5))
    energy = torch.norm(mag, dim=1)
    return mel, energy
