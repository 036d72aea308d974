"""Drop-in Python surface of the reference's acoustic model (model/fastspeech2.py, model/loss.py,
model/optimizer.py) over the HIP engine.

* `FastSpeech2(preprocess_config, model_config)` — same constructor, same positional `forward` signature and
  10-tuple result (reference model/fastspeech2.py:16-110), same `state_dict()` key schema and parameter order
  (SURVEY Appendix C), so reference checkpoints load unchanged.
* `FastSpeech2Loss`, `ScheduledOptim` — reference model/loss.py:5-92, model/optimizer.py:5-51.

Underneath, every parameter is a view into ONE flat fp32 buffer (and its gradient into one flat gradient
buffer) ordered by backward completion, so the data-parallel exchange is a handful of large contiguous
all-reduces and clip+Adam is two kernel launches.  There is no PyTorch/CPU compute fallback: `forward` on a
non-GPU tensor raises.
"""
import json
import math
import os

import numpy as np

import torch
import torch.nn as nn

from . import ops
from .engine import Engine

# reference text/symbols.py:21-29: 1 pad + 1 special + 10 punctuation + 52 letters + 84 ARPAbet + 209 pinyin + 3 silences
N_SYMBOLS = 360


def sinusoid_table(n_position, d_hid):
    """reference transformer/Models.py:10-30 (float64 angles -> float32)."""
    import numpy as np

    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)
    table = np.empty_like(angle)
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.from_numpy(table).float()


class _Node(nn.Module):
    """Pure parameter container: the module tree only reproduces the reference's state_dict key hierarchy."""


def _p(t, requires_grad=True):
    return nn.Parameter(t, requires_grad=requires_grad)


def _linear_like(node, wshape, fan_in):
    w = torch.empty(*wshape)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear / nn.Conv1d default init
    bound = 1.0 / math.sqrt(fan_in)
    node.weight = _p(w)
    node.bias = _p(torch.empty(wshape[0]).uniform_(-bound, bound))
    return node


def _linear(out_f, in_f):
    return _linear_like(_Node(), (out_f, in_f), in_f)


def _conv(out_c, in_c, k):
    return _linear_like(_Node(), (out_c, in_c, k), in_c * k)


def _layernorm(c):
    n = _Node()
    n.weight = _p(torch.ones(c))
    n.bias = _p(torch.zeros(c))
    return n


def _fft_block(d, n_head, d_inner, ks):
    blk = _Node()
    a = _Node()
    a.w_qs, a.w_ks, a.w_vs = _linear(d, d), _linear(d, d), _linear(d, d)
    a.layer_norm = _layernorm(d)
    a.fc = _linear(d, d)
    blk.slf_attn = a
    f = _Node()
    f.w_1 = _conv(d_inner, d, ks[0])
    f.w_2 = _conv(d, d_inner, ks[1])
    f.layer_norm = _layernorm(d)
    blk.pos_ffn = f
    return blk


def _variance_predictor(d_in, filt, k):
    vp = _Node()
    cl = _Node()
    c1 = _Node(); c1.conv = _conv(filt, d_in, k)
    cl.conv1d_1 = c1
    cl.layer_norm_1 = _layernorm(filt)
    c2 = _Node(); c2.conv = _conv(filt, filt, k)
    cl.conv1d_2 = c2
    cl.layer_norm_2 = _layernorm(filt)
    vp.conv_layer = cl
    vp.linear_layer = _linear(1, filt)
    return vp


class FastSpeech2(nn.Module):
    """reference model/fastspeech2.py:13-110."""

    def __init__(self, preprocess_config, model_config, compute_dtype=None):
        super().__init__()
        self.model_config = model_config
        self.preprocess_config = preprocess_config
        tc = model_config["transformer"]
        d = tc["encoder_hidden"]
        assert tc["decoder_hidden"] == d, "encoder/decoder hidden sizes must match (as in every reference config)"
        n_position = model_config["max_seq_len"] + 1
        n_vocab = model_config.get("n_src_vocab", N_SYMBOLS + 1)
        ks = tc["conv_kernel_size"]

        # ---- encoder (transformer/Models.py:36-71)
        enc = _Node()
        emb = _Node()
        w = torch.empty(n_vocab, d).normal_()
        w[0].zero_()                                               # padding_idx = 0
        emb.weight = _p(w)
        enc.src_word_emb = emb
        enc.position_enc = _p(sinusoid_table(n_position, d).unsqueeze(0), requires_grad=False)
        enc.layer_stack = nn.ModuleList([_fft_block(d, tc["encoder_head"], tc["conv_filter_size"], ks)
                                         for _ in range(tc["encoder_layer"])])
        self.encoder = enc

        # ---- variance adaptor (model/modules.py:20-78)
        pp = preprocess_config["preprocessing"]
        self.pitch_feature_level = pp["pitch"]["feature"]
        self.energy_feature_level = pp["energy"]["feature"]
        assert self.pitch_feature_level in ["phoneme_level", "frame_level"]
        assert self.energy_feature_level in ["phoneme_level", "frame_level"]
        ve = model_config["variance_embedding"]
        assert ve["pitch_quantization"] in ["linear", "log"]
        assert ve["energy_quantization"] in ["linear", "log"]
        n_bins = ve["n_bins"]
        with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "stats.json")) as f:
            stats = json.load(f)
        pitch_min, pitch_max = stats["pitch"][:2]
        energy_min, energy_max = stats["energy"][:2]

        def bins(kind, lo, hi):
            import numpy as np

            if kind == "log":
                return torch.exp(torch.linspace(np.log(lo), np.log(hi), n_bins - 1))
            return torch.linspace(lo, hi, n_bins - 1)

        vp_cfg = model_config["variance_predictor"]
        va = _Node()
        va.duration_predictor = _variance_predictor(d, vp_cfg["filter_size"], vp_cfg["kernel_size"])
        va.pitch_predictor = _variance_predictor(d, vp_cfg["filter_size"], vp_cfg["kernel_size"])
        va.energy_predictor = _variance_predictor(d, vp_cfg["filter_size"], vp_cfg["kernel_size"])
        va.pitch_bins = _p(bins(ve["pitch_quantization"], pitch_min, pitch_max), requires_grad=False)
        va.energy_bins = _p(bins(ve["energy_quantization"], energy_min, energy_max), requires_grad=False)
        pe = _Node(); pe.weight = _p(torch.empty(n_bins, d).normal_())
        ee = _Node(); ee.weight = _p(torch.empty(n_bins, d).normal_())
        va.pitch_embedding, va.energy_embedding = pe, ee
        self.variance_adaptor = va

        # ---- decoder (transformer/Models.py:106-137)
        dec = _Node()
        dec.position_enc = _p(sinusoid_table(n_position, d).unsqueeze(0), requires_grad=False)
        dec.layer_stack = nn.ModuleList([_fft_block(d, tc["decoder_head"], tc["conv_filter_size"], ks)
                                         for _ in range(tc["decoder_layer"])])
        self.decoder = dec

        n_mel = pp["mel"]["n_mel_channels"]
        self.mel_linear = _linear(n_mel, d)

        # ---- postnet (transformer/Layers.py:72-127: 80 -> 512 x3 -> 80, k=5, BatchNorm1d)
        pn = _Node()
        convs = nn.ModuleList()
        chans = [n_mel, 512, 512, 512, 512, n_mel]
        for i in range(5):
            seq = nn.ModuleList()
            cn = _Node(); cn.conv = _conv(chans[i + 1], chans[i], 5)
            bn = _Node()
            bn.weight = _p(torch.ones(chans[i + 1])); bn.bias = _p(torch.zeros(chans[i + 1]))
            bn.register_buffer("running_mean", torch.zeros(chans[i + 1]))
            bn.register_buffer("running_var", torch.ones(chans[i + 1]))
            bn.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
            seq.append(cn); seq.append(bn)
            convs.append(seq)
        pn.convolutions = convs
        self.postnet = pn

        self.speaker_emb = None
        if model_config["multi_speaker"]:
            with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")) as f:
                n_speaker = len(json.load(f))
            se = _Node(); se.weight = _p(torch.empty(n_speaker, d).normal_())
            self.speaker_emb = se

        if compute_dtype is None:
            compute_dtype = os.environ.get("FS2_DTYPE", "fp32")
        if isinstance(compute_dtype, str):
            compute_dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                             "bfloat16": torch.bfloat16}[compute_dtype]
        self.compute_dtype = compute_dtype
        # shape limits of the hand-written kernels, checked at CONSTRUCTION (not in the middle of the first backward)
        vp = model_config["variance_predictor"]
        problems = []
        for side in ("encoder", "decoder"):
            if d % tc[side + "_head"] or d // tc[side + "_head"] != 128:
                problems.append(f"{side}_hidden / {side}_head = {d}/{tc[side + '_head']} (attention kernels: head size 128 only)")
        if d % 256:
            problems.append(f"encoder_hidden = {d} (LayerNorm backward: multiples of 256)")
        if vp["filter_size"] % 256:
            problems.append(f"variance_predictor.filter_size = {vp['filter_size']} (LayerNorm backward: multiples of 256)")
        if tc["conv_filter_size"] % 8 or d % 8:
            problems.append("channel counts must be multiples of 8 (16-byte rows)")
        if len(ks) != 2 or any(k % 2 == 0 for k in ks):
            problems.append(f"conv_kernel_size = {ks} (two odd kernel sizes)")
        if problems:
            raise ValueError("FastSpeech2: configuration outside the shapes the HIP kernels are built for: " + "; ".join(problems))
        self._dropout_step = 0          # resume: how many training forwards came before (fast-forwards the dropout counter)
        self._engine = None
        self._flat = None
        self._flat_grad = None
        self._flat_names = None
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    # ------------------------------------------------------------------ flat parameter storage
    def _trainable_in_backward_order(self):
        """(name, param) pairs ordered by when their gradient completes in backward: postnet (last conv first),
        mel_linear, decoder layers N-1..0, variance adaptor, [speaker], encoder layers N-1..0, embedding.
        QKV weights (and biases) are adjacent so the fused [3d, d] projection is one matrix."""
        cached = getattr(self, "_tibo", None)
        if cached is not None:
            return cached
        named = dict(self.named_parameters())
        order = []

        def add(prefix, names):
            for n in names:
                order.append(prefix + n)

        for i in reversed(range(5)):
            add(f"postnet.convolutions.{i}.", ["0.conv.weight", "0.conv.bias", "1.weight", "1.bias"])
        add("mel_linear.", ["weight", "bias"])

        def fft(prefix):
            add(prefix + "pos_ffn.", ["layer_norm.weight", "layer_norm.bias", "w_2.weight", "w_2.bias", "w_1.weight", "w_1.bias"])
            add(prefix + "slf_attn.", ["layer_norm.weight", "layer_norm.bias", "fc.weight", "fc.bias",
                                       "w_qs.weight", "w_ks.weight", "w_vs.weight", "w_qs.bias", "w_ks.bias", "w_vs.bias"])

        for i in reversed(range(len(self.decoder.layer_stack))):
            fft(f"decoder.layer_stack.{i}.")
        for kind in ("energy", "pitch", "duration"):
            pre = f"variance_adaptor.{kind}_predictor."
            add(pre, ["linear_layer.weight", "linear_layer.bias", "conv_layer.layer_norm_2.weight", "conv_layer.layer_norm_2.bias",
                      "conv_layer.conv1d_2.conv.weight", "conv_layer.conv1d_2.conv.bias", "conv_layer.layer_norm_1.weight",
                      "conv_layer.layer_norm_1.bias", "conv_layer.conv1d_1.conv.weight", "conv_layer.conv1d_1.conv.bias"])
        add("variance_adaptor.", ["energy_embedding.weight", "pitch_embedding.weight"])
        if self.speaker_emb is not None:
            order.append("speaker_emb.weight")
        for i in reversed(range(len(self.encoder.layer_stack))):
            fft(f"encoder.layer_stack.{i}.")
        order.append("encoder.src_word_emb.weight")
        trainable = [n for n, p in named.items() if p.requires_grad]
        assert sorted(order) == sorted(trainable), "flat layout does not cover the trainable parameters"
        self._tibo = [(n, named[n]) for n in order]      # the parameter set is fixed after construction
        return self._tibo

    def _invalidate(self, lowp_synced=False):
        """parameters changed: the engine must refresh its packed weights (lowp_synced: the bf16 shadow copy of the
        flat buffer is already up to date - the Adam kernel wrote it)."""
        if self._engine is not None:
            self._engine.weights_dirty = True
            self._engine.lp_synced = bool(lowp_synced)

    def _ensure_flat(self, device):
        """(Re)build the flat fp32 parameter / gradient buffers on `device` and alias every Parameter to it."""
        pairs = self._trainable_in_backward_order()
        first = pairs[0][1]
        if (self._flat is not None and self._flat.device == device and first.data_ptr() == self._flat.data_ptr()
                and first.device == device):
            return
        offsets, total = {}, 0
        for n, p in pairs:
            offsets[n] = total
            total += (p.numel() + 7) // 8 * 8            # every view 16-byte aligned in fp32 AND in the bf16 shadow copy
        flat = torch.zeros(total, device=device, dtype=torch.float32)
        grad = torch.zeros(total, device=device, dtype=torch.float32)
        for n, p in pairs:
            view = self._view(flat, offsets[n], p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = None
        for n, p in self.named_parameters():           # position tables / bins: plain device tensors
            if not p.requires_grad and p.device != device:
                p.data = p.data.to(device)
        for n, b in self.named_buffers():
            if b.device != device:
                b.data = b.data.to(device)
        self._flat, self._flat_grad, self._flat_offsets = flat, grad, offsets
        self._flat_names = [n for n, _ in pairs]
        self._engine = Engine(self, device)
        self._engine.reseed(step=self._dropout_step)

    @staticmethod
    def _view(buf, offset, shape):
        """Parameter view into a flat buffer.  Conv weights (Cout, Cin, k>1) are STORED tap-major [Cout][k][Cin]
        (the K-contiguous order the MFMA contraction and the coalesced weight-gradient epilogue want); the tensor
        handed to torch is the permuted view, so state_dict()/load_state_dict() still see (Cout, Cin, k)."""
        n = 1
        for s_ in shape:
            n *= s_
        if len(shape) == 3 and shape[2] > 1:
            return buf[offset:offset + n].view(shape[0], shape[2], shape[1]).permute(0, 2, 1)
        return buf[offset:offset + n].view(shape)

    def flat_parameters(self):
        return self._flat

    def flat_gradients(self):
        return self._flat_grad

    def grad_view(self, name):
        p = dict(self.named_parameters())[name]
        return self._view(self._flat_grad, self._flat_offsets[name], p.shape)

    def attach_grads(self):
        """Point every trainable Parameter's .grad at its slice of the flat gradient buffer."""
        for n, p in self._trainable_in_backward_order():
            p.grad = self._view(self._flat_grad, self._flat_offsets[n], p.shape)

    def train(self, mode=True):
        self._invalidate()
        if self._engine is not None:
            # training writes the BatchNorm running statistics from the kernels (raw pointers: no torch _version bump), so the
            # eval-mode [mean | rstd] cache is dropped on every mode switch - eval() after training always rebuilds it
            self._engine._bn_eval.clear()
        return super().train(mode)

    def set_length_hint(self, src_lens_host, mel_lens_host=None):
        """Host copies (numpy / list / CPU tensor) of the NEXT forward's src_lens and mel_lens.  The reference's positional forward
        only carries device tensors; whether the contractions skip wholly padded tiles (Engine.lens_skip_min) is decided on the
        host, without a device round trip, from these - or, when this is not called, from the copy `utils.to_device` attaches to
        the lengths tensors it makes.  Neither present: no tile skipping (results are identical either way).  A captured hipGraph
        keeps the choice made at capture time."""
        if self._engine is None and next(self.parameters()).is_cuda:
            self._ensure_flat(next(self.parameters()).device)
        if self._engine is not None:
            self._engine.length_hint = (None if src_lens_host is None else np.asarray(src_lens_host),
                                        None if mel_lens_host is None else np.asarray(mel_lens_host))

    # ------------------------------------------------------------------ forward (reference signature)
    def forward(self, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                p_targets=None, e_targets=None, d_targets=None, p_control=1.0, e_control=1.0, d_control=1.0):
        if not texts.is_cuda:
            raise RuntimeError("fastspeech2_amd.FastSpeech2 runs on an AMD GPU only (no CPU fallback): move the "
                               "batch with to_device(batch, torch.device('cuda'))")
        self._ensure_flat(texts.device)
        return self._engine.run(speakers, texts, src_lens, int(max_src_len), mels, mel_lens,
                                None if max_mel_len is None else int(max_mel_len), p_targets, e_targets, d_targets,
                                float(p_control), float(e_control), float(d_control))


class _FusedLoss(torch.autograd.Function):
    """All five loss terms in one pass over the padded tensors (fs2_loss_fwd), gradients in one more (fs2_loss_bwd)."""

    @staticmethod
    def forward(ctx, mel, post, p_pred, e_pred, logd, mel_t, mel_lens, src_lens, p_t, e_t, dur, cnt, p_frame, e_frame):
        mel, post, p_pred, e_pred, logd = (t.contiguous() for t in (mel, post, p_pred, e_pred, logd))
        losses = ops.loss_fwd(mel, post, mel_t, mel_lens, src_lens, p_pred, p_t, e_pred, e_t, logd, dur, cnt, p_frame, e_frame)
        ctx.save_for_backward(mel, post, p_pred, e_pred, logd, mel_t, mel_lens, src_lens, p_t, e_t, dur, cnt)
        ctx.flags = (p_frame, e_frame)
        return losses

    @staticmethod
    def backward(ctx, g):
        mel, post, p_pred, e_pred, logd, mel_t, mel_lens, src_lens, p_t, e_t, dur, cnt = ctx.saved_tensors
        grads = ops.loss_bwd(mel, post, mel_t, mel_lens, src_lens, p_pred, p_t, e_pred, e_t, logd, dur, cnt,
                             g.contiguous().float(), *ctx.flags)
        return grads + (None,) * 9


class FastSpeech2Loss(nn.Module):
    """reference model/loss.py:5-92: masked L1 on mel / post-net mel, masked MSE on pitch / energy / log-duration, and
    their sum, as HIP kernels (fs2_loss.hip) behind one autograd node.  Returns the reference's 6-tuple
    (total, mel, postnet_mel, pitch, energy, duration) of 0-dim tensors."""

    def __init__(self, preprocess_config, model_config, count_reduce=None):
        super().__init__()
        self.pitch_feature_level = preprocess_config["preprocessing"]["pitch"]["feature"]
        self.energy_feature_level = preprocess_config["preprocessing"]["energy"]["feature"]
        # data-parallel runs pass ddp.global_counts so that every rank normalises by (global valid count / world):
        # the rank-averaged gradient then equals the reference's global-batch mean (train.py:82-86)
        self.count_reduce = count_reduce

    def forward(self, inputs, predictions):
        mel_targets, _, _, pitch_targets, energy_targets, duration_targets = inputs[6:]
        (mel_pred, post_pred, pitch_pred, energy_pred, logd_pred, _, src_masks, mel_masks, src_lens, mel_lens) = predictions
        if not mel_pred.is_cuda:
            raise RuntimeError("FastSpeech2Loss runs on an AMD GPU only (no CPU fallback)")
        T, L = mel_pred.shape[1], logd_pred.shape[1]
        # valid positions exactly as the masks the model returned (utils/tools.py:91-99): t < min(len, padded length)
        src_lens = src_lens.to(torch.int64)
        mel_lens = mel_lens.to(torch.int64)
        pre = getattr(predictions[9], "_fs2_counts", None)       # computed next to the masks by the model's own forward
        if pre is not None and pre[1] is predictions[8] and pre[2] == L and pre[3] == T:
            counts = pre[0]
        else:
            counts = torch.stack([src_lens.clamp(max=L).sum(), mel_lens.clamp(max=T).sum()]).float()
        if self.count_reduce is not None:
            counts = self.count_reduce(counts)
        mel_t = mel_targets if mel_targets.dtype == torch.float32 else mel_targets.float()
        assert mel_t.shape[1] >= T and mel_t.stride(2) == 1 and mel_t.stride(1) == mel_t.shape[2], "mel targets must be (B, >=T, n_mel) rows"
        p_t = pitch_targets.float() if pitch_targets.dtype != torch.float32 else pitch_targets
        e_t = energy_targets.float() if energy_targets.dtype != torch.float32 else energy_targets
        dur = duration_targets if duration_targets.dtype == torch.int64 else duration_targets.long()
        assert p_t.stride(1) == 1 and e_t.stride(1) == 1 and dur.stride(1) == 1
        losses = _FusedLoss.apply(mel_pred, post_pred, pitch_pred, energy_pred, logd_pred, mel_t, mel_lens, src_lens, p_t, e_t,
                                  dur, counts, self.pitch_feature_level != "phoneme_level",
                                  self.energy_feature_level != "phoneme_level")
        return tuple(losses.unbind(0))


class ScheduledOptim:
    """reference model/optimizer.py:5-51 (Noam warm-up + annealing around Adam) with train.py:93's
    clip_grad_norm_ folded in: one sum-of-squares launch + one fused clip/Adam launch over the flat buffers.
    `state_dict()/load_state_dict()` speak torch.optim.Adam's format (per-parameter exp_avg / exp_avg_sq in
    `model.parameters()` order), so reference checkpoints' "optimizer" entry round-trips."""

    def __init__(self, model, train_config, model_config, current_step):
        self.model = model
        oc = train_config["optimizer"]
        self.betas = tuple(oc["betas"])
        self.eps = oc["eps"]
        self.weight_decay = oc["weight_decay"]
        self.grad_clip_thresh = oc.get("grad_clip_thresh", 0.0)
        self.n_warmup_steps = oc["warm_up_step"]
        self.anneal_steps = oc["anneal_steps"]
        self.anneal_rate = oc["anneal_rate"]
        self.current_step = current_step
        self.init_lr = model_config["transformer"]["encoder_hidden"] ** -0.5
        self._m = self._v = self._hyper = self._nsq = None
        self._adam_step = 0      # Adam's own step count (bias correction); equals number of updates performed
        self._pending_state = None
        self.fold_clip = True    # clip inside step_and_update_lr (train.py calls clip_grad_norm_ separately: see clip())

    # -- schedule (optimizer.py:33-51)
    def _get_lr_scale(self):
        lr = min(self.current_step ** -0.5, self.n_warmup_steps ** -1.5 * self.current_step)
        for s in self.anneal_steps:
            if self.current_step > s:
                lr = lr * self.anneal_rate
        return lr

    def _ensure(self):
        flat = self.model.flat_parameters()
        if flat is None:
            raise RuntimeError("ScheduledOptim: run the model once (or call model._ensure_flat(device)) before stepping")
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._hyper = torch.zeros(4, device=flat.device, dtype=torch.float32)
            # lr / bias corrections travel through a ring of PINNED host slots: the copy is truly asynchronous (no per-step
            # host sync, no reliance on the runtime staging a pageable temporary) and a slot is rewritten only after the
            # copy that read it has executed
            self._hyper_host = torch.zeros(16, 4, dtype=torch.float32).pin_memory() if flat.is_cuda else None
            self._hyper_ev = [None] * 16
            self._hyper_i = 0
            self._nsq = torch.zeros(1, device=flat.device, dtype=torch.float32)
            self._nsq_ws = torch.empty(1024, device=flat.device, dtype=torch.float32)
            if self._pending_state is not None:
                self._load_now(self._pending_state)
                self._pending_state = None

    def step_and_update_lr(self, zero_grad=False):
        """optimizer.py:22-24.  zero_grad=True also clears the gradients inside the Adam pass (then skip zero_grad())."""
        self._ensure()
        self.current_step += 1
        self._adam_step += 1
        lr = self.init_lr * self._get_lr_scale()
        b1, b2 = self.betas
        self.set_hyper(lr, 1 - b1 ** self._adam_step, 1 - b2 ** self._adam_step)
        self.apply_update(zero_grad=zero_grad)
        self.last_lr = lr

    def set_hyper(self, lr, bc1, bc2):
        """[lr, 1 - b1^t, 1 - b2^t, 0] -> device (read by the Adam kernel), stream-ordered, without a host sync."""
        if self._hyper_host is None:
            self._hyper.copy_(torch.tensor([lr, bc1, bc2, 0.0]))
            return
        i = self._hyper_i
        self._hyper_i = (i + 1) % len(self._hyper_ev)
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()
        h = self._hyper_host[i]
        h[0], h[1], h[2], h[3] = lr, bc1, bc2, 0.0
        self._hyper.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[i] = ev

    def apply_update(self, zero_grad=False):
        """The capturable part: ||g||^2, then clip+Adam over the flat buffers (reads lr / bias corrections from device).
        The same pass refreshes the bf16 shadow parameters the forward GEMMs read and (optionally) clears the gradients."""
        b1, b2 = self.betas
        g = self.model.flat_gradients()
        self._nsq.zero_()
        ops.sumsq(g, self._nsq, self._nsq_ws)
        lowp = self.model._engine.lowp_buffer() if self.model._engine is not None else None
        ops.adam_step(self.model.flat_parameters(), g, self._m, self._v, self._nsq, self.grad_clip_thresh, self._hyper,
                      b1, b2, self.eps, self.weight_decay, p_lowp=lowp, zero_grad=zero_grad)
        self.model._invalidate(lowp_synced=lowp is not None)

    def zero_grad(self):
        g = self.model.flat_gradients()
        if g is not None:
            g.zero_()
            self.model.attach_grads()

    def grad_norm(self):
        return float(self._nsq.sqrt().item())

    # -- torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        self._ensure()
        params = list(self.model.parameters())
        offs = self.model._flat_offsets
        names = {id(p): n for n, p in self.model.named_parameters()}
        state = {}
        for i, p in enumerate(params):
            n = names[id(p)]
            if n in offs and self._adam_step > 0:
                state[i] = {"step": torch.tensor(float(self._adam_step)),
                            "exp_avg": self.model._view(self._m, offs[n], p.shape).clone(),
                            "exp_avg_sq": self.model._view(self._v, offs[n], p.shape).clone()}
        group = {"lr": getattr(self, "last_lr", 0.001), "betas": self.betas, "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def _load_now(self, sd):
        params = list(self.model.parameters())
        offs = self.model._flat_offsets
        names = {id(p): n for n, p in self.model.named_parameters()}
        for i, st in sd["state"].items():
            p = params[int(i)]
            n = names[id(p)]
            if n not in offs:
                continue
            self.model._view(self._m, offs[n], p.shape).copy_(st["exp_avg"])
            self.model._view(self._v, offs[n], p.shape).copy_(st["exp_avg_sq"])
            self._adam_step = int(float(st["step"]))

    def load_state_dict(self, sd):
        if self.model.flat_parameters() is None:
            self._pending_state = sd
        else:
            self._ensure()
            self._load_now(sd)
