"""Factories and batch plumbing with the reference's names and conventions (utils/model.py, utils/tools.py).

get_model / get_vocoder / vocoder_infer / to_device / get_mask_from_lengths / pad_1D / pad_2D are what train.py,
evaluate.py and synthesize.py call around the hot path; only the device work differs (HIP kernels behind the C ABI,
PCM conversion fused into the vocoder's last kernel).
"""
import json
import os

import numpy as np
import torch

from . import hifigan
from .model import FastSpeech2, ScheduledOptim

# hifigan/config.json of the reference (HiFi-GAN V1) — used when no config file sits next to the checkpoint
HIFIGAN_V1 = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
              "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
              "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 80, "sampling_rate": 22050,
              "hop_size": 256, "win_size": 1024, "n_fft": 1024, "fmin": 0, "fmax": 8000}


def get_model(args, configs, device, train=False, compute_dtype=None):
    """utils/model.py:11-34."""
    preprocess_config, model_config, train_config = configs
    model = FastSpeech2(preprocess_config, model_config, compute_dtype=compute_dtype).to(device)
    ckpt = None
    if args.restore_step:
        ckpt_path = os.path.join(train_config["path"]["ckpt_path"], "{}.pth.tar".format(args.restore_step))
        # (checkpoints written by the reference hold a numpy scalar learning rate: see load_checkpoint)
        ckpt = load_checkpoint(ckpt_path)
        model.load_state_dict(ckpt["model"])
        model._dropout_step = int(args.restore_step)      # dropout masks continue their sequence instead of replaying it
    if train:
        scheduled_optim = ScheduledOptim(model, train_config, model_config, args.restore_step)
        if args.restore_step:
            scheduled_optim.load_state_dict(ckpt["optimizer"])
        model.train()
        return model, scheduled_optim
    model.eval()
    model.requires_grad_ = False                 # attribute assignment in the reference too (utils/model.py:33)
    return model


def load_checkpoint(path):
    """torch.load with the SAFE (weights_only) unpickler.  Checkpoints written by the reference hold one non-tensor object, a
    numpy scalar (param_groups[0]["lr"]: init_lr is np.power(...), model/optimizer.py:19,50), which torch >= 2.6's default
    unpickler rejects; exactly the numpy scalar reconstruction globals are allow-listed for the call instead of switching the
    safe unpickler off."""
    import numpy as _np
    allow = [_np.dtype, _np.ndarray]
    try:
        from numpy._core import multiarray as _ma      # numpy >= 2 (pickles written by numpy 1.x name numpy.core.multiarray,
    except ImportError:                                 # which numpy 2 resolves to the same objects)
        from numpy.core import multiarray as _ma
    allow += [_ma.scalar, _ma._reconstruct]
    # torch matches allow-listed globals by "module.qualname" STRING: under numpy 2 the two objects above are named
    # numpy._core.multiarray.*, while a checkpoint pickled with numpy 1.x - every checkpoint the reference's authors published -
    # names numpy.core.multiarray.*.  The tuple form registers the same objects under the legacy names too.
    allow += [(_ma.scalar, "numpy.core.multiarray.scalar"), (_ma._reconstruct, "numpy.core.multiarray._reconstruct")]
    allow += [type(_np.dtype(t)) for t in ("float64", "float32", "int64", "int32")]
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def get_param_num(model):
    return sum(param.numel() for param in model.parameters())


def get_vocoder(config, device, hifigan_dir="hifigan", compute_dtype="fp32", allow_random_init=False):
    """utils/model.py:42-71 (HiFi-GAN branch; MelGAN is a torch.hub download and out of scope)."""
    name = config["vocoder"]["model"]
    speaker = config["vocoder"]["speaker"]
    if name != "HiFi-GAN":
        raise NotImplementedError(f"vocoder {name!r}: only HiFi-GAN is built (MelGAN needs torch.hub / network)")
    cfg_path = os.path.join(hifigan_dir, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path, "r") as f:
            h = json.load(f)
    else:
        h = dict(HIFIGAN_V1)
    vocoder = hifigan.Generator(hifigan.AttrDict(h), compute_dtype=compute_dtype)
    ckpt_path = os.path.join(hifigan_dir, {"LJSpeech": "generator_LJSpeech.pth.tar",
                                           "universal": "generator_universal.pth.tar"}[speaker])
    if os.path.exists(ckpt_path):
        # a third-party download holding plain tensors: the safe unpickler, always
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        vocoder.load_state_dict(ckpt["generator"])
    elif not allow_random_init:
        raise FileNotFoundError(ckpt_path)
    vocoder.eval()
    vocoder.remove_weight_norm()
    vocoder.to(device)
    return vocoder


def vocoder_infer(mels, vocoder, model_config, preprocess_config, lengths=None):
    """utils/model.py:74-92: mels (B, 80, T) -> list of int16 numpy arrays (cut to `lengths` samples).
    (The PCM comes back with a plain `.cpu()`: a cached pinned buffer + asynchronous copy + reading `lengths` behind the vocoder's
    launches was measured twice - r03s: 8.77 vs 8.66 ms per batch, r05q: 7.63 vs 7.59 - and is no faster.)"""
    name = model_config["vocoder"]["model"]
    assert name == "HiFi-GAN"
    if torch.is_tensor(lengths):
        lengths = lengths.tolist()
    with torch.no_grad():
        pcm = vocoder.infer_pcm(mels, preprocess_config["preprocessing"]["audio"]["max_wav_value"])
    wavs = [w for w in pcm.cpu().numpy()]
    for i in range(len(mels)):
        if lengths is not None:
            wavs[i] = wavs[i][: lengths[i]]
    return wavs


def to_device(data, device):
    """utils/tools.py:18-66: numpy batch tuple (12 = train, 6 = synth) -> device tensors; non_blocking from pinned
    staging would be the data-pipeline row (§8(f) #1)."""
    if len(data) == 12:
        ids, raw_texts, speakers, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, pitches, energies, durations = data
        return (ids, raw_texts,
                torch.from_numpy(speakers).long().to(device), torch.from_numpy(texts).long().to(device),
                lens_to_device(src_lens, device), max_src_len,
                torch.from_numpy(mels).float().to(device), lens_to_device(mel_lens, device), max_mel_len,
                torch.from_numpy(pitches).float().to(device), torch.from_numpy(energies).to(device),
                torch.from_numpy(durations).long().to(device))
    if len(data) == 6:
        ids, raw_texts, speakers, texts, src_lens, max_src_len = data
        return (ids, raw_texts, torch.from_numpy(speakers).long().to(device), torch.from_numpy(texts).long().to(device),
                lens_to_device(src_lens, device), max_src_len)
    raise ValueError(f"to_device: batch of length {len(data)} (expected 12 or 6)")


def lens_to_device(lens, device):
    """a lengths vector (numpy or CPU tensor) -> device tensor that also carries its HOST copy (`_fs2_host`, numpy): the engine
    decides from it, without a device round trip, whether a batch is padded enough for the contractions to skip tiles
    (Engine.lens_skip_min)."""
    host = lens.numpy() if torch.is_tensor(lens) else np.asarray(lens)
    t = torch.from_numpy(np.ascontiguousarray(host)).to(device)
    t._fs2_host = host
    return t


def get_mask_from_lengths(lengths, max_len=None):
    """utils/tools.py:91-99 — True = padding."""
    batch_size = lengths.shape[0]
    if max_len is None:
        max_len = torch.max(lengths).item()
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0).expand(batch_size, -1)
    return ids >= lengths.unsqueeze(1).expand(-1, max_len)


def pad_1D(inputs, PAD=0):
    """utils/tools.py:265-275."""
    max_len = max(len(x) for x in inputs)
    return np.stack([np.pad(x, (0, max_len - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def pad_2D(inputs, maxlen=None):
    """utils/tools.py:278-296."""
    def pad(x, max_len):
        if np.shape(x)[0] > max_len:
            raise ValueError("not max_len")
        return np.pad(x, ((0, max_len - np.shape(x)[0]), (0, 0)), mode="constant", constant_values=0)

    max_len = maxlen if maxlen else max(np.shape(x)[0] for x in inputs)
    return np.stack([pad(x, max_len) for x in inputs])


def synth_samples(targets, predictions, vocoder, model_config, preprocess_config, path, write=True):
    """utils/tools.py:164-198 without the matplotlib figures: vocode the post-net mels of a synth batch and write
    `{path}/{basename}.wav`.  Returns the list of int16 arrays."""
    basenames = targets[0]
    mel_predictions = predictions[1].transpose(1, 2)                     # (B, 80, T) view — zero-copy into the vocoder
    lengths = _sample_lengths(predictions[9], preprocess_config)
    wavs = vocoder_infer(mel_predictions, vocoder, model_config, preprocess_config, lengths=lengths)
    if write:
        _write_wavs(path, basenames, wavs, preprocess_config)
    return wavs


def _sample_lengths(mel_lens, preprocess_config):
    """frames -> samples per utterance (utils/tools.py:176); from the host copy the engine's own length round trip left on the
    tensor when there is one (no second synchronisation), else from the tensor."""
    hop = preprocess_config["preprocessing"]["stft"]["hop_length"]
    host = getattr(mel_lens, "_fs2_host", None)
    return [int(n) * hop for n in host] if host is not None else (mel_lens * hop).tolist()


def _write_wavs(path, basenames, wavs, preprocess_config):
    from scipy.io import wavfile
    os.makedirs(path, exist_ok=True)
    sr = preprocess_config["preprocessing"]["audio"]["sampling_rate"]
    for wav, basename in zip(wavs, basenames):
        wavfile.write(os.path.join(path, "{}.wav".format(basename)), sr, wav)


class SynthPipeline:
    """Batch synthesis (synthesize.py:87-103: model -> synth_samples, one batch after the other) as a two-stage pipeline on HIP
    streams.  The acoustic model of batch i+1 - ~150 short launches around a host round trip for the output length - is issued on
    a high-priority stream while HiFi-GAN still works on batch i on another; the PCM of batch i comes back through pinned memory
    behind an event, and the host only waits for it AFTER the next batches' launches are queued, so the device never drains between
    batches.  Consecutive batches' vocoders go to `voc_streams` alternating streams: every HiFi-GAN launch ends in a partial round
    of tiles, and the other batch's launch fills it (same box, ms per val.txt-shaped batch of 8: sequential 7.05, one vocoder
    stream 5.88 - 5.96, two 5.16, three 5.07: profiles/r05w_*, r05x_*).  Per batch the launches, their order and their operands are
    those of the sequential loop: the samples are bit-identical (tests/test_vocoder_stft_gpu.py::test_synth_pipeline_*).

    for batch, output, wavs in SynthPipeline(model, vocoder, configs)(batches): ...   # in batch order, `voc_streams` batches behind"""

    def __init__(self, model, vocoder, configs, control_values=(1.0, 1.0, 1.0), device=None, path=None, write=False, voc_streams=3):
        self.model, self.vocoder = model, vocoder
        self.preprocess_config, self.model_config = configs[0], configs[1]
        assert self.model_config["vocoder"]["model"] == "HiFi-GAN"
        self.controls = control_values
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:          # "cuda" -> "cuda:N": the vocoder's weight images are cached per RESOLVED device, and a
            self.device = torch.device("cuda", torch.cuda.current_device())   # miss would re-pack them on a vocoder stream mid-pipeline
        self.path, self.write = path, write
        self.s_ac = torch.cuda.Stream(device=self.device, priority=-1)
        # voc_streams > 1: consecutive batches' vocoders on alternating streams (the tail of one launch filled by the other batch's)
        self.s_vocs = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(voc_streams)))]
        self._n = 0
        # pinned PCM staging: a RING of host buffers (one more than the batches in flight), each grown to the largest batch it has
        # carried - rounds 1-5 asked the pinned allocator for a fresh tensor per batch.  A slot is reused only after the batch that
        # last used it has been handed to the caller (its `done` event waited for and its samples copied out by _finish)
        self._pcm_ring = [None] * (len(self.s_vocs) + 2)
        # the vocoder's packed weight images: made once, here, and every stream of the pipeline is ordered behind that (the first
        # forward would otherwise pack them on ITS stream while the next batch's forward reads them on another)
        vocoder.prepare(self.device)
        packed = torch.cuda.Event()
        packed.record(torch.cuda.current_stream(self.device))
        for st in [self.s_ac] + self.s_vocs:
            st.wait_event(packed)

    def _launch(self, batch):
        p, e, d = self.controls
        self.s_ac.wait_stream(torch.cuda.current_stream(self.device))     # the batch's host-to-device copies
        with torch.no_grad():
            with torch.cuda.stream(self.s_ac):
                out = self.model(*(batch[2:]), p_control=p, e_control=e, d_control=d)
                lengths = _sample_lengths(out[9], self.preprocess_config)
                ready = torch.cuda.Event()
                ready.record(self.s_ac)
            s_voc = self.s_vocs[self._n % len(self.s_vocs)]
            self._n += 1
            with torch.cuda.stream(s_voc):
                s_voc.wait_event(ready)
                pcm = self.vocoder.infer_pcm(out[1].transpose(1, 2), self.preprocess_config["preprocessing"]["audio"]["max_wav_value"])
                slot = (self._n - 1) % len(self._pcm_ring)
                buf = self._pcm_ring[slot]
                if buf is None or buf.numel() < pcm.numel():
                    buf = self._pcm_ring[slot] = torch.empty(max(pcm.numel(), 1 << 20), dtype=pcm.dtype, pin_memory=True)
                host = buf[:pcm.numel()].view(pcm.shape)
                host.copy_(pcm, non_blocking=True)
                done = torch.cuda.Event()
                done.record(s_voc)
        # `out` and `pcm` were allocated on one stream and are read on another: they stay referenced until `done` has been waited for
        return batch, out, pcm, host, lengths, done

    def _finish(self, item):
        batch, out, pcm, host, lengths, done = item
        done.synchronize()
        wavs = [w[:n].copy() for w, n in zip(host.numpy(), lengths)]        # (out of the ring slot: it is reused len(ring) batches later)
        if self.write:
            _write_wavs(self.path, batch[0], wavs, self.preprocess_config)
        return batch, out, wavs

    def __call__(self, batches):
        flight = []
        clean = False
        try:
            for batch in batches:
                flight.append(self._launch(batch))
                while len(flight) > len(self.s_vocs):
                    yield self._finish(flight.pop(0))
            while flight:
                yield self._finish(flight.pop(0))
            clean = True
        finally:
            # the consumer stopped early, or a launch raised - possibly half way, with the acoustic model and part of the vocoder of a
            # batch that never reached `flight` already queued: tensors of one stream's pool are being read on another, and nothing
            # may go back to a pool before that work has finished.  Every stream of the pipeline is drained, unconditionally.
            if not clean:
                for item in flight:
                    item[-1].synchronize()
                self.s_ac.synchronize()
                for st in self.s_vocs:
                    st.synchronize()


def synth_one_sample(targets, predictions, vocoder, model_config, preprocess_config):
    """utils/tools.py:107-162 without the matplotlib figure: vocode the first utterance of a train batch from its
    ground-truth mel and from the predicted post-net mel.  Returns (None, wav_reconstruction, wav_prediction, basename)."""
    basename = targets[0][0]
    mel_len = int(predictions[9][0].item())
    mel_target = targets[6][0, :mel_len].detach().float().transpose(0, 1)
    mel_prediction = predictions[1][0, :mel_len].detach().float().transpose(0, 1)
    if vocoder is None:
        return None, None, None, basename
    wav_reconstruction = vocoder_infer(mel_target.unsqueeze(0).contiguous(), vocoder, model_config, preprocess_config)[0]
    wav_prediction = vocoder_infer(mel_prediction.unsqueeze(0).contiguous(), vocoder, model_config, preprocess_config)[0]
    return None, wav_reconstruction, wav_prediction, basename


class RunLogger:
    """What train.py / evaluate.py need from utils/tools.py:68-88's `log`: scalar losses per step (+ optional audio).
    TensorBoard is used when importable (it is absent from this image); a `log.jsonl` next to the reference's
    `log.txt` is always written so runs can be inspected without it."""

    LOSS_NAMES = ("total_loss", "mel_loss", "mel_postnet_loss", "pitch_loss", "energy_loss", "duration_loss")

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.path = path
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(path)
        except Exception:
            self.tb = None

    def log(self, step=None, losses=None, audio=None, sampling_rate=22050, tag=""):
        if losses is not None:
            with open(os.path.join(self.path, "log.jsonl"), "a") as f:
                f.write(json.dumps({"step": step, **{n: float(v) for n, v in zip(self.LOSS_NAMES, losses)}}) + "\n")
            if self.tb is not None:
                for n, v in zip(self.LOSS_NAMES, losses):
                    self.tb.add_scalar("Loss/" + n, float(v), step)
        if audio is not None:
            from scipy.io import wavfile
            wavfile.write(os.path.join(self.path, tag.replace("/", "_") + ".wav"), sampling_rate, audio)
            if self.tb is not None:
                self.tb.add_audio(tag, audio / max(abs(audio).max(), 1), sample_rate=sampling_rate)
