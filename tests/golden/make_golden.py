"""Generate the golden fixtures by running the LIVE reference (ming024/FastSpeech2 at /root/reference) on
seeded synthetic inputs.  Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes tests/golden/*.npz (small: inputs are regenerated from seeds by oracle/weights.py; only outputs,
losses and gradient summaries are stored).  Nothing is copied from the reference: it is imported and executed.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

# two pure-python packages the reference's text frontend imports but the hot path never uses
for name in ("unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["unidecode"].unidecode = lambda x: x
sys.modules["inflect"].engine = lambda: None

from tests.golden import configs  # noqa: E402
from oracle.weights import seeded_state_dict, synthetic_batch  # noqa: E402

torch.set_num_threads(8)


def grad_summary(named_grads):
    names = sorted(named_grads.keys())
    stats = np.zeros((len(names), 3), dtype=np.float64)
    for i, n in enumerate(names):
        g = named_grads[n].double()
        stats[i] = [g.sum().item(), g.abs().sum().item(), g.norm().item()]
    return names, stats


def build_reference(pcfg, mcfg):
    from model import FastSpeech2
    return FastSpeech2(pcfg, mcfg)


def case_train(tag, seed, B, L, batch_max_seq_len=None, **cfgkw):
    """train-mode forward+loss+backward with dropout neutralised (BatchNorm batch statistics live)."""
    import torch.nn.functional as F
    from model import FastSpeech2Loss
    pcfg, mcfg = configs.make(dropout=False, **cfgkw)
    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x       # PostNet hard-codes F.dropout(0.5)
    try:
        model = build_reference(pcfg, mcfg)
        sd = seeded_state_dict(model.state_dict(), seed)
        model.load_state_dict(sd)
        model.train()
        frame = cfgkw.get("frame_level", False)
        nspk = 4 if cfgkw.get("multi_speaker") else 1
        b = synthetic_batch(seed + 1, B, L, n_speaker=nspk, frame_level=frame,
                            max_seq_len=batch_max_seq_len or mcfg["max_seq_len"])
        out = model(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
                    b["pitches"], b["energies"], b["durations"])
        batch12 = (None, None, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                   b["max_mel_len"], b["pitches"], b["energies"], b["durations"])
        losses = FastSpeech2Loss(pcfg, mcfg)(batch12, out)
        losses[0].backward()
    finally:
        F.dropout = real_dropout
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    names, stats = grad_summary(grads)
    full = {}
    for n in ("mel_linear.bias", "decoder.layer_stack.0.slf_attn.layer_norm.weight",
              "variance_adaptor.duration_predictor.linear_layer.weight", "encoder.layer_stack.0.slf_attn.fc.bias",
              "postnet.convolutions.0.1.weight"):
        full["grad:" + n] = grads[n].numpy()
    bn = {k: v.numpy() for k, v in model.state_dict().items() if "running_" in k}
    np.savez_compressed(
        os.path.join(HERE, f"{tag}.npz"), seed=seed, B=B, L=L, torch_version=torch.__version__,
        mel=out[0].detach().numpy(), post=out[1].detach().numpy(), p_pred=out[2].detach().numpy(),
        e_pred=out[3].detach().numpy(), logd=out[4].detach().numpy(), mel_lens=out[9].numpy(),
        mel_masks=out[7].numpy(), losses=np.array([l.item() for l in losses], dtype=np.float64),
        grad_names=np.array(names), grad_stats=stats, state_keys=np.array(list(model.state_dict().keys())),
        **full, **{"bn:" + k: v for k, v in bn.items()})
    print(tag, "T=", out[0].shape[1], "losses", [round(l.item(), 5) for l in losses])


def case_eval(tag, seed, B, L, controls=(1.0, 1.0, 1.0), **cfgkw):
    """eval-mode free-running inference (predicted durations, controls)."""
    pcfg, mcfg = configs.make(**cfgkw)
    model = build_reference(pcfg, mcfg)
    sd = seeded_state_dict(model.state_dict(), seed)
    sd["variance_adaptor.duration_predictor.linear_layer.bias"] = torch.tensor([1.4])   # ~3 frames / phoneme
    model.load_state_dict(sd)
    model.eval()
    nspk = 4 if cfgkw.get("multi_speaker") else 1
    b = synthetic_batch(seed + 1, B, L, n_speaker=nspk)
    with torch.no_grad():
        out = model(b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], p_control=controls[0],
                    e_control=controls[1], d_control=controls[2])
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), seed=seed, B=B, L=L, controls=np.array(controls),
                        torch_version=torch.__version__, mel=out[0].numpy(), post=out[1].numpy(), p_pred=out[2].numpy(),
                        e_pred=out[3].numpy(), logd=out[4].numpy(), d_rounded=out[5].numpy(), mel_lens=out[9].numpy(),
                        mel_masks=out[7].numpy())
    print(tag, "T=", out[0].shape[1], "mel_lens", out[9].tolist())


def case_length_regulator():
    from model.modules import LengthRegulator
    import model.modules as mm
    mm.device = torch.device("cpu")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 11, 8, generator=g)
    d_float = torch.tensor([[2.0, 1.6, -0.4, 0.0, 3.2, 0.999, 1.0, 2.5, 0.0, 4.0, 1.0],
                            [0.0, 0.0, 5.0, 1.2, 2.0, 2.0, 0.8, 0.0, 0.0, 0.0, 0.0],
                            [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 9.7]])
    lr = LengthRegulator()
    res = {}
    for tag, max_len in (("none", None), ("crop", 10), ("pad", 30)):
        out, mel_len = lr(x, d_float, max_len)
        res[f"out_{tag}"] = out.numpy()
        res[f"len_{tag}"] = mel_len.numpy()
    np.savez_compressed(os.path.join(HERE, "length_regulator.npz"), x=x.numpy(), d=d_float.numpy(), **res)
    print("length_regulator ok")


GAIN = float(os.environ.get('HIFI_GAIN', '1.2'))


def case_hifigan(seed=11):
    import json
    import hifigan
    with open(os.path.join(REF, "hifigan", "config.json")) as f:
        h = hifigan.AttrDict(json.load(f))
    gen = hifigan.Generator(h)
    g = torch.Generator().manual_seed(seed)
    sd = gen.state_dict()
    new = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if k.endswith("weight_v"):
            fan_in = v[0].numel() if "ups" not in k else v.shape[0] * v.shape[2]
            new[k] = torch.randn(v.shape, generator=g) * (GAIN / fan_in ** 0.5)
        elif k.endswith("weight_g"):
            new[k] = None
        else:
            new[k] = torch.randn(v.shape, generator=g) * 0.05
    for k in list(new.keys()):
        if k.endswith("weight_g"):
            v = new[k[:-1] + "v"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).view(sd[k].shape)
            new[k] = norm * (0.8 + 0.4 * torch.rand(sd[k].shape, generator=g))
    gen.load_state_dict(new)
    gen.eval()
    gen.remove_weight_norm()
    mel = torch.clamp(torch.randn(2, 80, 24, generator=g) * 2 - 5, -11.5, 2.0)
    with torch.no_grad():
        wav = gen(mel)
    np.savez_compressed(os.path.join(HERE, "hifigan.npz"), seed=seed, mel=mel.numpy(), wav=wav.numpy(),
                        keys=np.array(sorted(sd.keys())), rms=float(wav.pow(2).mean().sqrt()))
    print("hifigan wav rms", float(wav.pow(2).mean().sqrt()), wav.shape)


def case_stft():
    """The reference's own audio/stft.py TacotronSTFT on CPU.  librosa (absent here) is stubbed: pad_center / tiny are
    trivial; filters.mel serves the ORACLE's restated Slaney filterbank (so the filterbank itself stays "parity
    unpinned"; everything else — DFT basis, window, reflect pad, framing, magnitude, mel matmul, log-clamp, energy —
    is the reference's code).  `.cuda()` is neutralised (stft.py:68-69 call it unconditionally)."""
    from oracle import fs2_oracle as O
    lib = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    filt = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1):
        n = data.shape[axis]
        lpad = (size - n) // 2
        return np.pad(data, (lpad, size - n - lpad))
    util.pad_center = pad_center
    util.tiny = lambda x: np.finfo(np.float32).tiny
    util.normalize = lambda x, norm=None: x
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lib.util, lib.filters = util, filt
    sys.modules.update({"librosa": lib, "librosa.util": util, "librosa.filters": filt})
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        from audio.stft import TacotronSTFT
        from scipy.io import wavfile
        stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000)
        sr, wav = wavfile.read(os.path.join(REF, "demo", "LJSpeech", "LJ001-0012_ground-truth.wav"))
        assert sr == 22050
        real = (wav[20000:20000 + 7 * 256 + 100].astype(np.float32) / 32768.0)
        g = torch.Generator().manual_seed(21)
        n = real.shape[0]
        tt = torch.arange(n, dtype=torch.float32) / 22050.0
        synth = 0.4 * torch.sin(2 * np.pi * 440.0 * tt) + 0.2 * torch.sin(2 * np.pi * 3000.0 * tt) + 0.05 * torch.randn(n, generator=g)
        y = torch.stack([torch.from_numpy(real), torch.clamp(synth, -1, 1)])
        mel, energy = stft.mel_spectrogram(y)
        mag, _ = stft.stft_fn.transform(y)
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(HERE, "stft.npz"), y=y.numpy(), mel=mel.numpy(), energy=energy.numpy(),
                        mag_sample=mag[:, ::37, :].numpy(), forward_basis_rows=stft.stft_fn.forward_basis[::101, 0, :].numpy())
    print("stft mel", tuple(mel.shape), float(mel.min()), float(mel.max()), "energy max", float(energy.max()))


if __name__ == "__main__":
    os.chdir(REF)
    case_length_regulator()
    case_train("train_lj", 1234, B=3, L=24)
    case_train("train_multi_frame", 77, B=2, L=16, multi_speaker=True, frame_level=True, dec_layers=2, enc_layers=2)
    case_train("train_trunc", 99, B=2, L=20, dec_layers=1, enc_layers=1, max_seq_len=64, batch_max_seq_len=1000)
    case_eval("eval_lj", 4321, B=3, L=20, controls=(1.2, 0.9, 1.1))
    case_eval("eval_multi", 55, B=2, L=12, controls=(1.0, 1.0, 0.8), multi_speaker=True, dec_layers=2, enc_layers=2)
    case_hifigan()
    case_stft()
