"""Container-only measurement behind bench.py's cpu_baseline.kind = "port": the LIVE reference (ming024/FastSpeech2 imported from
/root/reference) and the oracle port (oracle/fs2_oracle.py) run the same train step - B = 4, L = 128, T ~ 900, 4 + 4 layers,
dropout on, forward + loss + backward + clip + Adam - on the same cores, alternating, 10 steps each at every thread count, first
2 discarded, median of 8.  The GPU box has no /root/reference, so this cannot run there; its log is committed under profiles/.

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_port_vs_reference.py > profiles/r04_cpu_port_vs_reference.log"""
import os
import sys
import time
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
import torch  # noqa: E402

for name in ("unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["unidecode"].unidecode = lambda x: x
sys.modules["inflect"].engine = lambda: None

from oracle import fs2_oracle as O  # noqa: E402
from oracle.weights import synthetic_batch  # noqa: E402
from tests.golden import configs  # noqa: E402


def main():
    cwd = os.getcwd()
    os.chdir(REF)                                           # the reference reads ./preprocessed_data/... relative to its root
    try:
        from model import FastSpeech2, FastSpeech2Loss
        pcfg, mcfg = configs.make(dec_layers=4, enc_layers=4, dropout=True)
        torch.manual_seed(1234)
        ref = FastSpeech2(pcfg, mcfg)
        ref.train()
        loss_fn = FastSpeech2Loss(pcfg, mcfg)
        opt_r = torch.optim.Adam(ref.parameters(), betas=(0.9, 0.98), eps=1e-9)
    finally:
        os.chdir(cwd)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and not any(s in k for s in ("position_enc", "_bins", "running_")):
            v.requires_grad_(True)
            params.append(v)
    opt_o = torch.optim.Adam(params, betas=(0.9, 0.98), eps=1e-9)
    b = synthetic_batch(1234, 4, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
    bn = {k: v for k, v in sd.items() if "running_" in k}
    batch = (None, None, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"], b["max_mel_len"],
             b["pitches"], b["energies"], b["durations"])

    def step_ref():
        t0 = time.perf_counter()
        out = ref(*batch[2:])
        loss = loss_fn(batch, out)[0]
        opt_r.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt_r.step()
        return time.perf_counter() - t0

    def step_port():
        t0 = time.perf_counter()
        out = O.fastspeech2_forward(sd, mcfg, pcfg, b["speakers"], b["texts"], b["src_lens"], b["max_src_len"], b["mels"], b["mel_lens"],
                                    b["max_mel_len"], b["pitches"], b["energies"], b["durations"], training=True, dropout=True, bn_buffers=bn)
        loss = O.fastspeech2_loss(pcfg, (b["mels"], b["pitches"], b["energies"], b["durations"]), out)[0]
        opt_o.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt_o.step()
        return time.perf_counter() - t0

    frames = int(b["mel_lens"].sum())
    print(f"host: {os.cpu_count()} logical CPUs; torch {torch.__version__}; batch B=4 L=128 T={int(b['max_mel_len'])} valid frames {frames}")
    print("| threads | reference s/step (median of 8) | port s/step | port / reference | reference frames/s | port frames/s |")
    print("|---|---|---|---|---|---|")
    for nt in [t for t in (4, 8, 16, 32) if t <= (os.cpu_count() or 1)]:
        torch.set_num_threads(nt)
        tr, tp = [], []
        for _ in range(10):                                  # alternating: both see the same machine state
            tr.append(step_ref())
            tp.append(step_port())
        tr, tp = sorted(tr[2:]), sorted(tp[2:])
        mr, mp = 0.5 * (tr[3] + tr[4]), 0.5 * (tp[3] + tp[4])
        print(f"| {nt} | {mr:.3f} | {mp:.3f} | {mp / mr:.2f} | {frames / mr:.0f} | {frames / mp:.0f} |", flush=True)


if __name__ == "__main__":
    main()
