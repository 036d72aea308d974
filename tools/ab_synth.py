"""Dev tool: same-box A/B of the batch-synthesis pass: the shipped vocoder_infer (pinned, asynchronous PCM copy; lengths read
after the vocoder is queued) against the round-2 form (`lengths.tolist()` before the vocoder, `pcm.cpu()` into pageable memory)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastspeech2_amd import utils

def old_vocoder_infer(mels, vocoder, model_config, preprocess_config, lengths=None):
    if torch.is_tensor(lengths):
        lengths = lengths.tolist()
    with torch.no_grad():
        pcm = vocoder.infer_pcm(mels, preprocess_config["preprocessing"]["audio"]["max_wav_value"])
    wavs = [w for w in pcm.cpu().numpy()]
    for i in range(len(mels)):
        if lengths is not None:
            wavs[i] = wavs[i][: lengths[i]]
    return wavs

new = utils.vocoder_infer
args = bench.parse(["--mode", "synth", "--no-roofline", "--no-cpu-baseline"])
dev = bench.init_rank(1, 0)
for rnd in range(2):
    for name, f in (("pinned/async", new), ("pageable/sync-first", old_vocoder_infer)):
        utils.vocoder_infer = f
        r = bench.synth_measure(args, dev, 0, 1, args.steps, args.warmup, False)
        print(f"{name:22s} {r['dt'] / args.steps * 1e3:.3f} ms/step  RTF {r['dt'] / r['audio_s']:.3e}", flush=True)
