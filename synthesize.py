#!/usr/bin/env python
"""synthesize.py — batch / single-utterance synthesis with the reference's CLI (reference synthesize.py:86-214):

    python synthesize.py --restore_step N --mode batch --source val.txt -p ... -m ... -t ...
    python synthesize.py --restore_step N --mode single --text "{HH AH0 L OW1}" -p ... -m ... -t ...

The acoustic model and the HiFi-GAN generator run as HIP kernels; wavs land in train_config.path.result_path.
Batch mode reads phoneme strings (the `{...}` field of train.txt / val.txt), exactly what the reference's TextDataset
does.  Single mode takes the phoneme string directly: the grapheme-to-phoneme step of the reference (g2p_en / pypinyin +
lexicon, synthesize.py:20-84) is host-side string processing outside the hot path and its packages are not in this
image, so raw text without braces is rejected with a clear message instead of being silently mis-read.
With WORLD_SIZE > 1 (torch.distributed.run) the source list is sharded across GPUs: replicas only, no collective.
"""
import argparse
import os
import re

import numpy as np
import torch
import yaml

import fastspeech2_amd
from fastspeech2_amd.data import DevicePrefetcher, TextDataset
from fastspeech2_amd.text import text_to_sequence
from fastspeech2_amd.utils import SynthPipeline, get_model, get_vocoder


def synthesize(model, step, configs, vocoder, batchs, control_values, device=None):
    preprocess_config, model_config, train_config = configs
    pitch_control, energy_control, duration_control = control_values
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = 0
    # the reference's loop (synthesize.py:87-103) as a stream pipeline: the next batch's acoustic model under the previous batches' vocoders
    pipeline = SynthPipeline(model, vocoder, configs, control_values, device=device, path=train_config["path"]["result_path"],
                             write=True)
    for batch, _output, _wavs in pipeline(DevicePrefetcher(batchs, device)):
        n += len(batch[0])
    return n


def single_batch(args, preprocess_config):
    if not re.search(r"\{.+?\}", args.text):
        raise SystemExit("--mode single expects a phoneme string in braces, e.g. --text \"{HH AH0 L OW1 sp W ER1 L D}\" "
                         "(grapheme-to-phoneme conversion needs g2p_en/pypinyin, which this build does not ship)")
    texts = np.array([np.array(text_to_sequence(args.text, preprocess_config["preprocessing"]["text"]["text_cleaners"]))])
    text_lens = np.array([len(texts[0])])
    ids = raw_texts = [args.text[:100].replace("{", "").replace("}", "").replace(" ", "_")]
    return [(ids, raw_texts, np.array([args.speaker_id]), texts, text_lens, max(text_lens))]


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--restore_step", type=int, required=True)
    parser.add_argument("--mode", type=str, choices=["batch", "single"], required=True,
                        help="Synthesize a whole dataset or a single sentence")
    parser.add_argument("--source", type=str, default=None,
                        help="path to a source file with format like train.txt and val.txt, for batch mode only")
    parser.add_argument("--text", type=str, default=None, help="phoneme string in braces, for single-sentence mode only")
    parser.add_argument("--speaker_id", type=int, default=0, help="speaker ID for multi-speaker synthesis, single mode only")
    parser.add_argument("-p", "--preprocess_config", type=str, required=True, help="path to preprocess.yaml")
    parser.add_argument("-m", "--model_config", type=str, required=True, help="path to model.yaml")
    parser.add_argument("-t", "--train_config", type=str, required=True, help="path to train.yaml")
    parser.add_argument("--pitch_control", type=float, default=1.0)
    parser.add_argument("--energy_control", type=float, default=1.0)
    parser.add_argument("--duration_control", type=float, default=1.0)
    parser.add_argument("--batch_size", type=int, default=8, help="utterances per batch (reference: 8)")
    parser.add_argument("--dtype", default=None, choices=[None, "fp32", "bf16"])
    parser.add_argument("--vocoder_dtype", default="fp32", choices=["fp32", "bf16"])
    parser.add_argument("--hifigan_dir", default="hifigan", help="directory with config.json + generator_*.pth.tar")
    parser.add_argument("--hw_queues", type=int, default=fastspeech2_amd.HW_QUEUES_DEFAULT,
                        help="HIP hardware queues of this process (GPU_MAX_HW_QUEUES; the runtime default 4 makes streams share queues: "
                             "utils.SynthPipeline / the engine's side streams); the same for every world size; an exported value wins; 0 = leave the runtime default")
    parser.add_argument("--random_vocoder", action="store_true", help="allow a random-init vocoder when no checkpoint exists (smoke runs)")
    return parser.parse_args(argv)


def main(args):
    if args.mode == "batch":
        assert args.source is not None and args.text is None
    if args.mode == "single":
        assert args.source is None and args.text is not None
    configs = tuple(yaml.load(open(p, "r"), Loader=yaml.FullLoader)
                    for p in (args.preprocess_config, args.model_config, args.train_config))
    preprocess_config, model_config, train_config = configs
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", torch.cuda.current_device())
    model = get_model(args, configs, device, train=False, compute_dtype=args.dtype)
    vocoder = get_vocoder(model_config, device, hifigan_dir=args.hifigan_dir, compute_dtype=args.vocoder_dtype,
                          allow_random_init=args.random_vocoder)
    if args.mode == "batch":
        dataset = TextDataset(args.source, preprocess_config)
        mine = list(range(rank, len(dataset), world))          # replicas only: each GPU takes every world-th utterance
        batchs = (dataset.collate_fn([dataset[i] for i in mine[s:s + args.batch_size]])
                  for s in range(0, len(mine), args.batch_size))
    else:
        batchs = single_batch(args, preprocess_config)
    n = synthesize(model, args.restore_step, configs, vocoder, batchs,
                   (args.pitch_control, args.energy_control, args.duration_control), device=device)
    print(f"[rank {rank}] synthesized {n} utterances -> {train_config['path']['result_path']}")


if __name__ == "__main__":
    _args = parse_args()
    fastspeech2_amd.configure_hw_queues(_args.hw_queues)      # before the first HIP call; the pipeline keeps 4+ streams busy
    main(_args)
