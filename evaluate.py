#!/usr/bin/env python
"""evaluate.py — validation pass with the reference's CLI and message format (reference evaluate.py:18-119):

    python evaluate.py --restore_step N -p preprocess.yaml -m model.yaml -t train.yaml

`evaluate(model, step, configs, logger, vocoder)` is also what train.py calls every `val_step`: the forward runs under
no_grad through the HIP engine (eval mode: BatchNorm running statistics, no dropout), losses are batch-size-weighted
means over val.txt exactly as evaluate.py:37-51.
"""
import argparse

import torch
import yaml

import fastspeech2_amd
from fastspeech2_amd.data import Dataset, DevicePrefetcher
from fastspeech2_amd.model import FastSpeech2Loss
from fastspeech2_amd.utils import get_model, get_vocoder, synth_one_sample


def evaluate(model, step, configs, logger=None, vocoder=None, device=None):
    preprocess_config, model_config, train_config = configs
    device = device or torch.device("cuda", torch.cuda.current_device())
    dataset = Dataset("val.txt", preprocess_config, train_config, sort=False, drop_last=False)
    batch_size = train_config["optimizer"]["batch_size"]
    Loss = FastSpeech2Loss(preprocess_config, model_config).to(device)

    def batches():
        for s in range(0, len(dataset), batch_size):
            idx = list(range(s, min(len(dataset), s + batch_size)))
            data = [dataset[i] for i in idx]
            yield Dataset.reprocess(data, list(range(len(data))))

    loss_sums = [0.0] * 6
    batch = output = None
    for batch in DevicePrefetcher(batches(), device):
        with torch.no_grad():
            output = model(*(batch[2:]))
            losses = Loss(batch, output)
        for i in range(6):
            loss_sums[i] += losses[i].item() * len(batch[0])
    loss_means = [s / len(dataset) for s in loss_sums]
    message = ("Validation Step {}, Total Loss: {:.4f}, Mel Loss: {:.4f}, Mel PostNet Loss: {:.4f}, Pitch Loss: {:.4f}, "
               "Energy Loss: {:.4f}, Duration Loss: {:.4f}").format(*([step] + loss_means))
    if logger is not None:
        logger.log(step, losses=loss_means)
        if vocoder is not None and batch is not None:
            _, wav_rec, wav_pred, tag = synth_one_sample(batch, output, vocoder, model_config, preprocess_config)
            sr = preprocess_config["preprocessing"]["audio"]["sampling_rate"]
            logger.log(audio=wav_rec, sampling_rate=sr, tag="Validation/step_{}_{}_reconstructed".format(step, tag))
            logger.log(audio=wav_pred, sampling_rate=sr, tag="Validation/step_{}_{}_synthesized".format(step, tag))
    return message


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--restore_step", type=int, default=30000)
    parser.add_argument("-p", "--preprocess_config", type=str, required=True, help="path to preprocess.yaml")
    parser.add_argument("-m", "--model_config", type=str, required=True, help="path to model.yaml")
    parser.add_argument("-t", "--train_config", type=str, required=True, help="path to train.yaml")
    parser.add_argument("--dtype", default=None, choices=[None, "fp32", "bf16"], help="compute dtype of the HIP engine (default: FS2_DTYPE or fp32)")
    parser.add_argument("--hw_queues", type=int, default=fastspeech2_amd.HW_QUEUES_DEFAULT,
                        help="HIP hardware queues of this process (GPU_MAX_HW_QUEUES; the runtime default 4 makes streams share queues: "
                             "utils.SynthPipeline / the engine's side streams); the same for every world size; an exported value wins; 0 = leave the runtime default")
    return parser.parse_args(argv)


def load_configs(args):
    return tuple(yaml.load(open(p, "r"), Loader=yaml.FullLoader)
                 for p in (args.preprocess_config, args.model_config, args.train_config))


if __name__ == "__main__":
    args = parse_args()
    fastspeech2_amd.configure_hw_queues(args.hw_queues)       # before the first HIP call
    configs = load_configs(args)
    device = torch.device("cuda")
    model = get_model(args, configs, device, train=False, compute_dtype=args.dtype)
    print(evaluate(model, args.restore_step, configs, device=device))
