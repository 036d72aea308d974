#!/usr/bin/env python
"""train.py — FastSpeech 2 training with the reference's CLI, YAML configs, log lines and checkpoint format
(reference train.py:21-198), one process per GPU:

    python train.py -p preprocess.yaml -m model.yaml -t train.yaml [--restore_step N]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -p ... -m ... -t ...

What differs from the reference is below the model call: forward/backward are HIP kernels behind libfs2hip.so, the
optimiser is one fused clip+Adam pass over flat buffers, nn.DataParallel (train.py:42) is replaced by RCCL gradient
all-reduce overlapped with backward (fastspeech2_amd/ddp.py), and the loader is a length-bucketed sampler + pinned-memory
prefetch thread (fastspeech2_amd/data.py) instead of a synchronous DataLoader with num_workers=0.
"""
import argparse
import os

import torch
import yaml

from evaluate import evaluate
import fastspeech2_amd
from fastspeech2_amd import ddp
from fastspeech2_amd.data import BucketedBatchSampler, Dataset, DevicePrefetcher, train_batches
from fastspeech2_amd.model import FastSpeech2Loss
from fastspeech2_amd.utils import RunLogger, get_model, get_param_num, get_vocoder, synth_one_sample


def save_checkpoint(model, optimizer, path):
    """{"model": state_dict, "optimizer": Adam state_dict} at {ckpt_path}/{step}.pth.tar (train.py:152-161).  Parameters
    are views into the engine's flat buffer: they are saved as compact contiguous CPU tensors in the reference layout."""
    sd = {k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}
    torch.save({"model": sd, "optimizer": optimizer.state_dict()}, path)


def main(args, configs):
    preprocess_config, model_config, train_config = configs
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", torch.cuda.current_device())
    # the step runs on a HIGH-priority stream: the engine's side stream (weight gradients) keeps normal priority, so the
    # dispatcher serves the critical forward / data-gradient chain first (bench A/B: -1.5..3 % step time)
    torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=-1))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=device)
    main_rank = rank == 0
    if main_rank:
        print("Prepare training ...")

    dataset = Dataset("train.txt", preprocess_config, train_config, sort=True, drop_last=True)
    batch_size = train_config["optimizer"]["batch_size"]
    # sort inside windows of `group_size` (x world) batches: 4 = the reference's own value (train.py:31), the default.  A wider
    # window leaves less padding per batch (LibriTTS-shaped pool, epoch-wide valid mel rows 0.76 / 0.89 / 0.92 at 4 / 16 / 64) but
    # buys only +3.5 % frames/s at 64: a B = 48 step of short utterances sits at the ~4.3 ms issue floor whatever its padding
    # (profiles/r06c_libritts_sweep.log) - and it changes which utterances share a batch, so it stays opt-in
    group_size = getattr(args, "group_size", 4)
    assert batch_size * group_size * world < len(dataset) or len(dataset) >= batch_size * world
    sampler = BucketedBatchSampler([dataset.length(i) for i in range(len(dataset))], batch_size, world, rank, group_size,
                                   shuffle=True, seed=train_config.get("seed", 1234))

    model, optimizer = get_model(args, configs, device, train=True, compute_dtype=args.dtype)
    model._ensure_flat(device)
    exchange = None
    if world > 1:
        exchange = ddp.GradExchange(model.flat_gradients(), world)
        torch.distributed.broadcast(model.flat_parameters(), 0)
        for _, buf in model.named_buffers():
            torch.distributed.broadcast(buf, 0)
        model._invalidate()
    Loss = FastSpeech2Loss(preprocess_config, model_config, count_reduce=ddp.CountExchange() if world > 1 else None).to(device)
    if main_rank:
        print("Number of FastSpeech2 Parameters:", get_param_num(model))

    vocoder = None
    if main_rank and not args.no_vocoder:
        try:
            vocoder = get_vocoder(model_config, device, hifigan_dir=args.hifigan_dir)
        except (FileNotFoundError, NotImplementedError) as e:
            print(f"[train] vocoder unavailable ({e}); audio samples are skipped")

    for p in train_config["path"].values():
        os.makedirs(p, exist_ok=True)
    train_log_path = os.path.join(train_config["path"]["log_path"], "train")
    val_log_path = os.path.join(train_config["path"]["log_path"], "val")
    train_logger = RunLogger(train_log_path) if main_rank else None
    val_logger = RunLogger(val_log_path) if main_rank else None

    step = args.restore_step + 1
    epoch = 1
    oc, sc = train_config["optimizer"], train_config["step"]
    grad_acc_step = oc["grad_acc_step"]
    total_step, log_step, save_step = sc["total_step"], sc["log_step"], sc["save_step"]
    synth_step, val_step = sc["synth_step"], sc["val_step"]
    sampling_rate = preprocess_config["preprocessing"]["audio"]["sampling_rate"]

    while True:
        sampler.set_epoch(epoch)
        for batch in DevicePrefetcher(train_batches(dataset, sampler), device):
            update = step % grad_acc_step == 0
            if exchange is not None:                    # all-reduce only the micro-step that completes the accumulation
                model._engine.grad_hook = exchange.ready if update else None
            if Loss.count_reduce is not None:           # valid-position counts: all-reduced while the forward pass runs
                Loss.count_reduce.start(batch[4], batch[7], int(batch[5]), min(int(batch[8]), model_config["max_seq_len"]))
            output = model(*(batch[2:]))
            losses = Loss(batch, output)
            total_loss = losses[0] / grad_acc_step
            total_loss.backward()
            if update:
                if exchange is not None:
                    exchange.finish()
                optimizer.step_and_update_lr(zero_grad=True)    # clip (grad_clip_thresh) + Adam + zero_grad, one pass

            if main_rank and step % log_step == 0:
                vals = [l.item() for l in losses]
                message1 = "Step {}/{}, ".format(step, total_step)
                message2 = ("Total Loss: {:.4f}, Mel Loss: {:.4f}, Mel PostNet Loss: {:.4f}, Pitch Loss: {:.4f}, "
                            "Energy Loss: {:.4f}, Duration Loss: {:.4f}").format(*vals)
                with open(os.path.join(train_log_path, "log.txt"), "a") as f:
                    f.write(message1 + message2 + "\n")
                print(message1 + message2)
                train_logger.log(step, losses=vals)

            if main_rank and vocoder is not None and step % synth_step == 0:
                _, wav_rec, wav_pred, tag = synth_one_sample(batch, output, vocoder, model_config, preprocess_config)
                train_logger.log(audio=wav_rec, sampling_rate=sampling_rate, tag="Training/step_{}_{}_reconstructed".format(step, tag))
                train_logger.log(audio=wav_pred, sampling_rate=sampling_rate, tag="Training/step_{}_{}_synthesized".format(step, tag))

            if step % val_step == 0:
                if main_rank:
                    model.eval()
                    message = evaluate(model, step, configs, val_logger, vocoder, device=device)
                    with open(os.path.join(val_log_path, "log.txt"), "a") as f:
                        f.write(message + "\n")
                    print(message)
                    model.train()
                if world > 1:
                    torch.distributed.barrier()

            if main_rank and step % save_step == 0:
                save_checkpoint(model, optimizer, os.path.join(train_config["path"]["ckpt_path"], "{}.pth.tar".format(step)))

            if step == total_step:
                if world > 1:
                    torch.distributed.barrier()
                return model, optimizer
            step += 1
        epoch += 1


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--restore_step", type=int, default=0)
    parser.add_argument("-p", "--preprocess_config", type=str, required=True, help="path to preprocess.yaml")
    parser.add_argument("-m", "--model_config", type=str, required=True, help="path to model.yaml")
    parser.add_argument("-t", "--train_config", type=str, required=True, help="path to train.yaml")
    parser.add_argument("--dtype", default=None, choices=[None, "fp32", "bf16"],
                        help="compute dtype of the HIP engine (default: FS2_DTYPE or fp32; master weights are always fp32)")
    parser.add_argument("--hifigan_dir", default="hifigan")
    parser.add_argument("--group_size", type=int, default=4, help="sorting window of the batch sampler in batches (reference train.py:31: 4)")
    parser.add_argument("--hw_queues", type=int, default=fastspeech2_amd.HW_QUEUES_DEFAULT,
                        help="HIP hardware queues of this process (GPU_MAX_HW_QUEUES; the runtime default 4 makes streams share queues: "
                             "utils.SynthPipeline / the engine's side streams); the same for every world size; an exported value wins; 0 = leave the runtime default")
    parser.add_argument("--no_vocoder", action="store_true", help="skip audio samples in the logs")
    return parser.parse_args(argv)


if __name__ == "__main__":
    args = parse_args()
    fastspeech2_amd.configure_hw_queues(args.hw_queues)       # before the first HIP call; identical for every world size
    configs = tuple(yaml.load(open(p, "r"), Loader=yaml.FullLoader)
                    for p in (args.preprocess_config, args.model_config, args.train_config))
    main(args, configs)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
