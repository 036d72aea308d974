"""`python preprocess.py config/LJSpeech/preprocess.yaml` — the reference's preprocess.py:1-16 over the GPU corpus pipeline
(fastspeech2_amd/preprocess.py).  `--pack` additionally writes the packed feature shard the training data pipeline maps
(fastspeech2_amd/data.pack_features)."""
import argparse

import yaml

from fastspeech2_amd.preprocess import Preprocessor

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("config", type=str, help="path to preprocess.yaml")
    parser.add_argument("--batch_seconds", type=float, default=1800.0, help="audio per ragged STFT batch on the GPU")
    parser.add_argument("--num_workers", type=int, default=8, help="host threads for TextGrid / wav / F0")
    parser.add_argument("--seed", type=int, default=None, help="seed of the train/val shuffle (reference: unseeded)")
    parser.add_argument("--pack", action="store_true", help="also write the packed feature shards for train.txt / val.txt")
    args = parser.parse_args()

    config = yaml.load(open(args.config, "r"), Loader=yaml.FullLoader)
    Preprocessor(config, batch_seconds=args.batch_seconds, num_workers=args.num_workers, seed=args.seed).build_from_path()
    if args.pack:
        from fastspeech2_amd.data import pack_features
        for split in ("train.txt", "val.txt"):
            pack_features(config["path"]["preprocessed_path"], split)
