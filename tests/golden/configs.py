"""Config dicts for parity tests: the reference's YAML structure (config/LJSpeech/{model,preprocess}.yaml) built
in code so tests do not depend on /root/reference being mounted."""
import copy
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")

from fastspeech2_amd.synthetic import LJ_STATS, MODEL, TRAIN, make_configs  # noqa: E402,F401
from fastspeech2_amd import synthetic as _syn

PREPROCESS = dict(_syn.PREPROCESS, path={"preprocessed_path": DATA})

HIFIGAN = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
           "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
           "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}


def ensure_data(n_speaker=4):
    _syn.ensure_data(DATA, n_speaker)


def make(dec_layers=4, multi_speaker=False, frame_level=False, dropout=True, enc_layers=4, max_seq_len=1000):
    return make_configs(dec_layers=dec_layers, multi_speaker=multi_speaker, frame_level=frame_level, dropout=dropout,
                        enc_layers=enc_layers, max_seq_len=max_seq_len, data_dir=DATA)
