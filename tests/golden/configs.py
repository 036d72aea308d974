"""Config dicts for parity tests: the reference's YAML structure (config/LJSpeech/{model,preprocess}.yaml) built
in code so tests do not depend on /root/reference being mounted."""
import copy
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")

# reference preprocessed_data/LJSpeech/stats.json (pitch/energy [min, max, mean, std])
LJ_STATS = {"pitch": [-2.917079304729967, 11.391254536985784, 207.6309860026605, 46.77559025098988],
            "energy": [-1.431044578552246, 8.184337615966797, 37.32621679053821, 26.044180782835863]}

MODEL = {
    "transformer": {"encoder_layer": 4, "encoder_head": 2, "encoder_hidden": 256, "decoder_layer": 4, "decoder_head": 2,
                    "decoder_hidden": 256, "conv_filter_size": 1024, "conv_kernel_size": [9, 1], "encoder_dropout": 0.2,
                    "decoder_dropout": 0.2},
    "variance_predictor": {"filter_size": 256, "kernel_size": 3, "dropout": 0.5},
    "variance_embedding": {"pitch_quantization": "linear", "energy_quantization": "linear", "n_bins": 256},
    "multi_speaker": False,
    "max_seq_len": 1000,
    "vocoder": {"model": "HiFi-GAN", "speaker": "LJSpeech"},
}

PREPROCESS = {
    "dataset": "LJSpeech",
    "path": {"preprocessed_path": DATA},
    "preprocessing": {
        "val_size": 512,
        "text": {"text_cleaners": ["english_cleaners"], "language": "en"},
        "audio": {"sampling_rate": 22050, "max_wav_value": 32768.0},
        "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
        "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": 8000},
        "pitch": {"feature": "phoneme_level", "normalization": True},
        "energy": {"feature": "phoneme_level", "normalization": True},
    },
}

TRAIN = {
    "path": {"ckpt_path": "./output/ckpt/LJSpeech", "log_path": "./output/log/LJSpeech", "result_path": "./output/result/LJSpeech"},
    "optimizer": {"batch_size": 48, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "grad_clip_thresh": 1.0,
                  "grad_acc_step": 1, "warm_up_step": 4000, "anneal_steps": [300000, 400000, 500000], "anneal_rate": 0.3},
    "step": {"total_step": 900000, "log_step": 100, "synth_step": 1000, "val_step": 1000, "save_step": 100000},
}

HIFIGAN = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
           "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
           "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}


def ensure_data(n_speaker=4):
    os.makedirs(DATA, exist_ok=True)
    p = os.path.join(DATA, "stats.json")
    if not os.path.exists(p):
        with open(p, "w") as f:
            json.dump(LJ_STATS, f)
    p = os.path.join(DATA, "speakers.json")
    if not os.path.exists(p):
        with open(p, "w") as f:
            json.dump({f"spk{i}": i for i in range(n_speaker)}, f)


def make(dec_layers=4, multi_speaker=False, frame_level=False, dropout=True, enc_layers=4, max_seq_len=1000):
    ensure_data()
    m = copy.deepcopy(MODEL)
    p = copy.deepcopy(PREPROCESS)
    m["transformer"]["decoder_layer"] = dec_layers
    m["transformer"]["encoder_layer"] = enc_layers
    m["multi_speaker"] = multi_speaker
    m["max_seq_len"] = max_seq_len
    if frame_level:
        p["preprocessing"]["pitch"]["feature"] = "frame_level"
        p["preprocessing"]["energy"]["feature"] = "frame_level"
    if not dropout:
        m["transformer"]["encoder_dropout"] = 0.0
        m["transformer"]["decoder_dropout"] = 0.0
        m["variance_predictor"]["dropout"] = 0.0
    return p, m
