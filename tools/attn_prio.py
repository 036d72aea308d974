"""Dev tool: forward attention with / without s_setprio around its MFMA runs (dev library switch fs2_dev_attn_prio)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FS2_LIB_PATH", os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so"))
import torch
from fastspeech2_amd import ops
from fastspeech2_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
B, S, H = 48, int(b["max_mel_len"]), 2
lens = torch.as_tensor(b["mel_lens"]).to(torch.int32).to(dev)
qkv = (torch.randn(B * S, 3 * H * 128, device=dev) * 0.5).to(torch.bfloat16)
lib = ctypes.CDLL(os.environ["FS2_LIB_PATH"])
def t():
    for _ in range(3): ops.attn_fwd(qkv, lens, B, S, H)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.attn_fwd(qkv, lens, B, S, H)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[2] * 1e3
for rnd in range(3):
    for on in (0, 1):
        assert lib.fs2_dev_attn_prio(on) == 0
        print(f"round {rnd} setprio={on}: fwd {t():.1f} us", flush=True)
