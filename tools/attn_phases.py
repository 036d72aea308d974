"""Dev tool (dev library): phase timeline of the pipelined dK/dV attention kernel - s_memtime stamps of workgroup 0, wave 0 at its
phase boundaries (fs2_attn.hip FS2_STAMP), printed as cycles per phase for the first tiles.  s_memtime ticks at 100 MHz on gfx950?
(the tool prints raw tick differences and the tile total; compare ratios)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FS2_LIB_PATH", os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so"))
import torch
from fastspeech2_amd import ops, _lib
from fastspeech2_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
b = synthetic_batch(1234, 48, 128, dur_lo=4, dur_hi=10, min_len_frac=0.75)
B, S, H = 48, int(b["max_mel_len"]), 2
lens = torch.as_tensor(b["mel_lens"]).to(torch.int32).to(dev)
qkv = (torch.randn(B * S, 3 * H * 128, device=dev) * 0.5).to(torch.bfloat16)
dctx = torch.randn(B * S, H * 128, device=dev).to(torch.bfloat16)
ctx, lse = ops.attn_fwd(qkv, lens, B, S, H)
lib = ctypes.CDLL(os.environ["FS2_LIB_PATH"])
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2          # 0 = forward, 2 = dK/dV
assert lib.fs2_dev_attn_stamp_select(which) == 0
for _ in range(3):
    ops.attn_fwd(qkv, lens, B, S, H)
    ops.attn_bwd(qkv, ctx, dctx, lse, lens, B, S, H)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16 * 12))()
assert lib.fs2_dev_attn_stamps(buf) == 0
if which == 1:
    print(f"dQ kernel: prologue (Q / dO / O rows, delta) {buf[1] - buf[0]} cycles")
    names = ["prefetch issue", "S,dP kb0", "softmax kb0", "dQ kb0", "S,dP kb1", "softmax kb1", "dQ kb1", "store+flip", "barrier"]
    print("tile " + " ".join(f"{n:>14s}" for n in names) + "       total")
    for t in range(2, 9):
        st = [buf[t * 12 + i] for i in range(10)]
        if st[9] == 0: break
        print(f"{t - 1:4d} " + " ".join(f"{st[i + 1] - st[i]:14d}" for i in range(9)) + f" {st[9] - st[0]:11d}")
    sys.exit(0)
if which == 0:
    names = ["prefetch issue", "QK^T", "softmax", "PV", "store", "barrier"]
    print("tile " + " ".join(f"{n:>14s}" for n in names) + "       total")
    for t in range(1, 12):
        st = [buf[t * 12 + i] for i in range(7)]
        if st[6] == 0: break
        print(f"{t:4d} " + " ".join(f"{st[i + 1] - st[i]:14d}" for i in range(6)) + f" {st[6] - st[0]:11d}")
    sys.exit(0)
names = ["reads+A0a", "A0b", "A1a|B0'", "A1b|B0\"", "C0a|B1'", "C0b|B1\"", "C1", "vmcnt wait", "store+flip", "barrier"]
print("tile " + " ".join(f"{n:>11s}" for n in names) + "       total")
for t in range(1, 12):
    st = [buf[t * 12 + i] for i in range(11)]
    if st[10] == 0: break
    d = [st[i + 1] - st[i] for i in range(10)]
    print(f"{t:4d} " + " ".join(f"{x:11d}" for x in d) + f" {st[10] - st[0]:11d}")
