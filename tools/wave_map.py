"""Dev tool: wave -> SIMD placement of a 640-thread workgroup with 133 KB of LDS (the streaming kernel's shape).  Needs the dev lib."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastspeech2_amd import _lib
lib = _lib.load()
out = torch.zeros(8 * 16, dtype=torch.int32, device="cuda:0")
for threads in (640, 512, 768):
    out.zero_()
    lib.fs2_dev_wave_map(ctypes.c_void_p(out.data_ptr()), 8, threads, 133 * 1024, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(8, 16)
    for b in range(3):
        print(threads, "block", b, "simd per wave:", [(int(v) >> 4) & 3 for v in o[b, : threads // 64]], "cu", [(int(v) >> 8) & 15 for v in o[b, :1]])
