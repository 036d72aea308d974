"""kernel resource usage of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
    python tools/kres.py fastspeech2_amd/csrc/fs2_gemm.hip [name-substring]"""
import re
import subprocess
import sys

src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Ifastspeech2_amd/csrc", "-Iinclude", "-Wno-unused-result",
       "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.split("\n"):
    m = re.search(r"remark: (?:Function )?Name: (\S+)", l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if pat in name:
        print(f"{name[:70]:70s} vgpr {v.get('VGPRs')} agpr {v.get('AGPRs')} spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize')} occ {v.get('Occupancy')} lds {v.get('LDS Size')}")
