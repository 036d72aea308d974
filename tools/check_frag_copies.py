"""Dev tool: the persistent / wide contraction kernels track their LDS fragment reads by hand (inline-asm ds_read_b128 + counted
s_waitcnt lgkmcnt): the compiler does not know that a fragment register may still be IN FLIGHT, so a register-to-register copy of
one between its read and the wait that lands it carries stale data.  This scans a kernel's ISA (hipcc -S) and lists every v_mov /
v_accvgpr move whose SOURCE is a destination of an asm ds_read_b128 and that is not preceded (in the same straight-line run since the
last such read) by an lgkmcnt(0) wait.
    python tools/check_frag_copies.py fastspeech2_amd/csrc/fs2_gemm_p.hip [kernel-name-substring]"""
import re, subprocess, sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_gemm_p_kernel"
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Ifastspeech2_amd/csrc", "-Iinclude",
                      "-Wno-unused-result", "-S", "--cuda-device-only", "-o", "-", src] + sys.argv[3:], capture_output=True, text=True).stdout
kern, cur = {}, None
for l in asm.split("\n"):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        cur = m.group(1); kern[cur] = []
    elif cur is not None:
        kern[cur].append(l)
        if "s_endpgm" in l:
            cur = None

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

bad_total = 0
for name, lines in kern.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    inflight, bad = set(), []
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith("ds_read_b128"):
            inflight |= regs(t.split()[1].rstrip(","))
        elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            inflight = set()
        elif t.startswith("v_mov_b") or t.startswith("v_accvgpr"):
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            if len(ops) >= 2 and regs(ops[1]) & inflight:
                bad.append((i, t))
    print(f"{dem[:90]:90s} copies of in-flight fragment registers: {len(bad)}")
    for i, t in bad[:6]:
        print(f"      line {i}: {t}")
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
