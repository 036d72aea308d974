"""ctypes binding of libfs2hip.so — the C-ABI drop-in boundary (include/fs2hip.h).

Prototypes are parsed from the header itself so the Python side cannot drift from the ABI.  There is NO
fallback: if the shared library is missing or a symbol is absent, importing the product path fails loudly.
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- must come first: torch bundles its own libamdhip64.so.7; loading ours afterwards makes
#                          libfs2hip.so bind to that SAME HIP runtime (two runtimes in one process cannot share streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FS2_LIB_PATH") or os.path.join(_HERE, "libfs2hip.so")   # FS2_LIB_PATH: dev builds (make prof)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fs2hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "fs2_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in fs2hip.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(fs2_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if a == "void":
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
                argnames.append(a.split("*")[-1].strip())
            else:
                parts = a.split()
                argtypes.append(_CTYPES[parts[-2]])
                argnames.append(parts[-1])
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes, argnames)
    return protos


class Fs2Error(RuntimeError):
    pass


_lib = None
_protos = None


def load():
    """Load libfs2hip.so and bind every symbol declared in the header. Raises if anything is missing."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Fs2Error(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback exists). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    for name, (ret, argtypes, _) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = ret
        fn.argtypes = argtypes
    _lib, _protos = lib, protos
    return lib


def call(name, *args):
    """Invoke an fs2_* entry point; negative return codes become exceptions (ValueError for bad arguments,
    mirroring the assert / ValueError behaviour of the reference's Python surface)."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.fs2_last_error().decode()
        if rc == -1:
            raise ValueError(f"{name}: {msg}")
        if rc == -2:
            raise TypeError(f"{name}: {msg}")
        raise Fs2Error(f"{name}: rc={rc}: {msg}")
    return rc


def kernel_source_sha(prefix="fs2_gemm"):
    """sha1 over the contraction kernels' sources (csrc/<prefix>*.hip|.h + the scheduler header): stamps measurement files that are
    only valid for the kernels they were taken on (profiles/*pmc_traffic.json: bench.py refuses a stale one)."""
    import glob
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(d, prefix + "*")) + [os.path.join(d, "fs2_sched.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
