"""CPU: the C-ABI library loads without a GPU and exports every symbol include/fs2hip.h declares (no compute
calls); argument validation returns the documented error codes before any launch."""
import ctypes
import os
import re
import subprocess

import pytest

from fastspeech2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    protos = _lib.parse_header()
    assert len(protos) >= 33
    for name in protos:
        assert hasattr(lib, name), name
    # and nothing the header promises is missing from the dynamic symbol table
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\b(fs2_\w+)\b", out))
    assert set(protos) <= exported
    # every exported fs2_* entry point is declared (no undocumented ABI)
    undeclared = {s for s in exported if not s.startswith("fs2_set_error")} - set(protos)
    assert not undeclared, undeclared


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.fs2_version() >= 100
    assert isinstance(lib.fs2_last_error(), bytes)


def test_bad_arguments_are_rejected_before_launch():
    """FS2_EINVAL (-1) -> ValueError, FS2_EDTYPE (-2) -> TypeError: mirrors the reference's assert/ValueError style."""
    with pytest.raises(ValueError):
        _lib.call("fs2_conv_gemm", None, 0, None, None, None, 0, None, 0, None, None, 0, 0, 0, 0, 1, 1, 0, 0, 0.0, 0, 0.0, 0, 1.0,
                  0, None)
    with pytest.raises((TypeError, ValueError)):
        _lib.call("fs2_cast", None, 7, None, 9, 16, None)


def test_header_is_plain_c():
    """include/fs2hip.h must compile as C (extern "C" boundary, no C++/torch types)."""
    src = '#include "fs2hip.h"\nint main(void){return fs2_version()>0?0:1;}\n'
    p = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


def test_product_forward_refuses_cpu_tensors():
    import torch
    from tests.golden import configs
    from fastspeech2_amd.model import FastSpeech2

    pcfg, mcfg = configs.make(dec_layers=1, enc_layers=1)
    m = FastSpeech2(pcfg, mcfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, dtype=torch.long), torch.ones(1, 4, dtype=torch.long), torch.tensor([4]), 4)


def test_dispatch_queries_are_pure_host_functions():
    """fs2_conv_gemm_variant / fs2_conv_gemm_lrelu_io_variant answer without a device (bench.py attributes its HIP-event durations
    to kernel names with them): HiFi-GAN's stored-leaky-ReLU launches with a residual / accumulate operand and a short reduction
    go to the ring kernel (3), the same shapes without one to the persistent kernel (5); the FFN convolution stays on 5."""
    lib = _lib.load()
    BF16 = 1
    for (B, S, C, k) in [(4, 12000, 128, 7), (8, 5600, 256, 11), (4, 12000, 128, 3)]:
        M = B * S
        assert lib.fs2_conv_gemm_lrelu_io_variant(C, C, C, 0, M, C, C, S, k, 1, 0, 10.0, 0.1, BF16) == 3
        assert lib.fs2_conv_gemm_lrelu_io_variant(C, C, 0, 1, M, C, C, S, k, 1, 0, 0.0, 0.1, BF16) == 3
        assert lib.fs2_conv_gemm_lrelu_io_variant(C, C, 0, 0, M, C, C, S, k, 1, 0, 0.0, 0.1, BF16) == 5
    assert lib.fs2_conv_gemm_variant(256, 1024, 0, 0, 0, 48 * 925, 1024, 256, 925, 9, 1, 0, 0.0, BF16) == 5
    assert lib.fs2_conv_gemm_variant(256, 768, 0, 0, 0, 48 * 925, 768, 256, 925, 1, 1, 0, 0.0, BF16) == 9


def test_no_copy_of_an_in_flight_fragment_register_in_the_persistent_kernel():
    """The persistent contraction kernel tracks its LDS fragment reads by hand (inline-asm ds_read_b128 + counted lgkmcnt); the
    compiler does not know that such a register may still be in flight, so a register-to-register copy of one between its read and
    the wait that lands it carries stale data (round 6: the eight-consumer-wave form came out 1 % wrong, timing-dependent, when its
    K loop was two nested loops).  tools/check_frag_copies.py compiles the kernel to ISA and lists such copies: there must be none."""
    import shutil
    import subprocess
    import sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_frag_copies.py"), os.path.join(ROOT, "fastspeech2_amd", "csrc", "fs2_gemm_p.hip")],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "conv_gemm_p_kernel<false, false, 0, 8>" in r.stdout and "conv_gemm_p_kernel<false, false, 0, 4>" in r.stdout
