"""RCCL readiness: the driver's own N>1 launch (`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`) on the
REAL "nccl" backend (= RCCL over xGMI on ROCm), one rank per GPU.  Self-skips on a box with fewer than two devices (the
1-GPU test box); tests/test_bench_contract_gpu.py and tests/test_ddp_gpu.py cover the same code path over gloo there.
bench.py itself asserts dist.get_world_size() == N and reports whether the replicas' parameters are bit-identical after the
timed steps (every rank trains on its own batch; only the exchanged gradients couple them)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_rccl(dev):
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"RCCL needs one rank per GPU and this box exposes {ndev} device(s) (rocm-smi / HIP_VISIBLE_DEVICES="
                    f"{os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}): the nccl backend has NOT been exercised here; the same exchange "
                    f"code runs over gloo in tests/test_ddp_gpu.py (2 ranks sharing this GPU) and tests/test_ddp_gloo.py (CPU)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FS2_BENCH_BACKEND", None)
    env.pop("FS2_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--windows", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["world_size"] == 2 and cfg["backend"] == "nccl" and cfg["parallelism"] == "dp2"
    assert cfg["replicas_bit_identical"] is True
    frames_per_step = d["value"] * d["ms_per_step"] * 1e-3
    assert 2 * 0.75 * 48 * 925 < frames_per_step <= 2 * 48 * 925 * 1.001
