"""bench.py's output contract (GPU): ONE JSON line with the metric BASELINE.json names, whole-job value, the roofline object
(dominant kernel, live HIP-event timing, algorithmic FLOPs) and the cpu_baseline object (oracle on the host cores)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    """the driver's environment: no hardware-queue setting of the user's (bench.py decides, and says so in the line)"""
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FASTSPEECH2_AMD_HW_QUEUES")}
    env.update(extra)
    return env


def test_bench_line_contract(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("mel-frames/sec") and d["unit"] == "mel-frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    frames_per_step = d["value"] * d["ms_per_step"] * 1e-3                  # value = valid frames per step / step time
    assert d["value"] > 1e5 and 0.75 * 48 * 925 < frames_per_step <= 48 * 925 * 1.001, frames_per_step
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["kernel"].startswith("conv_gemm_p_kernel<false>") and r["launches_per_step"] >= 16 and r["avg_launch_us"] > 0
    assert 0 < r["frac_valid_rows"] <= r["frac"]                           # FLOPs of valid rows only (fully padded tiles are skipped)
    assert abs(r["gflop_per_launch"] * r["launches_per_step"] / r["kernel_ms_per_step"] - r["achieved"]) < 0.02 * r["achieved"]
    assert r["traffic"] is not None and r["traffic"]["bytes_per_launch"] > 0          # profiles/*_pmc_traffic.json travels with the repo
    # what this box sustains, next to the nominal peaks: matrix pipes, HBM copy, L2 -> LDS-DMA stream
    assert 0.5 < r["mfma_sustained"]["frac_of_nominal_peak"] <= 1.0
    assert 2.0 < r["hbm_copy"]["tb_per_s"] < 8.0 and 2.0 < r["l2_to_lds"]["tb_per_s"] < 60.0
    # tamper evidence + measurement quality: the shipped library, no development variables, the timed
    # window repeated and the median reported, the reference's own arithmetic (fp32) as a secondary figure of the same line
    cfg = d["config"]
    assert cfg["library"] == "fastspeech2_amd/libfs2hip.so" and cfg["dev_env"] == []
    assert cfg["hw_queues"] == {"value": 16, "source": "fastspeech2_amd"} and 0.7 < cfg["valid_row_fraction"] <= 1.0
    assert cfg["windows"] >= 5 and len(cfg["window_ms_per_step"]) == cfg["windows"]
    assert sorted(cfg["window_ms_per_step"])[cfg["windows"] // 2] == d["ms_per_step"]
    assert cfg["fp32_ms_per_step"] > d["ms_per_step"] and 0.05 < cfg["fp32_frac_of_f32_peak"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "mel-frames/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "median of 8" in c["sample"] and "2 discarded" in c["sample"] and f"{c['cores']} torch threads" in c["sample"]
    assert c["cores"] <= 64                                                # swept, never every logical CPU of the box
    assert d["value"] > 100 * c["value"]
    # the weight-gradient family is in the line (it was invisible next to the "dominant" kernel)
    w = r["wgrad_family"]
    assert w["launches_per_step"] >= 40 and w["ms_per_step"] > 0 and abs(w["frac"] - w["tflops"] / r["peak"]) < 1e-3
    assert 0 < w["taps_ge_3"]["ms_per_step"] <= w["ms_per_step"]
    # the second half of BASELINE's metric rides on the driver's line: batch-synthesis RTF on the val.txt-shaped workload
    sy = cfg["synth"]
    assert 0 < sy["rtf"] < 0.01 and sy["steps"] == 64 and "val.txt" in sy["workload"] and sy["mel_frames_per_s"] > 1e4
    assert abs(sy["rtf"] * sy["x_realtime"] - 1.0) < 0.02
    # ... and in the reference's own arithmetic beside it (fp32 acoustic model + fp32 vocoder, same batches, same loop)
    assert sy["fp32_rtf"] > sy["rtf"] and sy["fp32_ms_per_step"] > sy["ms_per_step"] and 0 < sy["fp32_frac_of_f32_peak"] < 1
    sr = sy["roofline"]
    assert sr["bound"] == "mfma" and 0 < sr["frac"] < 1 and abs(sr["frac"] - sr["achieved"] / sr["peak"]) < 1e-3
    sc = sy["cpu_baseline"]
    assert sc["kind"] == "port" and sc["value"] > 10 * sy["rtf"] and sc["cores"] >= 1


def test_bench_refuses_development_switches(dev):
    env = dict(os.environ, FS2_GEMM_ABL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 3 and "FS2_GEMM_ABL" in out.stderr


def test_bench_synth_line_has_roofline(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "synth", "--steps", "16", "--warmup", "16",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][0])
    assert d["higher_is_better"] is False and d["value"] > 0 and d["steps"] == 16 and "val.txt" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["kernel"].startswith("conv_") and r["launches_per_step"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_two_ranks_shared_gpu(dev):
    """The N>1 launch the driver uses (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), exercised with two
    ranks on this box's single GPU: gloo instead of RCCL (RCCL refuses two ranks on one device), everything else — gradient
    exchange hooks, global loss counts, barrier + max-over-ranks timing, rank 0's local roofline replay, joint exit — as shipped."""
    env = _clean_env(FS2_BENCH_BACKEND="gloo", FS2_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 96
    assert d["config"]["parallelism"] == "dp2" and d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["replicas_bit_identical"] is True
    frames_per_step = d["value"] * d["ms_per_step"] * 1e-3                  # whole-job: both ranks' valid frames
    assert 2 * 0.75 * 48 * 925 < frames_per_step <= 2 * 48 * 925 * 1.001, frames_per_step
    assert d["roofline"]["kernel"].startswith("conv_gemm_p_kernel<false>") and d["roofline"]["launches_per_step"] >= 16, d["roofline"]
    assert d["cpu_baseline"] is None                                        # reported at N=1 only
    assert "synth" not in d["config"]                                       # (replicas only; reported at N=1)
    ms = d["config"]["per_rank_local_ms"]
    assert len(ms) == 2 and min(ms) > 0 and abs(d["config"]["slowest_over_fastest"] - max(ms) / min(ms)) < 1e-2
    # what the N = 8 line will be read for: the exchange's schedule, every rank's local step, and the part of the
    # exchange backward did not hide = data-parallel step - slowest local step
    ex = d["config"]["exchange"]
    assert ex["collectives_total"] > 0 and ex["last_step_under_backward"] >= 1 and ex["last_step_in_finish"] <= 1
    assert 100 < ex["flat_gradient_mb"] < 125
    assert abs(d["config"]["exchange_exposed_ms"] - (d["ms_per_step"] - max(ms))) < 2e-3
    # ranks SHARING one device keep the runtime's hardware-queue default (their queues add up on it); one rank per GPU gets 16 at
    # every world size (test_bench_line_contract, tests/test_nccl_gpu.py)
    assert d["config"]["hw_queues"] == {"value": None, "source": "runtime default"}


def test_bench_two_ranks_libritts_buckets_differ(dev):
    """config 4's shape: every rank takes ITS OWN bucket of one step from the real length-bucketed sampler; the line carries the
    per-rank shapes and local step times."""
    env = dict(os.environ, FS2_BENCH_BACKEND="gloo", FS2_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "libritts",
           "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][0])
    c = d["config"]
    assert c["replicas_bit_identical"] is True and len(c["per_rank_T"]) == 2 and len(c["per_rank_frames"]) == 2
    assert c["per_rank_frames"][0] != c["per_rank_frames"][1]              # different buckets ...
    assert abs(c["per_rank_L"][0] - c["per_rank_L"][1]) <= 0.2 * max(c["per_rank_L"])   # ... of the same length class (card-wise deal)
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - sum(c["per_rank_frames"])) <= 1e-3 * sum(c["per_rank_frames"])
