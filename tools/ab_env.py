"""Dev tool: same-box A/B of the whole train step under development switches (the -DFS2_DEV library):
    python tools/ab_env.py "" FS2_WGRAD_WGS=512 FS2_WGRAD_WGS=1024,FS2_P_ORDER=0 ...
Each setting runs in its own process (the switches are read once), alternating, two rounds; bench.py itself refuses FS2_* variables,
so this calls its build / make_step pieces directly."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import bench
    class A: pass
    a = A(); a.dtype = "bf16"; a.batch = 48; a.phonemes = 128; a.workload = os.environ.get("AB_WORKLOAD", "ljspeech"); a.dec_layers = 4; a.frame_level = False; a.side_stream = 1
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    model, loss_fn, opt, b, _, _ = bench.build(a, dev, 0, 1)
    if os.environ.get("AB_LN_DEFER") == "0":                 # tool-level switch (the product reads no environment)
        model._engine.defer_ln_reduce = False
    if os.environ.get("AB_BRANCH") == "0":
        model._engine.concurrent_branches = False
    if os.environ.get("AB_SKIP_WGRAD") == "1":              # the critical chain alone (weight gradients not computed at all)
        from fastspeech2_amd import ops
        ops.conv_wgrad = lambda *a, **k: None
    if os.environ.get("AB_FUSE_LN") == "1":
        model._engine.fuse_proj_ln = True
    if os.environ.get("AB_FUSE_LN") == "stream":
        model._engine.fuse_proj_ln = "stream"
    if os.environ.get("AB_WGRAD_LATE") == "0":
        model._engine.wgrad_after_dgrad = False
    if os.environ.get("AB_LENS_FWD") in ("0", "1"):          # FFT-block contractions with / without lens (default: without)
        model._engine.gemm_lens_fwd = os.environ["AB_LENS_FWD"] == "1"
    if os.environ.get("AB_LENS_BWD") in ("0", "1"):
        model._engine.gemm_lens_bwd = os.environ["AB_LENS_BWD"] == "1"
    if os.environ.get("AB_SIDE") == "0":
        model._engine.use_side_stream = False
    # (stream priorities: torch.cuda.Stream.priority_range() is (0, -1) on this stack - the step runs on -1, the side stream on 0;
    # there is no lower level to try)
    step, _ = bench.make_step(model, loss_fn, opt, b, None)
    for _ in range(6):
        step()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("MS %.3f" % sorted(ts)[1], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(); sys.exit(0)
    for r in range(2):
        for setting in sys.argv[1:]:
            devlib = os.path.join(ROOT, "fastspeech2_amd", "libfs2hip_dev.so")        # (absent: the shipped library - runtime variables only)
            e = dict(os.environ, FS2_LIB_PATH=devlib if os.path.exists(devlib) else os.path.join(ROOT, "fastspeech2_amd", "libfs2hip.so"))
            e.update(dict(kv.split("=") for kv in setting.split(",") if kv))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e, capture_output=True, text=True, timeout=300)
            ms = [l for l in p.stdout.splitlines() if l.startswith("MS ")]
            print(f"{setting or '(default)':40s} {ms[0] if ms else 'FAILED ' + p.stderr[-300:]}", flush=True)
