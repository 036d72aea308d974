"""Dev tool: same-box A/B of the batch-synthesis loop under development switches (the -DFS2_DEV library) and tool-level settings:
    FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so python tools/ab_synth_env.py "" FS2_GEMM_P=0 AB_VOC_STREAMS=2 AB_VOC_STREAMS=4,FS2_GEMM_P=0
Each setting runs in its own process (the switches are read once), alternating, two rounds; bench.py refuses FS2_* variables, so this
calls bench.synth_measure directly.  AB_VOC_STREAMS=n: utils.SynthPipeline(voc_streams=n); AB_PIPE=0: the sequential loop; AB_DUMMY_STREAMS=n / AB_SET_HIGH=1: n more
streams created (and used once) / a high-priority current stream before the measurement, as bench.py's train mode has; any other
variable (GPU_MAX_HW_QUEUES=8) is passed to the child's environment as it is."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    if os.environ.get("AB_LATE_QUEUES"):               # set AFTER `import torch`, before the first HIP call: does the runtime still see it?
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ["AB_LATE_QUEUES"]
    import bench
    a = bench.parse(["--mode", "synth", "--no-cpu-baseline", "--no-roofline"])
    a.synth_voc_streams = int(os.environ.get("AB_VOC_STREAMS", a.synth_voc_streams))
    a.no_synth_pipeline = os.environ.get("AB_PIPE") == "0"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # what bench.py's train mode has done before it measures synthesis: a high-priority current stream, engines with side streams
    keep = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("AB_DUMMY_STREAMS", "0")))]
    for st in keep:
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
    if os.environ.get("AB_SET_HIGH") == "1":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    r = bench.synth_measure(a, dev, 0, 1, 64, 64, False)
    print("MS %.3f RTF %.3e" % (r["dt"] / 64 * 1e3, r["dt"] / r["audio_s"]), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD") == "1":
        one()
        sys.exit(0)
    settings = sys.argv[1:] or [""]
    for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
        for s in settings:
            env = dict(os.environ, AB_CHILD="1")
            for kv in filter(None, s.split(",")):
                k, v = kv.split("=", 1)
                env[k] = v
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("MS ")]
            print(f"round {rnd + 1} [{s or 'default'}]: {line[0] if line else 'FAILED ' + out.stderr[-300:]}", flush=True)
