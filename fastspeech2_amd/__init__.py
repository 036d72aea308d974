"""fastspeech2_amd - the FastSpeech 2 train / batch-synthesis hot path on MI355X (gfx950): DESIGN.md."""
import os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.  The
# batch-synthesis pipeline (utils.SynthPipeline) keeps four streams busy next to whatever the process created before (engine side
# streams, the prefetcher's copy stream): on the default it lost 15-20 % of its gain to queue sharing (same box, ms per batch: 5.10
# alone, 6.0 - 6.2 with two to four earlier streams, 5.08 - 5.12 with 8 or 16 queues: profiles/r05zc_*, r05zd_*; the train step is
# unchanged: r05ze_*).  The runtime reads the variable at its first HIP call, so a default set here - the package is imported before
# any device work - is in time; a value the user exported wins.  Single-process runs only: with more than one rank the runtime's
# own default stays (RCCL's channel kernels share the device with the step, and that combination has only ever been run on the
# default: nothing this builder can reach has more than one GPU).
if int(os.environ.get("WORLD_SIZE", "1") or "1") <= 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
