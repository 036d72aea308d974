"""fastspeech2_amd - the FastSpeech 2 train / batch-synthesis hot path on MI355X (gfx950): DESIGN.md."""
import os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (runtime default 4); streams that share a queue serialise.
# The batch-synthesis pipeline (utils.SynthPipeline) keeps four streams busy next to whatever the process created before (engine
# side streams, the prefetcher's copy stream): on the default it lost 15-20 % of its gain to queue sharing (same box, ms per
# batch: 5.10 alone, 6.0 - 6.2 with two to four earlier streams, 5.08 - 5.12 with 8 or 16 queues: profiles/r05zc_*, r05zd_*; the
# train step is unchanged: r05ze_*).
#
# Importing this package does NOT touch the environment (rounds 4-5 set the variable here, for single-process runs only: a
# library that rewrites os.environ on import is a poor neighbour inside someone else's train.py, and the measured and the
# N-rank configuration differed).  The entry points that own their process - train.py, synthesize.py, evaluate.py, bench.py -
# call configure_hw_queues() first thing in main(), with the SAME value for every world size, so the N-rank job runs the
# configuration the one-rank numbers were measured on; a launcher's children inherit the exported value.
HW_QUEUES_DEFAULT = 16
_MARK = "FASTSPEECH2_AMD_HW_QUEUES"                                  # value this package exported (tells "defaulted" from "user")
HW_QUEUES = {"value": None, "source": "runtime default"}          # what configure_hw_queues() decided, for logs / bench lines


def configure_hw_queues(n=HW_QUEUES_DEFAULT):
    """Set GPU_MAX_HW_QUEUES for this process (and the children it launches) unless the user exported a value.  Must run
    before the process's first HIP call - the runtime reads the variable once; returns the record {value, source}.
    `n` = None / 0 leaves the runtime's own default."""
    user = os.environ.get("GPU_MAX_HW_QUEUES")
    if user is not None and HW_QUEUES["source"] != "fastspeech2_amd":
        # (a rank started by a launcher that called this function inherits value + marker: still this package's decision)
        mine = os.environ.get(_MARK) == user
        HW_QUEUES.update(value=int(user), source="fastspeech2_amd" if mine else "user")
        return dict(HW_QUEUES)
    if not n:
        return dict(HW_QUEUES)
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("configure_hw_queues() after the first HIP call has no effect: call it first thing in main()")
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ[_MARK] = str(int(n))
    HW_QUEUES.update(value=int(n), source="fastspeech2_amd")
    return dict(HW_QUEUES)
