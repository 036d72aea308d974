"""Data-parallel gradient exchange over RCCL/xGMI (replaces the reference's nn.DataParallel, train.py:42).

One process per GPU.  Every trainable gradient lives in ONE flat fp32 buffer laid out in backward-completion
order (fastspeech2_amd.model.FastSpeech2._trainable_in_backward_order), so the exchange is a few large
contiguous all-reduces instead of 183 small ones: xGMI is point-to-point (7 links x ~153 GB/s per GPU) and
ring all-reduce is per-link bound, so big buckets amortise the per-collective latency best.  Buckets are
launched on a side stream as soon as the engine reports that a prefix of the flat buffer is final, which
overlaps the exchange with the rest of backward (the decoder's gradients travel while the encoder's are still
being computed).

The reference computes its loss over the gathered global batch (train.py:82-86).  `global_counts` provides
the all-reduced valid-position counts so each rank can normalise its loss terms by (global count / world):
averaging gradients over ranks then reproduces the global-batch mean exactly.

Device-agnostic on purpose: the same code runs with the `gloo` backend on CPU tensors in the unit tests.
"""
import torch
import torch.distributed as dist


class GradExchange:
    def __init__(self, flat_grad, world_size=None, bucket_bytes=32 << 20, group=None, overlap=True, tail_bucket_bytes=None,
                 tail_bytes=48 << 20, collectives=None):
        """`collectives`: None = issue the all-reduces whenever a process group exists - ALSO over a one-rank group (the identity
        reduction costs one small RCCL launch per bucket and keeps the one-rank run the same program as the N-rank run); without a
        process group there is nothing to call and the exchange is a no-op - unless world_size > 1, which raises here (replicas
        that never exchange anything must not look like a training job).  False forces the no-op; True keeps the exchange active
        whatever the process state (schedule inspection in the tests: the first real all_reduce raises without a group).

        Buckets are `bucket_bytes` except over the LAST `tail_bytes` of the flat buffer (the encoder's gradients, final only
        when backward ends), which travel in `tail_bucket_bytes` pieces (default bucket_bytes / 4): whatever is still in flight
        when backward finishes is exposed in front of the clip + Adam pass (it needs the norm of ALL gradients, so it cannot
        start per bucket), and a smaller last piece makes that exposure a quarter of a bucket instead of a whole one."""
        self.flat = flat_grad
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.n = flat_grad.numel()
        tb = tail_bucket_bytes if tail_bucket_bytes is not None else max(1, bucket_bytes // 4)
        self.tail_elems = max(1, tb // flat_grad.element_size())
        self.tail_start = max(0, self.n - tail_bytes // flat_grad.element_size())
        have_pg = dist.is_available() and dist.is_initialized()
        if have_pg and self.world != dist.get_world_size(group):
            # RCCL averages over the GROUP's size inside the collective, the gloo path divides by self.world: a caller-supplied
            # world_size that differs from the group's would make the two backends disagree silently
            raise ValueError(f"GradExchange: world_size={self.world} but the process group has {dist.get_world_size(group)} ranks")
        if not have_pg and self.world > 1 and collectives is None:
            raise RuntimeError(f"GradExchange: world_size={self.world} without an initialised process group - the replicas would "
                               "train unsynchronised (call torch.distributed.init_process_group first)")
        self.active = have_pg if collectives is None else bool(collectives)
        self.cuda = flat_grad.is_cuda
        self.overlap = overlap and self.cuda
        # RCCL averages inside the collective (ncclAvg): no scaling pass over the bucket.  gloo has no AVG: divide, then SUM
        # (one more read + write of the bucket - 220 MB per step at the real layout, on the high-priority stream)
        self.native_avg = have_pg and dist.get_backend(group) == "nccl"
        # high priority: the collective's few workgroups must get CUs while backward still fills the device, otherwise the
        # exchange only starts moving once compute drains and nothing overlaps
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device, priority=-1) if self.cuda else None
        self.n_buckets = 0           # all_reduce calls issued so far (cumulative)
        self.last_step = (0, 0)      # (buckets launched by ready() = under backward, buckets launched by finish()) of the last step
        self._early = 0
        self.reset()

    def reset(self):
        self.sent = 0          # prefix [0, sent) already handed to the collective
        self.handles = []
        self._early = 0

    def _launch(self, lo, hi, producers=()):
        view = self.flat[lo:hi]
        if self.cuda and self.overlap:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for s in producers:                 # e.g. the engine's weight-gradient stream: only the collective waits for it
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                h = self._reduce(view)
        else:
            if self.cuda:
                for s in producers:
                    torch.cuda.current_stream().wait_stream(s)
            h = self._reduce(view)
        self.handles.append(h)
        self.n_buckets += 1

    def _reduce(self, view):
        if self.native_avg:
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        if self.world > 1:
            view.div_(self.world)
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def ready(self, end, producers=()):
        """Engine hook: gradients in flat[0:end) are final once the work queued so far on the current stream and on every
        stream in `producers` has run.  Launch every full bucket inside that prefix."""
        if not self.active:
            return
        end = min(end, self.n)
        while True:
            size = self.bucket_elems if self.sent + self.bucket_elems <= self.tail_start else self.tail_elems
            if end - self.sent < size:
                break
            self._launch(self.sent, self.sent + size, producers)
            self.sent += size
            self._early += 1

    def finish(self):
        """Flush the tail bucket and make the compute stream wait for all collectives."""
        if not self.active:
            self.reset()
            return
        late = 0
        if self.sent < self.n:
            self._launch(self.sent, self.n)
            self.sent = self.n
            late = 1
        self.last_step = (self._early, late)
        for h in self.handles:
            h.wait()
        if self.cuda and self.overlap:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


class CountExchange:
    """The loss's valid-position counts, all-reduced AHEAD of the forward pass.

    The counts depend only on the batch's lengths (sum of min(src_len, L), sum of min(mel_len, T)), so `start` launches their
    all-reduce on the communication stream as soon as the batch is on the device and the train step's forward overlaps it;
    the instance is passed to FastSpeech2Loss as `count_reduce` and hands the result over (the compute stream waits for the
    collective's event, the host never blocks).  Without a pending `start` it falls back to the blocking `global_counts`."""

    def __init__(self, group=None):
        self.group = group
        self._pending = None
        self._stream = None

    def start(self, src_lens, mel_lens, L, T):
        if not dist.is_initialized():
            return
        c = torch.stack([src_lens.to(torch.int64).clamp(max=L).sum(), mel_lens.to(torch.int64).clamp(max=T).sum()]).float()
        if c.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=c.device, priority=-1)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                h = dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            c.record_stream(self._stream)
        else:
            h = dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (c, h)

    def __call__(self, counts):
        if self._pending is None:
            return global_counts(counts, self.group)
        c, h = self._pending
        self._pending = None
        h.wait()                                     # device tensors: makes the CURRENT stream wait for the collective
        if c.is_cuda:
            torch.cuda.current_stream().wait_stream(self._stream)
        return c / dist.get_world_size(self.group)


def global_counts(counts, group=None):
    """all-reduce (sum) a small tensor of valid-position counts; returns counts / world (so that a local loss
    normalised by it, averaged over ranks, equals the global-batch mean)."""
    if not dist.is_initialized():
        return counts
    c = counts.clone().float()
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return c / dist.get_world_size(group)


def shard_by_length(lengths, world_size, rank, batch_size):
    """Length-bucketed sharding (replaces dataset.py:127-146's sort-within-4x-batch for the multi-GPU case):
    sort the global index list by length, cut it into groups of world_size*batch_size, and deal each group's
    consecutive batch_size-chunks to the ranks so that every rank gets similar lengths in the same step.
    Returns the list of index lists (one per step) for `rank`."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    per_step = world_size * batch_size
    steps = []
    for s in range(0, len(order) - per_step + 1, per_step):
        grp = order[s:s + per_step]
        steps.append(grp[rank * batch_size:(rank + 1) * batch_size])
    return steps
