"""Hand-scheduled forward / backward of FastSpeech 2 over the HIP kernel library (no autograd tape inside).

The reference gets its backward from torch autograd over ~400 small ATen ops per step
(reference model/fastspeech2.py:43-110 and everything it calls).  Here the whole model is ONE autograd node:
`Engine.forward` launches the fused kernels in order and keeps exactly the activations the hand-written
backward needs; `Engine.backward` launches the gradient kernels in reverse and accumulates parameter
gradients straight into the flat gradient buffer (so gradient exchange and the optimiser see contiguous
memory).  All launches go to the current HIP stream with static shapes -> the full step is hipGraph-capturable.

Activation layout: time-major rows [B*S, C]; padding handled by per-sequence lengths (int32), never by masks.
"""
import numpy as np
import torch

from . import ops
from .ops import ACT_GATE, ACT_NONE, ACT_RELU, ACT_TANH

_GOLD = 0x9E3779B97F4A7C15


def _site_seed(i):
    return (i * _GOLD + 0x1234567) & 0x7FFFFFFFFFFFFFFF


def _seed_pair(site, step_seed):
    """(seed argument, device seed pointer) of a dropout site.  `step_seed` is this forward's position in the dropout stream:
    a Python int in eager mode (added on the host: no counter kernel, no clone per step) or a device tensor when the step is
    being captured into a hipGraph (the kernels add *seed_dev, so replays draw fresh masks)."""
    if step_seed is None or isinstance(step_seed, int):
        return (_site_seed(site) + (step_seed or 0)) & 0xFFFFFFFFFFFFFFFF, None
    return _site_seed(site), step_seed


def _seed_kw(name, site, step_seed):
    v, d = _seed_pair(site, step_seed)
    return {name: v, "seed_dev": d}


class _Saved:
    pass


class Engine:
    def __init__(self, model, device):
        self.m = model
        self.device = device
        self.cdt = model.compute_dtype
        self.weights_dirty = True
        self._packed = {}
        self._flat_lp = None
        self.lp_synced = False
        self._pe_cache = {}
        self.seed_counter = torch.zeros(1, device=device, dtype=torch.int64)
        self.device_seed = False        # True: dropout position read from device memory (graph capture: bench.py --graph 1)
        self.reseed()
        mc = model.model_config
        self.tc = mc["transformer"]
        self.d = self.tc["encoder_hidden"]
        self.max_seq_len = mc["max_seq_len"]
        self.vp = mc["variance_predictor"]
        self.P = dict(model.named_parameters())
        self.Bf = dict(model.named_buffers())
        self.G = None   # name -> grad view (built lazily)
        self.grad_hook = None   # callable(end_offset, producer_streams): flat_grad[0:end_offset) is final (see ddp.GradExchange.ready)
        self.use_side_stream = True     # weight gradients on a side HIP stream (set False for single-stream profiling: bench.py --side-stream 0)
        self._side_stream = None
        self._side = None
        self._side_keep = []
        self._ln_pending = []
        self.defer_ln_reduce = True
        # Whether the FFT blocks' contractions carry lens is decided PER BATCH (r03u / r03z).  With lens the persistent / wide
        # kernels leave out the 256-row tiles that lie wholly in a sequence's tail and zero the padded rows of the others in
        # their epilogue - and that epilogue and the tile-map walk cost more than they save when few tiles can be left out: the
        # LJSpeech bench batch (2 % of the tiles) runs the k = 9 forward convolution in 209-217 us with lens and 189 us
        # without (`profiles/r03u_bench_conv.log`; step -0.07 ms), the LibriTTS-shaped bucket (46 % valid rows, ~40 % of the
        # tiles) runs the step in 8.94 ms with lens and 9.26 without (`profiles/r03z_ab_env_libritts.log`).  Nothing needs the
        # zeros: LayerNorm (forward and backward) masks padded rows itself, attention and the weight gradients never read rows
        # t >= lens[b], and every backward operand is exactly zero there already, so without lens the padded rows of the results
        # are finite values nobody consumes.  The decision uses the lengths' HOST copy (utils.lens_to_device: no device round
        # trip); a lengths vector without one (inference: frame counts are computed on the device) runs without lens.
        self.gemm_lens_fwd = None       # None: per batch (skippable-tile fraction >= lens_skip_min); True / False: forced
        self.gemm_lens_bwd = None
        self.lens_skip_min = 0.10
        # explicit host copies of the NEXT batch's lengths (FastSpeech2.set_length_hint): take precedence over the copies riding on
        # the device tensors (`_fs2_host`, lost by .to() / clone / slicing, stale when a static buffer is refilled with copy_)
        self.length_hint = None
        self._wgrad_ws = ops.WgradWorkspaces()      # split-K scratch of the weight gradients: lives and dies with this engine
        self.fork_in_capture = False    # debug / measurement switch: let the side stream fork INSIDE a hipGraph capture (tools/dbg_fork_capture.py)
        self.res_in_ln = True           # see _fft_bwd: residual-branch gradients are added by the LayerNorm backward below, not by the dgrad epilogue
        self.fuse_proj_ln = False       # see _proj_ln: True = every N = 256 projection, "stream" = only where the streaming kernel runs it
        self._pack_pending = False
        self._bn_ws = {}
        self._bn_eval = {}
        self._tail_ws = [None, None]
        self._on_side = False
        self.concurrent_branches = True

    # ------------------------------------------------------------------ dropout stream
    _SEED_INC = 0x632BE59BD9B4E019 & 0x7FFFFFFFFFFFFFFF

    def reseed(self, seed=None, rank=None, step=0):
        """Dropout masks are a pure function of (site, element, counter); the counter starts from a mix of the process seed
        (torch.initial_seed(), i.e. torch.manual_seed / train.py's seed) and the data-parallel rank - replicas draw INDEPENDENT
        masks like the reference's DataParallel replicas do - and advances once per training forward; `step` fast-forwards it
        (resume from --restore_step continues the mask sequence instead of replaying it from the start)."""
        import os
        if seed is None:
            seed = torch.initial_seed()
        if rank is None:
            rank = int(os.environ.get("RANK", "0"))
        x = (int(seed) + (rank + 1) * _GOLD) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF          # splitmix64 finaliser
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        self.base_seed = x & 0x7FFFFFFFFFFFFFFF
        self._seed_k = int(step)
        self.seed_counter.fill_((self.base_seed + int(step) * self._SEED_INC) & 0x7FFFFFFFFFFFFFFF)

    # ------------------------------------------------------------------ weights
    def _flat_view(self, first_name, numel, shape, grad=False):
        o = self.m._flat_offsets[first_name]
        buf = self.m._flat_grad if grad else self.m._flat
        return buf[o:o + numel].view(shape)

    def _conv_list(self):
        """(key, weight tensor view (Cout,Cin,k), bias view) for every contraction of the model."""
        d = self.d
        out = []

        def fft(prefix):
            a = prefix + "slf_attn."
            out.append((a + "qkv", self._flat_view(a + "w_qs.weight", 3 * d * d, (3 * d, d, 1)),
                        self._flat_view(a + "w_qs.bias", 3 * d, (3 * d,))))
            out.append((a + "fc", self.P[a + "fc.weight"], self.P[a + "fc.bias"]))
            f = prefix + "pos_ffn."
            out.append((f + "w_1", self.P[f + "w_1.weight"], self.P[f + "w_1.bias"]))
            out.append((f + "w_2", self.P[f + "w_2.weight"], self.P[f + "w_2.bias"]))

        for i in range(self.tc["encoder_layer"]):
            fft(f"encoder.layer_stack.{i}.")
        for i in range(self.tc["decoder_layer"]):
            fft(f"decoder.layer_stack.{i}.")
        for kind in ("duration", "pitch", "energy"):
            pre = f"variance_adaptor.{kind}_predictor.conv_layer."
            out.append((pre + "conv1d_1", self.P[pre + "conv1d_1.conv.weight"], self.P[pre + "conv1d_1.conv.bias"]))
            out.append((pre + "conv1d_2", self.P[pre + "conv1d_2.conv.weight"], self.P[pre + "conv1d_2.conv.bias"]))
        out.append(("mel_linear", self.P["mel_linear.weight"], self.P["mel_linear.bias"]))
        for i in range(5):
            pre = f"postnet.convolutions.{i}.0.conv"
            out.append((pre, self.P[pre + ".weight"], self.P[pre + ".bias"]))
        return out

    def _build_layout(self):
        """One-time description of every contraction's weight inside the flat parameter buffer: the forward pack of a
        layer is a VIEW (master weights are stored in the GEMM's own [n][tap][c] order; for bf16 compute the view is into
        the bf16 shadow copy of the flat buffer that the Adam kernel maintains), the data-gradient packs live in one
        buffer filled by a single fs2_pack_dgrad_multi launch."""
        m = self.m
        lowp = self.cdt != torch.float32
        flat = m._flat
        self._flat_lp = torch.empty(flat.numel(), device=flat.device, dtype=self.cdt) if lowp else None
        self.lp_synced = False
        src = self._flat_lp if lowp else flat
        entries, rows, wd_off, tile0 = [], [], 0, 0
        base = flat.data_ptr()
        for key, w, b in self._conv_list():
            if w.dim() == 3:
                w = w.permute(0, 2, 1)                  # the contiguous tap-major storage [Cout][k][Cin]
                cout, k, cin = w.shape
            else:
                cout, cin = w.shape
                k = 1
            assert w.is_contiguous()
            off = (w.data_ptr() - base) // 4
            n = cout * k * cin
            entries.append((key, off, wd_off, cout, cin, k, b))
            rows.append([off, wd_off, cout, cin, k, tile0])
            wd_off += (n + 7) // 8 * 8
            tile0 += ((cout + 63) // 64) * ((cin + 63) // 64) * k
        self._wd_all = torch.empty(wd_off, device=flat.device, dtype=self.cdt)
        self._pack_table = torch.tensor(rows, dtype=torch.int64, device=flat.device)
        self._pack_tiles = tile0
        packed = {}
        for key, off, wdo, cout, cin, k, b in entries:
            n = cout * k * cin
            wf = src[off:off + n].view(cout, k, cin)
            wd = self._wd_all[wdo:wdo + n].view(cin, k, cout)
            packed[key] = (wf, wd, b, (cout, k, cin))
        self._packed = packed
        self._packed_has_dgrad = False

    def lowp_buffer(self):
        """bf16 shadow of the flat parameter buffer (None for fp32 compute); the Adam kernel writes it in its own pass."""
        if not self._packed:
            self._build_layout()
        return self._flat_lp

    def weights(self, need_dgrad):
        """Compute-dtype weights of every contraction: Wf[n][tap][c] views (+ Wd[c][tap][n] packs for data gradients)."""
        if not self._packed:
            self._build_layout()
        if self.weights_dirty or (need_dgrad and not self._packed_has_dgrad):
            if self._flat_lp is not None and not self.lp_synced:
                ops.cast(self.m._flat, self.cdt, out=self._flat_lp)
                self.lp_synced = True
            if need_dgrad:
                # the data-gradient packs are first read in BACKWARD: pack them on the side stream, off the forward chain
                if self.use_side_stream and not torch.cuda.is_current_stream_capturing():
                    if self._side_stream is None:
                        self._side_stream = torch.cuda.Stream(device=self.device)
                    cur = torch.cuda.current_stream()
                    self._side_stream.wait_stream(cur)
                    with ops.pinned_stream(self._side_stream):
                        ops.pack_dgrad_multi(self.m._flat, self._wd_all, self._pack_table, self._pack_tiles)
                    self._pack_pending = True
                else:
                    ops.pack_dgrad_multi(self.m._flat, self._wd_all, self._pack_table, self._pack_tiles)
            self._packed_has_dgrad = need_dgrad
            self.weights_dirty = False
        return self._packed

    def _grads(self):
        if self.G is None:
            G = {}
            for n in self.m._flat_names:
                G[n] = self.m.grad_view(n)
            d = self.d
            for pre in [f"encoder.layer_stack.{i}.slf_attn." for i in range(self.tc["encoder_layer"])] + \
                       [f"decoder.layer_stack.{i}.slf_attn." for i in range(self.tc["decoder_layer"])]:
                G[pre + "qkv.weight"] = self._flat_view(pre + "w_qs.weight", 3 * d * d, (3 * d, d, 1), grad=True)
                G[pre + "qkv.bias"] = self._flat_view(pre + "w_qs.bias", 3 * d, (3 * d,), grad=True)
            self.G = G
        return self.G

    def _pe(self, table_param, n):
        """position table rows [0, n): the parameter while n fits, regenerated sinusoid otherwise
        (reference transformer/Models.py:82-91,145-162)."""
        tab = table_param[0]
        if n <= tab.shape[0]:
            return tab
        key = (n, tab.shape[1])
        if key not in self._pe_cache:
            from .model import sinusoid_table
            self._pe_cache[key] = sinusoid_table(n, tab.shape[1]).to(self.device)
        return self._pe_cache[key]

    def _ready(self, next_name):
        """flat_grad[0:offset(next_name)) is final once the work queued so far on the main AND the side stream is done.  The
        hook gets the side stream as a `producer`: the exchange makes its COMMUNICATION stream wait for it when (and only
        when) it launches a bucket — the main stream never waits for weight gradients at a layer boundary."""
        if self.grad_hook is not None:
            end = self.m._flat_offsets[next_name] if next_name else self.m._flat.numel()
            self.grad_hook(end, (self._side,) if self._side is not None else ())

    # ------------------------------------------------------------------ building blocks
    def _bn_workspace(self, C):
        """[forward statistics, backward sums] workspaces for C channels: zeroed ONCE here, kept consistent by the kernels
        themselves (fs2_bn_train_stats / fs2_bn_bwd_acc: slab partial sums + arrival counters, bit-reproducible column sums),
        shared by every BatchNorm layer of that width (stream order)."""
        ws = self._bn_ws.get(C)
        if ws is None:
            ws = self._bn_ws[C] = [ops.bn_workspace(C, self.device), ops.bn_workspace(C, self.device)]
        return ws

    def _bn_eval_stats(self, pre):
        """[running_mean | rsqrt(running_var + eps)] of an eval-mode BatchNorm layer, cached until the buffers change: torch bumps a
        tensor's _version on its own in-place writes (load_state_dict, .copy_); the training kernels update the running statistics
        through raw pointers, so FastSpeech2.train() / .eval() also drops the cache on every mode switch (r03n: without that a
        train -> eval -> train -> eval sequence, i.e. train.py's validation, evaluated with the FIRST validation's statistics).
        Was three tiny torch launches (add, rsqrt, cat) per PostNet layer and synthesis batch - 15 of a synthesis step's launches."""
        rm, rv = self.Bf[pre + "1.running_mean"], self.Bf[pre + "1.running_var"]
        key = (rm.data_ptr(), rm._version, rv.data_ptr(), rv._version)
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a tensor made inside a capture lives in the graph's pool and holds nothing until a replay: never cache it, and never
            # hand a captured graph a cached tensor whose key could go stale between replays
            return torch.cat([rm, torch.rsqrt(rv + 1e-5)])
        hit = self._bn_eval.get(pre)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([rm, torch.rsqrt(rv + 1e-5)]))
            self._bn_eval[pre] = hit
        return hit[1]

    def _lens_pays(self, lens, forced):
        if lens is None:
            return False
        if forced is not None:
            return bool(forced)
        return getattr(lens, "_fs2_skip", 0.0) >= self.lens_skip_min

    @staticmethod
    def _skip_fraction(host_lens, S):
        """fraction of the 256-row M-tiles of the [B * S] row space that lie wholly inside one sequence's padded tail (what
        fs2_tile_map lists as padded), from the lengths' host copy; 0 when there is none."""
        if host_lens is None:
            return 0.0
        lens = np.minimum(np.asarray(host_lens, dtype=np.int64), S)
        M = int(len(lens)) * int(S)
        if M == 0:
            return 0.0
        m0 = np.arange(0, M, ops.TILE_ROWS, dtype=np.int64)
        mlast = np.minimum(m0 + ops.TILE_ROWS - 1, M - 1)
        b0, b1 = m0 // S, mlast // S
        padded = (b0 == b1) & ((m0 - b0 * S) >= lens[b0])
        return float(padded.mean())

    @staticmethod
    def _tmap(lens):
        """tile map that ops.lens_prep made together with this lengths tensor (None otherwise: the non-persistent kernels run).
        It travels ON the tensor object - the pair cannot be separated, outlive one another or be confused with another lengths
        vector that happens to reuse the same device address (the round-2 version looked it up by data_ptr())."""
        return getattr(lens, "_fs2_tmap", None) if lens is not None else None

    def _gemm(self, W, key, x, S, taps=1, pad=0, act=ACT_NONE, lens=None, res=None):
        wf, _, b, shape = W[key]
        ragged = lens is not None
        if not self._lens_pays(lens, self.gemm_lens_fwd):
            lens = None
        return ops.conv_gemm(x, wf, b, S, taps=taps, pad=pad, act=act, lens=lens, res=res, tmap=self._tmap(lens),
                             tail_ws=self._tail_workspace(x.device), ragged=ragged)

    def _proj_ln(self, W, key, x, res, ln, lens, B, S, p, site, seed_dev):
        """N = 256 projection -> dropout -> + residual -> LayerNorm.  Returns (z, out, mean, rstd) with z = what ln_bwd needs.
        `fuse_proj_ln` runs it as ONE launch (fs2_gemm_res_ln_fwd: the pre-norm tensor makes one HBM trip instead of three).
        True = every projection: on the wide-tile kernel that is SLOWER than two launches (8.90 vs 8.68 ms per step,
        profiles/r03k_ab_env.log, r03l_ab_env.log: 174 workgroups for 256 CUs).  "stream" = only where the streaming K = 256 kernel
        runs it (the attention sub-layer's fc at the decoder's row count): faster in isolation (27.4 vs 33.9 us without dropout,
        33.9 vs 37.6 us with p = 0.2: the per-element dropout hash sits in an epilogue nothing overlaps) and step-neutral (8.35 vs
        8.37 ms, profiles/r04z_ab_fuse_ln.log), so the default stays two launches - no change of the step's rounding pattern
        for nothing."""
        gamma, beta = self.P[ln + "weight"], self.P[ln + "bias"]
        kw = _seed_kw("seed_pre", site, seed_dev)
        if self.fuse_proj_ln and lens is not None:
            wf, _, b, _ = W[key]
            r = ops.gemm_res_ln(x, wf, b, res, gamma, beta, lens, self._tmap(lens), B, S, p_pre=p,
                                streaming_only=self.fuse_proj_ln == "stream", **kw)
            if r is not None:
                return r
        y = self._gemm(W, key, x, S, lens=lens)
        out, mean, rstd = ops.ln_fwd(y, res, gamma, beta, lens, B, S, p_pre=p, **kw)
        return y, out, mean, rstd

    def _dgemm(self, W, key, dy, S, taps=1, pad=0, act=ACT_NONE, res=None, lens=None):
        """data gradient through the contraction `key` (tap-flipped pack; pad' = (k-1) - pad)."""
        _, wd, _, shape = W[key]
        ragged = lens is not None
        if not self._lens_pays(lens, self.gemm_lens_bwd):
            lens = None
        # (few-tile, long-reduction shapes - the encoder's k=9 data gradient, 48 tiles x 144 K-steps - are split by the same
        # tail mechanism: a launch with fewer tiles than CUs is all tail)
        return ops.conv_gemm(dy, wd, None, S, taps=taps, pad=(taps - 1) - pad, act=act, res=res, lens=lens, tmap=self._tmap(lens),
                             tail_ws=self._tail_workspace(dy.device), ragged=ragged)

    def _tail_workspace(self, device):
        """scratch of the persistent kernel's tail split: one per stream the engine launches contractions on (launches that
        share one must not run concurrently)."""
        if device.type != "cuda":
            return None
        k = 1 if self._on_side else 0
        if self._tail_ws[k] is None:
            self._tail_ws[k] = ops.tail_workspace(device)
        return self._tail_ws[k]

    class _Branch:
        """`with engine._branch():` - the enclosed launches AND allocations go to the side stream, which first waits for
        everything queued on the main stream so far.  Used for the variance predictors: their outputs feed only the loss
        (forward) and their input gradients are needed only where the variance adaptor's backward adds them up, so 3 x ~9
        small latency-bound launches leave the critical chain in each direction and run beside the decoder's big kernels.
        Allocations inside belong to the side stream's pool (temporaries are recycled in that stream's order); a tensor that
        crosses back to the main stream is handed over with an event and record_stream."""

        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            e = self.eng
            e._side_stream.wait_stream(e._branch_main)
            self.ctx = torch.cuda.stream(e._side_stream)
            self.ctx.__enter__()
            self.pin = ops.pinned_stream(e._side_stream)
            self.pin.__enter__()
            e._on_side = True
            return self

        def __exit__(self, *exc):
            e = self.eng
            e._on_side = False
            self.pin.__exit__(*exc)
            self.ctx.__exit__(*exc)

    def _branch(self):
        return Engine._Branch(self)

    def _branch_ok(self):
        """side stream available for branch concurrency (creates it)"""
        if not (self.use_side_stream and self.concurrent_branches) or self.device.type != "cuda":
            return False
        if torch.cuda.is_current_stream_capturing() and not self.fork_in_capture:        # one stream inside a capture (see _side_begin)
            return False
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        self._branch_main = torch.cuda.current_stream()
        return True

    def _wgrad(self, gw, gb, dy, x, S, taps=1, pad=0, lens=None):
        """weight (+ bias) gradient of one contraction.  Weight gradients are OFF the critical path of backward (nothing
        downstream reads them until the optimiser), so they are issued on a side HIP stream: they fill the CUs that the
        data-gradient chain's kernels leave idle in their last partial round of workgroups (e.g. the k=9 data gradient runs
        348 one-per-CU workgroups = 1.36 rounds) and overlap the chain's small latency-bound launches."""
        if gw.dim() == 3:
            gw = gw.permute(0, 2, 1)            # tap-major storage of the gradient
        side = self._side
        if side is None:
            ops.conv_wgrad(dy, x, gw, S, taps=taps, pad=pad, lens=lens, dbias=gb, ws_owner=self._wgrad_ws)
            return
        if not self._on_side:
            # dy was produced on the main stream.  NOT inside a branch section: there the launches already run on the side stream,
            # dy was produced on it, and an event recorded on the capture's origin stream from inside the section made a forked
            # hipGraph capture lose the origin stream's edge loss_bwd -> bn_bwd_acc (rounds 3-5's "2.8e-3 off" forked capture:
            # tools/dbg_fork_capture.py poisons every capture-time buffer and names the reader that ran early)
            side.wait_stream(self._main)
        with ops.pinned_stream(side):           # (no allocation happens inside: only the launch needs the side stream)
            self._ln_flush()
            ops.conv_wgrad(dy, x, gw, S, taps=taps, pad=pad, lens=lens, dbias=gb, ws_owner=self._wgrad_ws)
        self._side_keep.append((dy, x))         # the caching allocator must not recycle them before the join

    def _ln_bwd(self, z, dout, gamma, lens, mean, rstd, gw, gb, B, S, **kw):
        """LayerNorm backward; with a side stream the affine-gradient reduction (a 5 us launch nothing on the data-gradient chain
        waits for) is deferred to the next weight-gradient section of that stream."""
        if self._side is None or not self.defer_ln_reduce:
            return ops.ln_bwd(z, dout, gamma, lens, mean, rstd, gw, gb, B, S, **kw)
        d1, d2, ws = ops.ln_bwd(z, dout, gamma, lens, mean, rstd, gw, gb, B, S, defer=True, **kw)
        self._ln_pending.append((ws, z.shape[-1], gw, gb))
        return d1, d2

    def _ln_flush(self):
        """(inside a side-stream section that already waits for the main stream)"""
        for ws, C, gw, gb in self._ln_pending:
            ops.ln_bwd_reduce(ws, C, gw, gb)
            self._side_keep.append(ws)
        self._ln_pending = []

    def _side_begin(self):
        """fork: weight gradients of this backward go to the side stream (FS2_SIDE_STREAM=0 keeps one stream)."""
        self._side_keep = []
        self._ln_pending = []
        self._main = torch.cuda.current_stream()
        if self._pack_pending:                  # data-gradient packs were written on the side stream during forward
            self._main.wait_stream(self._side_stream)
            self._pack_pending = False
        if not self.use_side_stream or (torch.cuda.is_current_stream_capturing() and not self.fork_in_capture):
            # inside a hipGraph capture the step stays on ONE stream whatever use_side_stream says: a forked capture replayed
            # with gradients 2.8e-3 away from the eager steps (round 3, cause not found) and was the slowest variant anyway
            self._side = None
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)      # normal priority; the step runs on a high one
        self._side = self._side_stream

    def _side_join(self):
        """join: everything queued on the side stream happens-before whatever the main stream does next."""
        if self._side is not None:
            if self._ln_pending:
                self._side.wait_stream(self._main)
                with ops.pinned_stream(self._side):
                    self._ln_flush()
            self._main.wait_stream(self._side)
            self._wgrad_ws.release_retired()        # (main-stream allocations from here on are ordered behind the side stream's work)
        self._side_keep = []

    def _fft_fwd(self, W, pre, x, lens, B, S, n_head, p, seed_dev, site, keep):
        ks = self.tc["conv_kernel_size"]
        a, f = pre + "slf_attn.", pre + "pos_ffn."
        sv = _Saved()
        # padded rows are never consumed downstream (keys masked, LN output re-masked): their tiles are skipped
        qkv = self._gemm(W, a + "qkv", x, S, lens=lens)
        ctx, lse = ops.attn_fwd(qkv, lens, B, S, n_head, self.d // n_head)
        y1, h, mean1, rstd1 = self._proj_ln(W, a + "fc", ctx, x, a + "layer_norm.", lens, B, S, p, site, seed_dev)
        hid = self._gemm(W, f + "w_1", h, S, taps=ks[0], pad=(ks[0] - 1) // 2, act=ACT_RELU,
                         lens=lens if ks[1] == 1 else None)
        if ks[1] == 1:
            y2, out, mean2, rstd2 = self._proj_ln(W, f + "w_2", hid, h, f + "layer_norm.", lens, B, S, p, site + 1, seed_dev)
        else:
            y2 = self._gemm(W, f + "w_2", hid, S, taps=ks[1], pad=(ks[1] - 1) // 2)
            out, mean2, rstd2 = ops.ln_fwd(y2, h, self.P[f + "layer_norm.weight"], self.P[f + "layer_norm.bias"], lens, B, S,
                                           p_pre=p, **_seed_kw("seed_pre", site + 1, seed_dev))
        if keep:
            sv.x, sv.qkv, sv.ctx, sv.lse, sv.z1, sv.mean1, sv.rstd1 = x, qkv, ctx, lse, y1, mean1, rstd1
            sv.h, sv.hid, sv.z2, sv.mean2, sv.rstd2 = h, hid, y2, mean2, rstd2
            sv.p, sv.site = p, site
        return out, sv

    def _fft_bwd(self, W, G, pre, sv, dout, lens, B, S, n_head, seed_dev, dout2=None, split_out=False):
        """backward of one FFT block.  The gradient of a sub-layer's input is "data gradient through the sub-layer + the gradient
        that bypassed it" (output + residual, transformer/SubLayers.py:55,91).  `res_in_ln` (default): the two terms travel as a
        PAIR to the LayerNorm backward below, which adds them while it reads its rows (ops.ln_bwd dout2) - the contraction that
        makes the first term then has no residual operand (its epilogue paid 25-45 us per launch for 16 dependent loads behind
        the tile's stores).  dout2: second term of this block's upstream gradient; split_out: return (dx, bypass) instead of their
        sum (the caller hands the pair to the next block)."""
        ks = self.tc["conv_kernel_size"]
        a, f = pre + "slf_attn.", pre + "pos_ffn."
        p = sv.p
        pair = self.res_in_ln
        dz2, dy2 = self._ln_bwd(sv.z2, dout, self.P[f + "layer_norm.weight"], lens, sv.mean2, sv.rstd2,
                              G[f + "layer_norm.weight"], G[f + "layer_norm.bias"], B, S, want_d1=True, want_d2=p > 0,
                              p_pre=p, dout2=dout2, **_seed_kw("seed_pre", sv.site + 1, seed_dev))
        if dy2 is None:
            dy2 = dz2
        # all gradients below are zero on padded rows (ln_bwd zeroes them) -> lens lets every kernel skip those tiles
        l2 = lens if ks[1] == 1 else None
        self._wgrad(G[f + "w_2.weight"], G[f + "w_2.bias"], dy2, sv.hid, S, taps=ks[1], pad=(ks[1] - 1) // 2, lens=lens)
        dhid = self._dgemm(W, f + "w_2", dy2, S, taps=ks[1], pad=(ks[1] - 1) // 2, act=ACT_GATE, res=sv.hid, lens=l2)
        self._wgrad(G[f + "w_1.weight"], G[f + "w_1.bias"], dhid, sv.h, S, taps=ks[0], pad=(ks[0] - 1) // 2, lens=l2)
        dh = self._dgemm(W, f + "w_1", dhid, S, taps=ks[0], pad=(ks[0] - 1) // 2, res=None if pair else dz2, lens=lens)
        dz1, dy1 = self._ln_bwd(sv.z1, dh, self.P[a + "layer_norm.weight"], lens, sv.mean1, sv.rstd1,
                              G[a + "layer_norm.weight"], G[a + "layer_norm.bias"], B, S, want_d1=True, want_d2=p > 0,
                              p_pre=p, dout2=dz2 if pair else None, **_seed_kw("seed_pre", sv.site, seed_dev))
        if dy1 is None:
            dy1 = dz1
        self._wgrad(G[a + "fc.weight"], G[a + "fc.bias"], dy1, sv.ctx, S, lens=lens)
        dctx = self._dgemm(W, a + "fc", dy1, S, lens=lens)
        dqkv = ops.attn_bwd(sv.qkv, sv.ctx, dctx, sv.lse, lens, B, S, n_head, self.d // n_head)
        self._wgrad(G[a + "qkv.weight"], G[a + "qkv.bias"], dqkv, sv.x, S, lens=lens)
        if split_out and pair:
            return self._dgemm(W, a + "qkv", dqkv, S), dz1
        dx = self._dgemm(W, a + "qkv", dqkv, S, res=dz1)
        return (dx, None) if split_out else dx

    def _pred_fwd(self, W, kind, x, lens, B, S, p, seed_dev, site, keep):
        pre = f"variance_adaptor.{kind}_predictor."
        cl = pre + "conv_layer."
        k = self.vp["kernel_size"]
        sv = _Saved()
        c1 = self._gemm(W, cl + "conv1d_1", x, S, taps=k, pad=(k - 1) // 2, act=ACT_RELU)
        n1, m1, r1 = ops.ln_fwd(c1, None, self.P[cl + "layer_norm_1.weight"], self.P[cl + "layer_norm_1.bias"], None, B, S,
                                p_post=p, **_seed_kw("seed_post", site, seed_dev))
        c2 = self._gemm(W, cl + "conv1d_2", n1, S, taps=k, pad=1, act=ACT_RELU)
        n2, m2, r2 = ops.ln_fwd(c2, None, self.P[cl + "layer_norm_2.weight"], self.P[cl + "layer_norm_2.bias"], None, B, S,
                                p_post=p, **_seed_kw("seed_post", site + 1, seed_dev))
        pred = ops.rowdot_fwd(n2, self.P[pre + "linear_layer.weight"], self.P[pre + "linear_layer.bias"], lens, B, S)
        if keep:
            sv.x, sv.c1, sv.m1, sv.r1, sv.n1, sv.c2, sv.m2, sv.r2, sv.n2 = x, c1, m1, r1, n1, c2, m2, r2, n2
            sv.p, sv.site, sv.lens, sv.S = p, site, lens, S
        return pred, sv

    def _pred_bwd(self, W, G, kind, sv, dpred, B, seed_dev, dx_acc):
        """returns dx_acc + d(input) ; dx_acc may be None."""
        pre = f"variance_adaptor.{kind}_predictor."
        cl = pre + "conv_layer."
        k = self.vp["kernel_size"]
        S, p = sv.S, sv.p
        dn2 = ops.rowdot_bwd(sv.n2, self.P[pre + "linear_layer.weight"], dpred, sv.lens, G[pre + "linear_layer.weight"],
                             G[pre + "linear_layer.bias"], B, S)
        _, dc2 = self._ln_bwd(sv.c2, dn2, self.P[cl + "layer_norm_2.weight"], None, sv.m2, sv.r2, G[cl + "layer_norm_2.weight"],
                            G[cl + "layer_norm_2.bias"], B, S, want_d1=False, want_d2=True, p_post=p,
                            relu_bwd=True, **_seed_kw("seed_post", sv.site + 1, seed_dev))
        self._wgrad(G[cl + "conv1d_2.conv.weight"], G[cl + "conv1d_2.conv.bias"], dc2, sv.n1, S, taps=k, pad=1)
        dn1 = self._dgemm(W, cl + "conv1d_2", dc2, S, taps=k, pad=1)
        _, dc1 = self._ln_bwd(sv.c1, dn1, self.P[cl + "layer_norm_1.weight"], None, sv.m1, sv.r1, G[cl + "layer_norm_1.weight"],
                            G[cl + "layer_norm_1.bias"], B, S, want_d1=False, want_d2=True, p_post=p,
                            relu_bwd=True, **_seed_kw("seed_post", sv.site, seed_dev))
        self._wgrad(G[cl + "conv1d_1.conv.weight"], G[cl + "conv1d_1.conv.bias"], dc1, sv.x, S, taps=k, pad=(k - 1) // 2)
        return self._dgemm(W, cl + "conv1d_1", dc1, S, taps=k, pad=(k - 1) // 2, res=dx_acc)

    # ------------------------------------------------------------------ whole-model forward
    def run(self, speakers, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, p_targets, e_targets, d_targets,
            p_control, e_control, d_control):
        training = self.m.training
        need_grad = training and torch.is_grad_enabled()
        st = _Saved()
        st.speakers, st.texts, st.src_lens, st.L = speakers, texts.contiguous(), src_lens, max_src_len
        st.mel_lens, st.max_mel_len = mel_lens, max_mel_len
        st.p_t, st.e_t, st.d_t = p_targets, e_targets, d_targets
        st.ctl = (p_control, e_control, d_control)
        st.training, st.need_grad = training, need_grad
        if need_grad:
            params = [p for _, p in self.m._trainable_in_backward_order()]
            mel, post, p_pred, e_pred, logd = _FS2Function.apply(self, st, *params)
        else:
            (mel, post, p_pred, e_pred, logd), _ = self.forward(st)
        B = texts.shape[0]
        dev = texts.device
        # the valid-position counts the loss normalises by came out of the same launches as the masks: hand them over on the
        # tensor the loss receives (FastSpeech2Loss falls back to computing them when the attribute is absent)
        st.mel_lens_out._fs2_counts = (st.counts, src_lens, max_src_len, st.Tdec)
        return (mel, post, p_pred, e_pred, logd, st.d_rounded, st.src_masks, st.mel_masks, src_lens, st.mel_lens_out)

    def forward(self, st):
        with ops.pinned_stream():
            return self._forward(st)

    def _forward(self, st):
        m, P, cdt, d = self.m, self.P, self.cdt, self.d
        training, keep = st.training, st.need_grad
        W = self.weights(need_dgrad=keep)
        B, L = st.texts.shape
        assert L == st.L, "texts.shape[1] must equal max_src_len"
        sv = _Saved()
        seed_dev = None
        drop = training and not getattr(m, "disable_dropout", False)   # test hook == patching F.dropout in the reference
        if drop:
            if self.device_seed:                    # hipGraph capture: the counter lives on the device and is bumped by a kernel
                ops.bump_counter(self.seed_counter, self._SEED_INC)
                seed_dev = self.seed_counter.clone()
            else:                                   # eager: the same stream of values, carried in the kernel arguments
                self._seed_k += 1
                seed_dev = (self.base_seed + self._seed_k * self._SEED_INC) & 0x7FFFFFFFFFFFFFFF
        p_enc = self.tc["encoder_dropout"] if drop else 0.0
        p_dec = self.tc["decoder_dropout"] if drop else 0.0
        p_vp = self.vp["dropout"] if drop else 0.0
        p_pn = 0.5 if drop else 0.0
        st.counts = torch.empty(2, device=self.device, dtype=torch.float32)
        src_lens32, st.src_masks, src_lens32._fs2_tmap = ops.lens_prep(st.src_lens, B, L, st.counts[0:1])
        hint, self.length_hint = self.length_hint, None     # consumed by this forward
        if hint is not None and any(h is not None and len(h) != B for h in hint):
            hint = None                                      # set for another batch (a validation call in between): never applied to this one
        src_host = hint[0] if (hint and hint[0] is not None) else getattr(st.src_lens, "_fs2_host", None)
        src_lens32._fs2_skip = self._skip_fraction(src_host, L)

        # ---- encoder (transformer/Models.py:73-100)
        n_head = self.tc["encoder_head"]
        x = ops.embed_pe_fwd(st.texts, P["encoder.src_word_emb.weight"], self._pe(P["encoder.position_enc"], L), cdt)
        sv.enc = []
        for i in range(self.tc["encoder_layer"]):
            x, s = self._fft_fwd(W, f"encoder.layer_stack.{i}.", x, src_lens32, B, L, n_head, p_enc, seed_dev, 10 + 2 * i, keep)
            sv.enc.append(s)
        if m.speaker_emb is not None:
            ops.add_rowvec(x, P["speaker_emb.weight"], st.speakers, B, L)

        # ---- variance adaptor (model/modules.py:102-158)
        p_control, e_control, d_control = st.ctl
        pitch_phone = m.pitch_feature_level == "phoneme_level"
        energy_phone = m.energy_feature_level == "phoneme_level"
        p_pred = e_pred = None
        # teacher-forced training step: the predictions feed only the loss, the embeddings come from the targets -> the three
        # predictors run on the side stream beside the rest of the forward pass (joined before the outputs are returned)
        sv.branch = bool(keep and pitch_phone and energy_phone and st.p_t is not None and st.e_t is not None and st.d_t is not None
                         and self._branch_ok())
        if sv.branch:
            with self._branch():
                logd, sv.dur = self._pred_fwd(W, "duration", x, src_lens32, B, L, p_vp, seed_dev, 100, keep)
                p_pred, sv.pitch = self._pred_fwd(W, "pitch", x, src_lens32, B, L, p_vp, seed_dev, 102, keep)
            x, sv.pitch_idx = ops.bucket_embed_add_fwd(x, st.p_t.contiguous().view(-1), 1.0, P["variance_adaptor.pitch_bins"],
                                                       P["variance_adaptor.pitch_embedding.weight"])
            with self._branch():
                e_pred, sv.energy = self._pred_fwd(W, "energy", x, src_lens32, B, L, p_vp, seed_dev, 104, keep)
            x, sv.energy_idx = ops.bucket_embed_add_fwd(x, st.e_t.contiguous().view(-1), 1.0, P["variance_adaptor.energy_bins"],
                                                        P["variance_adaptor.energy_embedding.weight"])
        else:
            logd, sv.dur = self._pred_fwd(W, "duration", x, src_lens32, B, L, p_vp, seed_dev, 100, keep)
        if pitch_phone and not sv.branch:
            p_pred, sv.pitch = self._pred_fwd(W, "pitch", x, src_lens32, B, L, p_vp, seed_dev, 102, keep)
            if st.p_t is None:
                p_pred = p_pred * p_control if p_control != 1.0 else p_pred
            vals = st.p_t if st.p_t is not None else p_pred
            x, sv.pitch_idx = ops.bucket_embed_add_fwd(x, vals.contiguous().view(-1), 1.0, P["variance_adaptor.pitch_bins"],
                                                       P["variance_adaptor.pitch_embedding.weight"])
        if energy_phone and not sv.branch:
            e_pred, sv.energy = self._pred_fwd(W, "energy", x, src_lens32, B, L, p_vp, seed_dev, 104, keep)
            if st.e_t is None:   # NB reference passes p_control here (model/modules.py:124)
                e_pred = e_pred * p_control if p_control != 1.0 else e_pred
            vals = st.e_t if st.e_t is not None else e_pred
            x, sv.energy_idx = ops.bucket_embed_add_fwd(x, vals.contiguous().view(-1), 1.0, P["variance_adaptor.energy_bins"],
                                                        P["variance_adaptor.energy_embedding.weight"])
        # ---- length regulator (model/modules.py:128-137,167-194)
        if st.d_t is not None:
            dur = st.d_t.contiguous()
            st.d_rounded = st.d_t
            T = st.max_mel_len
        else:
            dur = ops.duration_round(logd, d_control)
            st.d_rounded = dur
            T = st.max_mel_len
        ml_host = None
        if T is None:
            # inference: output length = longest expanded sequence (host sync, as utils/tools.py:93-94 does).  The whole vector
            # comes back in that one round trip and rides on the returned mel_lens (`_fs2_host`): synth_samples cuts the PCM by it
            # without a second synchronisation
            cum, _, mel_len = ops.lr_index(dur, 1)
            ml_host = mel_len.cpu().numpy()
            T = max(int(ml_host.max()), 1)
        # decoder truncation (transformer/Models.py:145-162)
        Tdec = T if ((not training) and T > self.max_seq_len) else min(T, self.max_seq_len)
        cum, idx, mel_len = ops.lr_index(dur, Tdec)
        st.mel_lens_out, st.Tdec = mel_len, Tdec
        if ml_host is not None:
            mel_len._fs2_host = ml_host
        dec_lens32, st.mel_masks, dec_lens32._fs2_tmap = ops.lens_prep(mel_len, B, Tdec, st.counts[1:2])
        # (training: the frame counts equal the batch's mel_lens - sum of the target durations - whose host copy came with the batch)
        dec_lens32._fs2_skip = self._skip_fraction(((hint[1] if hint and hint[1] is not None else getattr(st.mel_lens, "_fs2_host", None))
                                                    if st.d_t is not None else None), Tdec)
        frame_level = (not pitch_phone) or (not energy_phone)
        pe_dec = self._pe(P["decoder.position_enc"], Tdec)
        sv.x_lr_in = None
        y = ops.lr_gather_fwd(x, idx, None if frame_level else pe_dec, B, L, Tdec)
        if frame_level:
            if not pitch_phone:
                p_pred, sv.pitch = self._pred_fwd(W, "pitch", y, dec_lens32, B, Tdec, p_vp, seed_dev, 102, keep)
                if st.p_t is None:
                    p_pred = p_pred * p_control if p_control != 1.0 else p_pred
                vals = st.p_t[:, :Tdec].contiguous() if st.p_t is not None else p_pred
                y, sv.pitch_idx = ops.bucket_embed_add_fwd(y, vals.view(-1), 1.0, P["variance_adaptor.pitch_bins"],
                                                           P["variance_adaptor.pitch_embedding.weight"])
            if not energy_phone:
                e_pred, sv.energy = self._pred_fwd(W, "energy", y, dec_lens32, B, Tdec, p_vp, seed_dev, 104, keep)
                if st.e_t is None:
                    e_pred = e_pred * p_control if p_control != 1.0 else e_pred
                vals = st.e_t[:, :Tdec].contiguous() if st.e_t is not None else e_pred
                y, sv.energy_idx = ops.bucket_embed_add_fwd(y, vals.view(-1), 1.0, P["variance_adaptor.energy_bins"],
                                                            P["variance_adaptor.energy_embedding.weight"])
            ops.add_pe(y, pe_dec, B, Tdec)

        # ---- decoder (transformer/Models.py:139-171)
        n_head = self.tc["decoder_head"]
        sv.dec = []
        for i in range(self.tc["decoder_layer"]):
            y, s = self._fft_fwd(W, f"decoder.layer_stack.{i}.", y, dec_lens32, B, Tdec, n_head, p_dec, seed_dev, 30 + 2 * i, keep)
            sv.dec.append(s)

        # ---- mel linear + postnet (model/fastspeech2.py:95-97, transformer/Layers.py:129-137)
        mel = self._gemm(W, "mel_linear", y, Tdec)
        h = mel
        sv.pn = []
        for i in range(5):
            pre = f"postnet.convolutions.{i}."
            c = self._gemm(W, pre + "0.conv", h, Tdec, taps=5, pad=2)
            act = ACT_TANH if i < 4 else ACT_NONE
            res = mel if i == 4 else None
            if training:
                h_out, mean_rstd = ops.bn_train_fwd(c, P[pre + "1.weight"], P[pre + "1.bias"], self.Bf[pre + "1.running_mean"],
                                                    self.Bf[pre + "1.running_var"], act, p_pn, _seed_pair(200 + i, seed_dev)[0], res=res,
                                                    seed_dev=_seed_pair(200 + i, seed_dev)[1], ws=self._bn_workspace(c.shape[1])[0],
                                                    num_batches_tracked=self.Bf[pre + "1.num_batches_tracked"])
            else:
                mean_rstd = self._bn_eval_stats(pre)
                h_out = torch.empty_like(c)
                ops._lib.call("fs2_bn_apply", c.data_ptr(), mean_rstd.data_ptr(), P[pre + "1.weight"].data_ptr(),
                              P[pre + "1.bias"].data_ptr(), res.data_ptr() if res is not None else None, h_out.data_ptr(),
                              c.shape[0], c.shape[1], act, 0.0, 0, None, ops.dt(c), ops._stream())
            if keep:
                s = _Saved()
                s.x, s.c, s.mean_rstd, s.act = h, c, mean_rstd, act
                sv.pn.append(s)
            h = h_out
        post = h

        if sv.branch:                               # the predictions (side stream) are outputs
            self._branch_main.wait_stream(self._side_stream)
            self._pack_pending = False              # (that join covers the data-gradient packs too)
        n_mel = mel.shape[1]
        mel_o = (ops.cast(mel, torch.float32) if cdt != torch.float32 else mel).view(B, Tdec, n_mel)
        post_o = (ops.cast(post, torch.float32) if cdt != torch.float32 else post).view(B, Tdec, n_mel)
        if keep:
            sv.W, sv.B, sv.L, sv.T = W, B, L, Tdec
            sv.src_lens32, sv.dec_lens32, sv.cum, sv.seed_dev = src_lens32, dec_lens32, cum, seed_dev
            sv.dec_out, sv.texts, sv.speakers = y, st.texts, st.speakers
            sv.pitch_phone, sv.energy_phone = pitch_phone, energy_phone
            sv.p_pn = p_pn
        return (mel_o, post_o, p_pred, e_pred, logd), sv

    # ------------------------------------------------------------------ whole-model backward
    def backward(self, sv, dmel, dpost, dp, de, dlogd):
        with ops.pinned_stream():
            return self._backward(sv, dmel, dpost, dp, de, dlogd)

    def _backward(self, sv, dmel, dpost, dp, de, dlogd):
        m, P, cdt = self.m, self.P, self.cdt
        G = self._grads()
        if any(p.grad is None for p in (P["mel_linear.weight"],)):
            m._flat_grad.zero_()
            m.attach_grads()
        W, B, L, T, seed_dev = sv.W, sv.B, sv.L, sv.T, sv.seed_dev
        n_mel = P["mel_linear.weight"].shape[0]
        self._side_begin()

        def to_c(t, shape):
            if t is None:
                return None
            t = t.contiguous().view(shape)
            return ops.cast(t, cdt) if t.dtype != cdt else t

        # variance predictors' backward: beside the PostNet / decoder backward, on the side stream (first in its queue).
        # d(x3) -> [energy embedding] -> + energy predictor -> [pitch embedding] -> + pitch and duration predictors
        ev_e = ev_pd = dv_e = dv_pd = None
        if sv.branch and self._side is not None:
            self._branch_main = self._main
            with self._branch():
                if de is not None:
                    dv_e = self._pred_bwd(W, G, "energy", sv.energy, de.contiguous(), B, seed_dev, None)
                ev_e = self._side_stream.record_event()
                if dp is not None:
                    dv_pd = self._pred_bwd(W, G, "pitch", sv.pitch, dp.contiguous(), B, seed_dev, None)
                if dlogd is not None:
                    dv_pd = self._pred_bwd(W, G, "duration", sv.dur, dlogd.contiguous(), B, seed_dev, dv_pd)
                ev_pd = self._side_stream.record_event()
            for t in (dv_e, dv_pd):
                if t is not None:
                    t.record_stream(self._main)
        dmel = to_c(dmel, (B * T, n_mel))
        dpost = to_c(dpost, (B * T, n_mel))
        zeros = None
        if dmel is None or dpost is None:
            zeros = torch.zeros(B * T, n_mel, device=self.device, dtype=cdt)
        dmel = dmel if dmel is not None else zeros
        dpost = dpost if dpost is not None else zeros

        # ---- postnet backward
        g = dpost
        for i in reversed(range(5)):
            pre = f"postnet.convolutions.{i}."
            s = sv.pn[i]
            dc = ops.bn_bwd_acc(s.c, g, s.mean_rstd, P[pre + "1.weight"], P[pre + "1.bias"], s.act, sv.p_pn,
                                _seed_pair(200 + i, seed_dev)[0], self._bn_workspace(s.c.shape[1])[1], G[pre + "1.weight"], G[pre + "1.bias"],
                                seed_dev=_seed_pair(200 + i, seed_dev)[1])
            self._wgrad(G[pre + "0.conv.weight"], G[pre + "0.conv.bias"], dc, s.x, T, taps=5, pad=2)
            g = self._dgemm(W, pre + "0.conv", dc, T, taps=5, pad=2, res=dpost if i == 0 else None)
        dmel_total = ops.add(g, dmel)
        # ---- mel linear
        self._wgrad(G["mel_linear.weight"], G["mel_linear.bias"], dmel_total, sv.dec_out, T)
        dy = self._dgemm(W, "mel_linear", dmel_total, T)
        # ---- decoder
        n_head = self.tc["decoder_head"]
        va_first = "variance_adaptor.energy_predictor.linear_layer.weight"
        dy2 = None
        for i in reversed(range(self.tc["decoder_layer"])):
            self._ready(f"decoder.layer_stack.{i}.pos_ffn.layer_norm.weight")
            r = self._fft_bwd(W, G, f"decoder.layer_stack.{i}.", sv.dec[i], dy, sv.dec_lens32, B, T, n_head, seed_dev, dout2=dy2,
                              split_out=i > 0)            # (the first block's input gradient leaves the stack: one tensor)
            dy, dy2 = r if i > 0 else (r, None)
        self._ready(va_first)
        # ---- frame-level variance branches + length regulator
        dpp = dp.contiguous() if dp is not None else None
        dee = de.contiguous() if de is not None else None
        if not sv.energy_phone:
            ops.bucket_embed_bwd(sv.energy_idx, dy, G["variance_adaptor.energy_embedding.weight"])
            if dee is not None:
                dy = self._pred_bwd(W, G, "energy", sv.energy, dee, B, seed_dev, dy)
        if not sv.pitch_phone:
            ops.bucket_embed_bwd(sv.pitch_idx, dy, G["variance_adaptor.pitch_embedding.weight"])
            if dpp is not None:
                dy = self._pred_bwd(W, G, "pitch", sv.pitch, dpp, B, seed_dev, dy)
        dx = ops.lr_gather_bwd(dy, sv.cum, B, L, T)
        # ---- phoneme-level variance branches
        if ev_e is not None:                        # the predictors' input gradients were computed on the side stream
            ops.bucket_embed_bwd(sv.energy_idx, dx, G["variance_adaptor.energy_embedding.weight"])
            if dv_e is not None:
                self._main.wait_event(ev_e)
                dx = ops.add(dx, dv_e)
            ops.bucket_embed_bwd(sv.pitch_idx, dx, G["variance_adaptor.pitch_embedding.weight"])
            if dv_pd is not None:
                self._main.wait_event(ev_pd)
                dx = ops.add(dx, dv_pd)
        else:
            if sv.energy_phone:
                ops.bucket_embed_bwd(sv.energy_idx, dx, G["variance_adaptor.energy_embedding.weight"])
                if dee is not None:
                    dx = self._pred_bwd(W, G, "energy", sv.energy, dee, B, seed_dev, dx)
            if sv.pitch_phone:
                ops.bucket_embed_bwd(sv.pitch_idx, dx, G["variance_adaptor.pitch_embedding.weight"])
                if dpp is not None:
                    dx = self._pred_bwd(W, G, "pitch", sv.pitch, dpp, B, seed_dev, dx)
            if dlogd is not None:
                dx = self._pred_bwd(W, G, "duration", sv.dur, dlogd.contiguous(), B, seed_dev, dx)
        if m.speaker_emb is not None:
            ops.rowvec_bwd(dx, G["speaker_emb.weight"], sv.speakers, B, L)
        # ---- encoder
        n_head = self.tc["encoder_head"]
        dx2 = None
        for i in reversed(range(self.tc["encoder_layer"])):
            self._ready(f"encoder.layer_stack.{i}.pos_ffn.layer_norm.weight")
            r = self._fft_bwd(W, G, f"encoder.layer_stack.{i}.", sv.enc[i], dx, sv.src_lens32, B, L, n_head, seed_dev, dout2=dx2,
                              split_out=i > 0)
            dx, dx2 = r if i > 0 else (r, None)
        self._ready("encoder.src_word_emb.weight")
        ops.embed_bwd(sv.texts, dx, G["encoder.src_word_emb.weight"], pad_idx=0)
        self._side_join()
        self._side = None
        self._ready(None)


class _FS2Function(torch.autograd.Function):
    """The whole acoustic model as one autograd node (parameters are inputs only so that outputs require grad;
    their gradients are written directly into the flat gradient buffer by Engine.backward)."""

    @staticmethod
    def forward(ctx, engine, st, *params):
        outs, sv = engine.forward(st)
        ctx.engine, ctx.sv, ctx.n = engine, sv, len(params)
        return outs

    @staticmethod
    def backward(ctx, dmel, dpost, dp, de, dlogd):
        sv, ctx.sv = ctx.sv, None            # release the saved activations with this backward, not with the next forward
        if sv is None:
            raise RuntimeError("FastSpeech2 backward called twice on the same forward (saved activations were released)")
        ctx.engine.backward(sv, dmel, dpost, dp, de, dlogd)
        return (None, None) + (None,) * ctx.n
