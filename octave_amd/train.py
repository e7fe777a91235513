"""The training step of the hot path (a17: absent from the reference, whose OctaScribbleNet.forward
raises -- models/octa.py:59-60; defined in SURVEY.md 3.5):

    att, agg, _ = net.segmentor(x);  p = softmax(agg, 1)
    L_seg = WPCE(p, ys) [+ Dice(p, ys)] + kl_w * InterlayerDivergence([p, *att]) + adv_w * LSGen(D(att))
    L_seg.backward(); Adam(segmentor)
    L_d = LSDisc(D(real_pyramid), D([a.detach() for a in att])); L_d.backward(); Adam(discriminator)

Data parallel: one process per GPU, gradients live in two flat fp32 arenas (segmentor,
discriminator) that are all-reduced over RCCL (torch.distributed backend "nccl" on ROCm) in BUCKETS
laid out in reverse execution order, and then consumed by ONE fused Adam launch each.  BatchNorm
statistics and the WPCE class weights are per-replica, like DistributedDataParallel on the reference
would be; parameters and buffers are broadcast from rank 0 at construction.
"""
import collections
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import functional as F_
from ._lib import lib
from .layers import defer_bn_counters, flush_bn_counters


def overlapping_stream(ref: Optional[torch.cuda.Stream] = None, tries: int = 6, also: Sequence[torch.cuda.Stream] = ()) -> torch.cuda.Stream:
    """A new stream whose kernels really run BESIDE `ref`'s (default: the current stream).  HIP spreads its streams over a few hardware
    queues (four by default) in creation order, and two streams that share a queue are served in order: a side stream drawn blindly
    serialises with the main stream one time in four (measured on the captured B = 16, 400 x 400 step: 24.8 ms, and 25.5 ms for every fourth
    TrainStep built in one process -- the discriminator's graph then waits behind the backward pass; profiles/r05_stream_probe.txt).
    Probe: one spin kernel on each stream at the same time must take about as long as one alone; the first candidate that passes is
    returned, the last one (with a warning) if none does.  `also`: further streams the new one must run beside (the discriminator's stream and
    the gradient-exchange stream at N > 1 must not share a queue either).  OCTA_STREAM_PROBE=0 returns a plain new stream."""
    if os.environ.get("OCTA_STREAM_PROBE", "1") == "0" or not hasattr(torch.cuda, "_sleep") or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream()
    ref = ref if ref is not None else torch.cuda.current_stream()

    def timed(cycles: int, other: Optional[torch.cuda.Stream], ref: torch.cuda.Stream = ref) -> float:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(ref):
            e0.record(ref)
            if other is not None:
                other.wait_event(e0)
            torch.cuda._sleep(cycles)
            if other is not None:
                with torch.cuda.stream(other):
                    torch.cuda._sleep(cycles)
                    done = torch.cuda.Event()
                    done.record(other)
                ref.wait_event(done)
            e1.record(ref)
        e1.synchronize()
        return e0.elapsed_time(e1)

    cycles = 200_000
    timed(cycles, None)
    t1 = min(timed(cycles, None) for _ in range(2))
    if t1 < 0.2:                                   # aim at ~0.5 ms per spin, whatever clock `_sleep` counts in
        cycles = int(cycles * min(0.5 / max(t1, 1e-3), 50.0))
        t1 = min(timed(cycles, None) for _ in range(2))
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream()
        t2 = max(min(timed(cycles, cand, r_) for _ in range(2)) for r_ in (ref, *also))
        if os.environ.get("OCTA_STREAM_PROBE") == "2":
            print(f"overlapping_stream: spin alone {t1:.3f} ms, beside candidate {_} {t2:.3f} ms -> {'runs beside' if t2 < 1.5 * t1 else 'SERIALISED'}", flush=True)
        if t2 < 1.5 * t1:
            return cand
    import warnings
    warnings.warn(f"overlapping_stream: none of {tries} new streams ran beside the reference stream (spin {t1:.2f} ms alone, {t2:.2f} ms with the last "
                  "candidate): side-stream work will serialise with the main stream")
    return cand


FORCE_ALLREDUCE = os.environ.get("OCTA_DIST_ALWAYS") == "1"     # exercise the RCCL path with a single rank (tests)

# top-level modules of ResnestUNet in the order their parameter gradients COMPLETE in the backward pass; each entry is
# (stage mark that fires when the group is complete, module names).  The flat arena is laid out in this order, so a gradient
# bucket is one contiguous slice and can be all-reduced as soon as its mark has fired (or, graph mode, in this order afterwards).
SEG_GRAD_ORDER: List[Tuple[str, Tuple[str, ...]]] = [
    ("decoder_0", ("fc", "aag_0", "decoder_0")),
    ("decoder_1", ("upsampling_0", "aag_1", "decoder_1")),
    ("decoder_2", ("upsampling_1", "aag_2", "decoder_2")),
    ("decoder_3", ("upsampling_2", "aag_3", "decoder_3")),
    ("decoder_4", ("upsampling_3", "aag_4", "decoder_4")),
    ("encoder_4", ("upsampling_4", "encoder_4")),
    ("encoder_3", ("encoder_3",)),
    ("encoder_2", ("encoder_2",)),
    ("encoder_1", ("encoder_1",)),
    ("end", ("encoder_0_1_2",)),
]


def _dist_on(world: int) -> bool:
    return world > 1 or (FORCE_ALLREDUCE and dist.is_available() and dist.is_initialized())


class FlatArena:
    """Flat fp32 storage for the grad-bearing parameters of a module, their gradients and the Adam
    moments.  Parameters keep their logical shapes AND strides (channels-last conv weights stay so).
    `named_params`: (name, parameter) pairs in arena order; `groups`: optional list of (tag, [names]) -- consecutive
    parameters of one group form one gradient bucket."""

    def __init__(self, named_params: Sequence[Tuple[str, nn.Parameter]], groups: Optional[List[Tuple[str, List[str]]]] = None,
                 min_bucket: int = 4 << 20):
        named_params = [np_ if isinstance(np_, tuple) else (f"p{i}", np_) for i, np_ in enumerate(named_params)]   # plain parameters are fine too
        named = [(n, p) for n, p in named_params if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            n = (n + 3) // 4 * 4            # 16-byte aligned slots
            offs.append(n)
            n += p.numel()
        self.numel = (n + 3) // 4 * 4
        self.offsets = dict(zip(self.names, offs))
        self.p = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense:
                    p.data = p.data.contiguous()
                view = self.p.as_strided(p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g.as_strided(p.shape, p.stride(), o)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # updates APPLIED so far (Adam's t - 1), kept on the device
        # gradient buckets: (tag, lo, hi) contiguous element ranges in arena order; small groups are merged into the NEXT one
        # (a bucket may only be reduced once every group in it is complete, i.e. when its LAST group's mark has fired)
        self.buckets: List[Tuple[str, int, int]] = []
        if groups:
            end_of = {}
            for name, p, o in zip(self.names, self.params, offs):
                end_of[name] = (o + p.numel() + 3) // 4 * 4
            lo = 0
            for tag, names in groups:
                present = [nm for nm in names if nm in end_of]
                if not present:
                    continue
                hi = max(end_of[nm] for nm in present)
                if hi - lo >= min_bucket or tag == groups[-1][0]:
                    self.buckets.append((tag, lo, hi))
                    lo = hi
            if not self.buckets or self.buckets[-1][2] != self.numel:
                last_lo = self.buckets[-1][2] if self.buckets else 0
                if self.buckets and self.buckets[-1][0] == groups[-1][0]:
                    t, l, _ = self.buckets[-1]
                    self.buckets[-1] = (t, l, self.numel)
                else:
                    self.buckets.append((groups[-1][0], last_lo, self.numel))
        else:
            self.buckets = [("end", 0, self.numel)]
        self._comm_bufs: Dict[int, Tensor] = {}

    @property
    def step_count(self) -> int:
        """Number of optimiser updates applied so far (synchronises: the counter lives on the device, where a step skipped
        for a non-finite gradient does not advance it)."""
        return int(self.step_dev.item())

    @step_count.setter
    def step_count(self, n: int):
        self.step_dev.fill_(int(n))

    # ------------------------------------------------------------------ optimiser state (checkpoint / resume)
    def state_dict(self) -> Dict:
        """Adam moments keyed by parameter name (+ the step counter): round-trips with net.state_dict()."""
        out = {"step": self.step_count, "exp_avg": {}, "exp_avg_sq": {}}
        for n, p in zip(self.names, self.params):
            o = self.offsets[n]
            out["exp_avg"][n] = self.m.as_strided(p.shape, p.stride(), o).detach().clone().contiguous()
            out["exp_avg_sq"][n] = self.v.as_strided(p.shape, p.stride(), o).detach().clone().contiguous()
        return out

    def load_state_dict(self, sd: Dict):
        with torch.no_grad():
            for n, p in zip(self.names, self.params):
                o = self.offsets[n]
                self.m.as_strided(p.shape, p.stride(), o).copy_(sd["exp_avg"][n])
                self.v.as_strided(p.shape, p.stride(), o).copy_(sd["exp_avg_sq"][n])
        self.step_count = int(sd["step"])

    def zero_grad(self):
        self.g.zero_()

    def param_range(self, params: Sequence[nn.Parameter]) -> List[Tuple[int, int]]:
        """Merged element ranges covering `params` (Adam on a subset: the discriminator head of the current resolution)."""
        ids = {id(p) for p in params}
        rs = sorted((self.offsets[n], (self.offsets[n] + p.numel() + 3) // 4 * 4) for n, p in zip(self.names, self.params) if id(p) in ids)
        out: List[Tuple[int, int]] = []
        for lo, hi in rs:
            if out and lo <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], hi))
            else:
                out.append((lo, hi))
        return out

    # ------------------------------------------------------------------ gradient exchange
    def _reduce_range(self, lo: int, hi: int, comm_dtype):
        sl = self.g[lo:hi]
        if comm_dtype is None or comm_dtype == torch.float32:
            dist.all_reduce(sl, op=dist.ReduceOp.SUM)
            return
        # reduced-precision exchange: half the xGMI bytes (143 MB instead of 286 MB for the segmentor), one rounding per gradient
        buf = self._comm_bufs.get(lo)
        if buf is None or buf.numel() != hi - lo or buf.dtype != comm_dtype:
            buf = torch.empty(hi - lo, dtype=comm_dtype, device=sl.device)
            self._comm_bufs[lo] = buf
        if not sl.is_cuda:
            # host tensors (the gloo tests of the bucket schedule): plain tensor casts, no kernel of the library involved
            buf.copy_(sl)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            sl.copy_(buf)
            return
        st = torch.cuda.current_stream().cuda_stream
        L = lib()
        L.octa_cast(sl.data_ptr(), 0, buf.data_ptr(), F_._dt(comm_dtype), hi - lo, st)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        L.octa_cast(buf.data_ptr(), F_._dt(comm_dtype), sl.data_ptr(), 0, hi - lo, st)

    def all_reduce(self, world: int, comm_dtype=None):
        if _dist_on(world):
            for _, lo, hi in self.buckets:
                self._reduce_range(lo, hi, comm_dtype)

    def _timed_reduce(self, idx: int, lo: int, hi: int, comm_dtype):
        """_reduce_range on the current (= comm) stream; with `self.diag` set (a list) the call is bracketed by two events:
        (bucket, handed to RCCL, complete) -- the self-diagnosis of a multi-GPU run (bench.py "dist")."""
        diag = getattr(self, "diag", None)
        if diag is None or not self.g.is_cuda:
            self._reduce_range(lo, hi, comm_dtype)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._reduce_range(lo, hi, comm_dtype)
        e1.record()
        diag.append((idx, e0, e1))

    def all_reduce_bucket_async(self, world: int, comm_stream, idx: int, comm_dtype=None):
        """Start the all-reduce of bucket `idx` on `comm_stream` once the gradients written so far on the current stream are
        complete; all_reduce_end() joins."""
        if not _dist_on(world):
            return
        _, lo, hi = self.buckets[idx]
        if comm_stream is None:                 # host tensors: no streams
            self._reduce_range(lo, hi, comm_dtype)
            return
        comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm_stream):
            self._timed_reduce(idx, lo, hi, comm_dtype)

    def all_reduce_begin(self, world: int, comm_stream, comm_dtype=None, skip: Sequence[int] = ()):
        """Start every bucket not yet started (in arena = completion order) on `comm_stream`; the caller overlaps independent
        work and calls all_reduce_end() before consuming the gradients."""
        if not _dist_on(world):
            return
        if comm_stream is None:                 # host tensors: no streams
            for i, (_, lo, hi) in enumerate(self.buckets):
                if i not in skip:
                    self._reduce_range(lo, hi, comm_dtype)
            return
        comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm_stream):
            for i, (_, lo, hi) in enumerate(self.buckets):
                if i not in skip:
                    self._timed_reduce(i, lo, hi, comm_dtype)

    def all_reduce_end(self, world: int, comm_stream):
        if _dist_on(world) and comm_stream is not None:
            torch.cuda.current_stream().wait_stream(comm_stream)

    def broadcast(self, src: int = 0):
        dist.broadcast(self.p, src)
        F_.bump_weight_epoch()

    def adam(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
             ranges: Optional[List[Tuple[int, int]]] = None, ls_state: Optional[Tensor] = None, ls_flag: int = 2, commit: bool = True):
        """One fused Adam launch over the arena (or one per element range in `ranges`).  The bias corrections come from the
        device-side counter `step_dev`, so the launch is hipGraph-capturable as it stands.  `ls_state` (loss scaling): the arena
        is checked for inf / nan first and a flagged update is skipped.  commit=False: the caller ends the step itself with
        octa_step_end (TrainStep does: one launch for both optimisers, the loss scale and the step tick)."""
        if commit and ls_state is not None:       # refuse BEFORE any launch: a late raise would leave moments updated and the counter not
            raise ValueError("FlatArena.adam(ls_state=..., commit=True): the loss-scale state is updated by octa_step_end, pass commit=False")
        st = torch.cuda.current_stream().cuda_stream
        if ls_state is not None:
            # flag the arena (after the all-reduce: every rank sees the same infs) before the update
            lib().octa_nonfinite_flag(self.g.data_ptr(), self.numel, ls_state.data_ptr() + 4 * ls_flag, st)
        for lo, hi in (ranges or [(0, self.numel)]):
            o = lo * 4
            lib().octa_adam_step(self.p.data_ptr() + o, self.g.data_ptr() + o, self.m.data_ptr() + o, self.v.data_ptr() + o, hi - lo, lr, betas[0],
                                 betas[1], eps, weight_decay, 0, grad_scale, self.step_dev.data_ptr(),
                                 None if ls_state is None else ls_state.data_ptr(), ls_flag, st)
        if commit:
            lib().octa_step_end(None, 0, 1.0, 1.0, 1, self.step_dev.data_ptr(), None, None, None, st)
        # one launch refreshes every cached packed conv operand of THIS network; nothing else went stale, so the global
        # weight epoch is left alone (bumping it here made the other network's operands look stale: ~170 redundant
        # per-weight pack launches were captured into every replayed step)
        F_.repack_all(self.params)
        F_.bump_param_epoch()         # caches keyed on parameter values that are NOT refreshed above (folded conv+BN of the eval path)


def mask_pyramid(mask: Tensor, levels: int = 5) -> List[Tensor]:
    """Real multi-scale pyramid for the discriminator: nearest down-sampling by 2**i (contract of
    discriminator/blocks.py:114-125).  On the GPU the levels are written densely by one launch; CPU tensors give views."""
    if mask.is_cuda:
        return F_.mask_pyramid_dense(mask, levels)
    return [mask[:, :, ::2 ** i, ::2 ** i] for i in range(levels)]


class _StepTick:
    """Step counter the device publishes to pinned host memory (octa_host_tick, the last kernel of every step).  Reading it is
    a plain load: the one way found to learn on the host, promptly and without a HIP call, how far the device has got while
    graph launches are queued (event queries came back 70-100 ms late in that situation)."""

    def __init__(self, device):
        self.dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.np = self.host.numpy()
        self.launched = 0            # steps handed to the device so far (eager launches and replays)

    def done(self) -> int:
        return int(self.np[0])

    def wait_until(self, step: int):
        while int(self.np[0]) < step:
            time.sleep(0.00005)


class _PinnedRing:
    """Per-step host -> device updates (Adam bias corrections, the discriminator's CPU random draws) while the GPU still runs
    earlier steps: the host writes slot (step % slots) of a pinned ring, a KERNEL (octa_ring_fetch) copies the slot selected by
    the device's own step counter into the static buffer the step reads.  No hipMemcpyAsync: a pinned H2D copy queued between
    hipGraph launches made the device depend on the host runtime and replayed steps stalled.  A slot is rewritten only after
    the step that read it has finished (step tick); with more slots than the host can run ahead that never waits."""

    def __init__(self, shape, device, tick: "_StepTick", slots=8):
        self.ring = torch.zeros((slots,) + tuple(shape), dtype=torch.float32).pin_memory()
        self.slots = slots
        self.dev = torch.zeros(shape, device=device)
        self.tick = tick
        self.slot_bytes = self.dev.numel() * 4

    def slot(self) -> Tensor:
        n = self.tick.launched + 1                       # the step being launched now
        self.tick.wait_until(n - self.slots)             # the step that read this slot last
        return self.ring[self.tick.launched % self.slots]

    def push(self):
        lib().octa_ring_fetch(F_._p(self.ring), self.slot_bytes, self.slots, F_._p(self.tick.dev), F_._p(self.dev), F_._st())


class _RngFeed:
    """The CPU random draws of the three discriminator calls of a step (InstanceNoise plane then LabelNoise
    uniform, per call, from the GLOBAL CPU generator exactly like the reference: blocks.py:149-170), staged in
    pinned memory and copied to static device buffers so that a captured hipGraph can consume them."""

    def __init__(self, disc, device, tick, calls=3):
        self.has_noise, self.has_label = disc._has_noise, disc._has_label_noise
        self.noise_mod = disc.stack_0[0] if self.has_noise else None
        self.label_mod = disc.out[2] if self.has_label else None
        hw = self.noise_mod.size if self.has_noise else (1, 1)
        self.calls = calls
        self.noise_ring = _PinnedRing((calls,) + tuple(hw), device, tick)
        self.sign_ring = _PinnedRing((calls,), device, tick)
        self.noise_dev, self.sign_dev = self.noise_ring.dev, self.sign_ring.dev
        self.sign_dev.fill_(1.0)
        self.i_noise = self.i_sign = 0

    def refill(self):
        nh, sh = self.noise_ring.slot(), self.sign_ring.slot()
        for c in range(self.calls):
            if self.has_noise:
                self.noise_mod.draw(out=nh[c])       # straight into the pinned slot: no parallel CPU copy (see InstanceNoise.draw)
            sh[c] = self.label_mod.draw_sign() if self.has_label else 1.0
        self.noise_ring.push()
        self.sign_ring.push()
        self.rewind()

    def rewind(self):
        self.i_noise = self.i_sign = 0

    def next_noise(self):
        t = self.noise_dev[self.i_noise % self.calls]
        self.i_noise += 1
        return t

    def next_sign(self):
        t = self.sign_dev[self.i_sign % self.calls:self.i_sign % self.calls + 1]
        self.i_sign += 1
        return t


class _Capture:
    """Static buffers and the four hipGraphs of one input resolution."""
    __slots__ = ("sx", "sys", "sreal", "feed", "out", "att", "graphs", "disc", "seg_graphs")


class TrainStep:
    """One optimiser step of the adversarial (or segmentor-only) loop on fused kernels.

    compute_dtype: torch.bfloat16 (default), torch.float16 (BASELINE config 5: pass loss_scale, e.g. 1024 -- the losses
    themselves are always accumulated in fp32) or torch.float32 (parity runs).
    extra_discriminators: {image size: DiscriminatorBlock} for mixed-resolution training (config 5).  The reference's
    discriminator is tied to ONE resolution (its full-extent head conv and its InstanceNoise plane, blocks.py:42,68-72), so a
    second resolution needs a second block; build it with `DiscriminatorBlock.share_body_with(net.discriminator)` so that
    only the head differs.  A step picks the block matching its input size and updates only that block's head."""

    def __init__(self, net: nn.Module, lr: float = 1e-4, lr_disc: Optional[float] = None, betas=(0.9, 0.999), compute_dtype=torch.bfloat16,
                 adversarial: bool = True, use_dice: bool = True, kl_weight: float = 0.1, adv_weight: float = 0.1,
                 loss_scale=1.0, grad_comm_dtype=None, extra_discriminators: Optional[Dict[int, nn.Module]] = None,
                 overlap_backward: Optional[bool] = None, loss_scale_growth: float = 2.0, loss_scale_backoff: float = 0.5,
                 loss_scale_interval: int = 2000):
        self.net = net
        self.seg, self.disc = net.segmentor, getattr(net, "discriminator", None)
        self.adversarial = adversarial and self.disc is not None
        self.use_dice, self.kl_weight, self.adv_weight = use_dice, kl_weight, adv_weight
        self.lr, self.lr_disc, self.betas = lr, lr_disc or lr, betas
        # loss_scale: a float (static), or "dynamic" (fp16): the scale lives in device memory (8 floats, octa_hip.h), starts at
        # 65536, is halved when an optimiser finds a non-finite gradient (that optimiser skips its update) and doubled after
        # `loss_scale_interval` clean steps - all by kernels inside the step, so it survives hipGraph capture with no host sync
        # A STATIC scale in fp16 uses the same device-side state with growth = backoff = 1: the scale never moves, but an
        # update whose gradients hold an inf / nan is still skipped instead of writing NaN into the weights.
        self.dynamic_scale = isinstance(loss_scale, str)
        if self.dynamic_scale and loss_scale != "dynamic":
            raise ValueError(f"loss_scale must be a number or 'dynamic', got {loss_scale!r}")
        self.device_scale = self.dynamic_scale or compute_dtype == torch.float16
        self.loss_scale = 1.0 if self.device_scale else float(loss_scale)
        self.ls_cfg = (float(loss_scale_growth), float(loss_scale_backoff), int(loss_scale_interval)) if self.dynamic_scale else (1.0, 1.0, 1)
        self.ls_state = None
        if self.device_scale:
            self.ls_state = torch.zeros(8, dtype=torch.float32, device=next(net.parameters()).device)
            self.ls_state[0] = 65536.0 if self.dynamic_scale else float(loss_scale)
        self.grad_comm_dtype = grad_comm_dtype
        self.seg.compute_dtype = compute_dtype
        self.discs: Dict[Optional[int], nn.Module] = {}
        if self.disc is not None:
            self.disc.compute_dtype = compute_dtype
            self.discs[getattr(self.disc, "input_hw", (None, None))[0]] = self.disc
            for hw, d in (extra_discriminators or {}).items():
                d.compute_dtype = compute_dtype
                self.discs[int(hw)] = d
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # linear_head_* never receive a gradient on this path: keep them out of the arena
        named = [(n, p) for n, p in self.seg.named_parameters() if not n.startswith("linear_head_")]
        rank = {m: i for i, (_, mods) in enumerate(SEG_GRAD_ORDER) for m in mods}
        named.sort(key=lambda np_: rank.get(np_[0].split(".", 1)[0], len(rank)))       # stable: registration order inside a module
        groups = [(tag, [n for n, _ in named if n.split(".", 1)[0] in mods]) for tag, mods in SEG_GRAD_ORDER]
        self.seg_arena = FlatArena(named, groups, min_bucket=int(os.environ.get("OCTA_MIN_BUCKET", str(4 << 20))))
        self.disc_arena = None
        if self.adversarial:
            seen, dn = set(), []
            for hw, d in self.discs.items():
                for n, p in d.named_parameters():
                    if id(p) not in seen:
                        seen.add(id(p))
                        dn.append((n if d is self.disc else f"hw{hw}.{n}", p))
            self.disc_arena = FlatArena(dn)
            self._disc_ranges = {id(d): (self.disc_arena.param_range(list(d.parameters())) if len(self.discs) > 1 else None) for d in self.discs.values()}
        if _dist_on(self.world):
            self._broadcast_state()
        F_.set_grad_sink(True)
        F_.defer_wgrads(True)        # weight gradients are queued per stage and run as batched launches (functional.flush_wgrads)
        defer_bn_counters(True)
        self._caps: Dict[int, _Capture] = {}
        self._comm_stream = None
        self._probed_for = None
        self._disc_stream = None
        # 64 MB of fp32 scratch for the 8-wave conv kernel's tail split (octa_conv_desc.ws): the 25 x 25 / 50 x 50 decoder
        # layers launch 316 / 626 tiles on 256 CUs
        self._sk_ws = torch.empty(16 << 20, dtype=torch.float32, device=next(net.parameters()).device) if os.environ.get("OCTA_SPLITK", "1") != "0" else None
        # scratch of the partial-store weight gradients (octa_conv2d_wgrad_batch's ws): one per phase, because the discriminator's step
        # replays on a second stream beside the segmentor's backward pass.  The largest batch of single-problem jobs (the four halo-kernel layers of decoder_0 / decoder_1: 38 MB each) needs ~170 MB.  (The batched 8-wave kernels can use it too, octa_tuning_set(4, 1)
        # with OCTA_WGRAD_FOLD_MB=1024: measured +0.1 ms per step, their atomics are not contended; off.)
        fold_on = os.environ.get("OCTA_WGRAD_FOLD", "1") != "0"
        dev0 = next(net.parameters()).device
        self._fold_ws = torch.empty(int(os.environ.get("OCTA_WGRAD_FOLD_MB", "192")) << 18, dtype=torch.float32, device=dev0) if fold_on else None
        self._fold_ws_disc = torch.empty(32 << 20, dtype=torch.float32, device=dev0) if (fold_on and self.adversarial) else None
        self._disc_slab = F_._ZeroSlab(2 << 20)
        # replay mode: the discriminator's own step (it needs the attention maps, not the segmentor's gradients) on a second stream
        # beside the segmentor's backward pass -- a chain of ~200 launch-latency-bound kernels that otherwise costs 3 ms on its own
        self.concurrent_disc = os.environ.get("OCTA_CONCURRENT_DISC", "1") != "0"
        # ... launched behind the backward piece that ends at this stage mark (None: right behind the forward graph).  Beside the
        # low-resolution half of the backward pass (MFMA-bound GEMMs on small grids, latency-bound 13 x 13 / 25 x 25 kernels) the
        # discriminator's HBM-bound kernels cost less than beside the full-resolution decoder stages: swept in situ, forward /
        # decoder_3 / decoder_4 / encoder_4 / encoder_3 = 28.27 / 28.03 / 27.90 / 28.03 / 28.12 ms per step
        self._disc_after_tag = os.environ.get("OCTA_DISC_AFTER", "decoder_4")
        # eager launches with more than one rank: start each gradient bucket's all-reduce from the stage mark that completes it,
        # i.e. overlapped with the REST of the backward pass (BASELINE config 4).  Captured graphs hold no collective: there the
        # buckets are issued, in the same order, right after the segmentor graph and overlap the discriminator step instead.
        self.overlap_backward = (_dist_on(self.world) or os.environ.get("OCTA_SPLIT_BACKWARD") == "1") if overlap_backward is None else bool(overlap_backward)
        # ... and with overlap_backward the SEGMENTOR graph is captured in pieces, cut at the stage marks that complete a gradient
        # bucket (capture()): a replay then starts each bucket's all-reduce right behind the piece that produced it, overlapped
        # with the rest of the backward pass exactly like the eager path
        self._started: List[int] = []
        self._tag_to_bucket = {tag: i for i, (tag, _, _) in enumerate(self.seg_arena.buckets)}
        self._disc_after = self._tag_to_bucket.get(self._disc_after_tag) if (self.concurrent_disc and self.adversarial) else None
        if self.concurrent_disc and self.adversarial and self._disc_after is None and self._disc_after_tag not in ("", "forward", "fwd"):
            import warnings
            warnings.warn(f"TrainStep: OCTA_DISC_AFTER={self._disc_after_tag!r} names no gradient bucket (have {sorted(self._tag_to_bucket)}): "
                          f"the discriminator's graph is launched right behind the forward graph")
        if self._disc_after is not None and self._disc_after >= len(self.seg_arena.buckets) - 1:
            self._disc_after = None           # the last bucket ends with the backward pass: nothing left to run beside
        self.launch = "graph"        # after capture(): "graph" replays the hipGraphs, "eager" launches the same step from Python
        self.max_ahead = int(os.environ.get("OCTA_MAX_AHEAD", "0"))      # replayed steps the host may be ahead of the device (0: no pacing; pacing did not cure the stalls)
        self._tick = _StepTick(next(net.parameters()).device)
        self.launch_timing = None

    # ------------------------------------------------------------------ replicas
    def _broadcast_state(self):
        """Every rank starts from rank 0's parameters AND buffers (BatchNorm running statistics, spectral-norm u / v, counters):
        what DistributedDataParallel does at construction.  Without it replicas silently diverge unless every rank seeded
        the same generator before building the network."""
        self.seg_arena.broadcast(0)
        if self.disc_arena is not None:
            self.disc_arena.broadcast(0)
        mods = [self.seg] + list(self.discs.values())
        seen = set()
        for m in mods:
            for b in m.buffers():
                if id(b) in seen or not b.is_cuda:
                    continue
                seen.add(id(b))
                dist.broadcast(b, 0)
        for p in self.seg.parameters():          # the heads that are not in the arena
            if id(p) not in {id(q) for q in self.seg_arena.params}:
                dist.broadcast(p.data, 0)

    def state_dict(self) -> Dict:
        """Optimiser state for checkpoint / resume (the network itself: net.state_dict())."""
        sd = {"segmentor": self.seg_arena.state_dict()}
        if self.disc_arena is not None:
            sd["discriminator"] = self.disc_arena.state_dict()
        if self.ls_state is not None:
            sd["loss_scale_state"] = self.ls_state.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd: Dict):
        self.seg_arena.load_state_dict(sd["segmentor"])
        if self.disc_arena is not None and "discriminator" in sd:
            self.disc_arena.load_state_dict(sd["discriminator"])
        if self.ls_state is not None and "loss_scale_state" in sd:
            self.ls_state.copy_(sd["loss_scale_state"])

    def _pick_disc(self, x: Tensor):
        if not self.adversarial:
            return None
        if len(self.discs) == 1:
            return self.disc
        d = self.discs.get(int(x.shape[-1]))
        if d is None:
            raise ValueError(f"no discriminator was registered for {x.shape[-1]} x {x.shape[-1]} inputs (have {sorted(k for k in self.discs if k)})")
        return d

    # ------------------------------------------------------------------ the phases of a step
    # (split at the two gradient all-reduces so that the collectives stay OUTSIDE any captured graph)
    def _on_mark(self, tag: str):
        i = self._tag_to_bucket.get(tag)
        if i is not None and i not in self._started:
            self.seg_arena.all_reduce_bucket_async(self.world, self._comm(), i, self.grad_comm_dtype)
            self._started.append(i)

    def _phase_segmentor(self, x, ys, out, disc, hooks=False, between=None):
        # the phase's scratch buffers are host-side context (functional._SPLITK_WS / _WGRAD_FOLD_WS) that every conv CALL of the phase
        # carries in its own arguments (octa_conv_desc.ws; the library keeps no pointer): entered for THIS phase only, so that a conv
        # launched between steps -- a validation forward, a second TrainStep -- never shares the partial-tile workspace
        F_.set_splitk_workspace(self._sk_ws)
        F_.set_wgrad_fold_workspace(self._fold_ws)
        try:
            return self._phase_segmentor_body(x, ys, out, disc, hooks, between)
        finally:
            F_.set_splitk_workspace(None)
            F_.set_wgrad_fold_workspace(None)

    def _phase_segmentor_body(self, x, ys, out, disc, hooks, between):
        self.seg_arena.zero_grad()
        F_.ZERO_SLAB.begin(x.device)          # one clear for every small fp32 accumulator of the step
        self._started = []
        hooked = frozen = False
        # ONE try / finally around everything behind begin(): whatever raises (the forward, a loss, the discriminator call, backward),
        # the slab is ended, the discriminator's parameters get their requires_grad back and the mark hooks are cleared -- the next
        # step or a validation forward never meets a half-open slab or a frozen discriminator
        try:
            att, agg, _ = self.seg(x)
            l = F_.wpce_dice(agg, ys, from_logits=True)
            out["wpce"], out["dice"] = l[0].detach(), l[1].detach()
            kl2 = g_adv = None
            if self.adversarial:
                p = F_.class_softmax(agg)
                # every attention map feeds the divergence AND the discriminator: the discriminator's gradient (its backward runs first:
                # the stash nodes are created after the divergence's) is parked and added by the divergence's backward kernels
                hs = [F_.GradHolder() for _ in att] if F_._FUSE_KL_FANOUT else None
                kl2 = F_.interlayer_kl([p, *att], [1] * len(att), holders=[None, *hs] if hs else None)
                # the generator pass only needs dL/d(att) through D: D's own weight gradients of this pass are discarded
                # (zeroed before D's step, here and in the reference), so they are not computed at all
                frozen = True
                for q in self.disc_arena.params:
                    q.requires_grad_(False)
                # (this is also what makes the concurrent discriminator graph safe: with requires_grad off nothing in the segmentor's
                # backward pass writes the discriminator's gradient arena, which that graph zeroes and fills on its own stream)
                g_adv = F_.lsgan_generator(disc([F_.stash_grad(a, h) for a, h in zip(att, hs)] if hs else att))
                assert not any(q.requires_grad for q in self.disc_arena.params)
                out["kl"], out["g_adv"] = kl2[0].detach(), g_adv.detach()
            # ((wpce + dice) + kl_w kl) + adv_w g_adv and its loss-scaled copy: one launch each way (functional.LossCombineFn)
            loss, scaled = F_.loss_combine(l, kl2, g_adv, (1.0, 1.0 if self.use_dice else 0.0, self.kl_weight, 0.0, self.adv_weight),
                                           self.ls_state[0:1] if self.device_scale else None, 1.0 if self.device_scale else self.loss_scale)
            att_out = [a.detach() for a in att]
            if between is not None:
                between(att_out)                  # capture(): the forward graph ends here (the discriminator's step only needs att_out)
            if hooks:
                hooked = True
                if callable(hooks):
                    F_.add_mark_hook(hooks, getattr(hooks, "tags", None) or self._tag_to_bucket.keys())     # capture(): cuts the graph at the bucket-completing marks
                else:
                    F_.add_mark_hook(self._on_mark, self._tag_to_bucket.keys())
            scaled.backward(gradient=F_.one_like_seed(scaled))
            F_.flush_wgrads()
        finally:
            if hooked:
                F_.clear_mark_hooks()
            if frozen:
                for q in self.disc_arena.params:
                    q.requires_grad_(True)
            F_.ZERO_SLAB.end()
        out["loss_seg"] = loss.detach()
        flush_bn_counters()
        return att_out

    def _phase_discriminator(self, att, real_pyramid, out, disc):
        """The discriminator's forward/backward.  It depends on the segmentor step only through the (detached)
        attention maps, not on the reduced segmentor gradients, so it runs WHILE those are being all-reduced."""
        if self.adversarial:
            self.disc_arena.zero_grad()
            # a zero slab of its own (spectral-norm workspaces, weight-gradient targets of the 15-channel convs): in replay mode this
            # phase runs on a second stream beside the segmentor's backward pass, whose slab is still in use
            seg_slab, F_.ZERO_SLAB = F_.ZERO_SLAB, self._disc_slab
            began = False
            try:
                F_.ZERO_SLAB.begin(self.disc_arena.g.device)
                began = True
                F_.set_splitk_workspace(None)          # (same reason: one scratch, two streams)
                F_.set_wgrad_fold_workspace(self._fold_ws_disc)
                d_real = disc(real_pyramid)
                d_fake = disc(att)
                l_d = F_.lsgan_discriminator(d_real, d_fake)
                _, scaled_d = F_.loss_combine(None, None, l_d, (0.0, 0.0, 0.0, 0.0, 1.0), self.ls_state[0:1] if self.device_scale else None,
                                              1.0 if self.device_scale else self.loss_scale)
                scaled_d.backward(gradient=F_.one_like_seed(scaled_d))
                F_.flush_wgrads()
            finally:
                F_.set_wgrad_fold_workspace(None)
                if began:
                    F_.ZERO_SLAB.end()
                F_.ZERO_SLAB = seg_slab
            out["loss_disc"] = l_d.detach()

    def _grad_scale(self) -> float:
        return 1.0 / (self.world * self.loss_scale)

    def _phase_seg_update(self):
        self.seg_arena.adam(self.lr, self.betas, grad_scale=self._grad_scale(), ls_state=self.ls_state, ls_flag=2, commit=False)

    def _phase_finish(self, disc):
        if self.adversarial:
            self.disc_arena.adam(self.lr_disc, self.betas, grad_scale=self._grad_scale(), ranges=self._disc_ranges[id(disc)],
                                 ls_state=self.ls_state, ls_flag=3, commit=False)
        # last kernel of the step: commit Adam's step counters (a flagged optimiser did not step), update the loss scale and
        # publish "step n done" to pinned host memory (paces the replayed path, see __call__)
        g, b, n = self.ls_cfg
        lib().octa_step_end(F_._p(self.ls_state), 2 if self.adversarial else 1, g, b, n, F_._p(self.seg_arena.step_dev),
                            F_._p(self.disc_arena.step_dev) if self.adversarial else None, F_._p(self._tick.dev), F_._p(self._tick.host), F_._st())
        if not torch.cuda.is_current_stream_capturing():
            self._tick.launched += 1           # an eagerly launched step (training or capture warm-up); replays count in __call__

    def _comm(self):
        if self._comm_stream is None:
            self._comm_stream = overlapping_stream() if _dist_on(self.world) else torch.cuda.Stream()
        return self._comm_stream

    def __call__(self, x: Tensor, ys: Tensor, real_pyramid: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
        if self.adversarial and real_pyramid is None:
            raise ValueError("the adversarial step needs the real mask pyramid")
        cap = self._caps.get(int(x.shape[-1]))
        if cap is not None and self.launch == "graph":
            # Optional pacing (OCTA_MAX_AHEAD > 0): at most that many replayed steps queued, measured with the step tick in
            # pinned host memory.  Off by default: on this ROCm stack a loop of replays stalls intermittently either way (the
            # device idles ~3 step times while the host waits in a HIP call, a sleep or a poll), see autotune_launch.
            if self.max_ahead > 0:
                self._tick.wait_until(self._tick.launched + 1 - self.max_ahead)
            out = self._replay(cap, x, ys, real_pyramid)
            self._tick.launched += 1
            return out
        # launch == "eager" (or nothing captured for this size): the step launched kernel by kernel from Python.  Same arenas,
        # step counters and global CPU generator as the replayed path, so the two can be mixed freely.
        out: Dict[str, Tensor] = {}
        disc = self._pick_disc(x)
        feed = getattr(disc, "rng_feed", None) if disc is not None else None
        if feed is not None:
            disc.rng_feed = None          # captured earlier: this path draws the discriminator's noise inline, like the reference
        try:
            att = self._phase_segmentor(x, ys, out, disc, hooks=self.overlap_backward)
            # buckets not started from a stage mark (all of them without overlap): 286 MB over xGMI, hidden behind the D step
            self.seg_arena.all_reduce_begin(self.world, self._comm(), self.grad_comm_dtype, skip=self._started)
            self._phase_discriminator(att, real_pyramid, out, disc)
            self.seg_arena.all_reduce_end(self.world, self._comm())
            self._phase_seg_update()
            if self.adversarial:
                self.disc_arena.all_reduce(self.world)
            self._phase_finish(disc)
        finally:
            if feed is not None:
                disc.rng_feed = feed
        return out

    # ------------------------------------------------------------------ hipGraph capture / replay
    def capture(self, x: Tensor, ys: Tensor, real_pyramid: Optional[Sequence[Tensor]] = None, warmup: int = 2):
        """Capture the step for inputs of x's size into hipGraphs (torch.cuda.CUDAGraph = hipGraph on ROCm: segmentor forward |
        segmentor backward, in pieces when gradients are all-reduced | discriminator step | segmentor Adam + operand repack |
        discriminator Adam) around the two all-reduces.  The discriminator step replays on a second stream beside the backward pass.
        Inputs are copied into static buffers on every call; the CPU random draws of the discriminator are staged through
        _RngFeed; Adam's step counters live on the device.  Call once per input resolution."""
        dev = x.device
        cap = _Capture()
        cap.sx, cap.sys = x.clone(), ys.clone()
        cap.sreal = [r.contiguous().clone() for r in real_pyramid] if real_pyramid is not None else None
        cap.disc = self._pick_disc(x)
        cap.feed = None
        if self.adversarial:
            cap.feed = _RngFeed(cap.disc, dev, self._tick)
            cap.disc.rng_feed = cap.feed
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):             # warm-up off the default stream, as graph capture requires
            F_.set_conv_autotune(True)            # the first warm-up step measures the fwd / dgrad kernel choice per layer shape
            try:
                for _ in range(warmup):
                    self._eager_static(cap)
            finally:
                F_.set_conv_autotune(False)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        cap.out = {}
        # the warm-up steps ended with Adam + repack_all: every cached operand is current, so no per-weight pack launch
        # is captured (a stale cache here would put ~180 redundant pack kernels into every replay)
        if self.adversarial:
            cap.feed.rewind()
        # "thread_local": RCCL's watchdog thread keeps polling the events of the warm-up all-reduces with hipEventQuery while this
        # thread captures; under the default global mode that query is an error and the watchdog aborts the process
        mode = "thread_local"
        pool = next(iter(self._caps.values())).graphs[0].pool() if self._caps else None
        cap.seg_graphs = None
        g1 = self._capture_segmentor_in_pieces(cap, pool)
        g2 = torch.cuda.CUDAGraph()
        # concurrent replay: a memory pool of its own (graphs that share a pool must replay one after the other, in capture order)
        d_pool = {} if (self.concurrent_disc and self.adversarial) else {"pool": g1.pool()}
        with torch.cuda.graph(g2, capture_error_mode=mode, **d_pool):
            self._phase_discriminator(cap.att, cap.sreal, cap.out, cap.disc)
        g2b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2b, pool=g1.pool(), capture_error_mode=mode):
            self._phase_seg_update()
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, pool=g1.pool(), capture_error_mode=mode):
            self._phase_finish(cap.disc)
        cap.graphs = (g1, g2, g2b, g3)
        self._caps[int(x.shape[-1])] = cap
        return self

    def _capture_segmentor_in_pieces(self, cap: "_Capture", pool):
        """The segmentor phase as a CHAIN of hipGraphs cut at the stage marks that complete a gradient bucket (SEG_GRAD_ORDER):
        piece k ends when bucket k's gradients are final, so a replay can hand bucket k to RCCL (side stream) while piece k + 1
        -- the rest of the backward pass -- runs.  The cuts happen inside autograd's backward pass, i.e. on the autograd worker
        thread: capture mode "relaxed" (stream capture may be ended / begun from any thread, and RCCL's watchdog may query its
        events meanwhile).  All pieces allocate from one private pool and are replayed in capture order."""
        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        pieces: List[Tuple[torch.cuda.CUDAGraph, Optional[int]]] = []
        state = {"g": torch.cuda.CUDAGraph()}
        kw = {"capture_error_mode": "relaxed"}

        def next_piece(marker):
            state["g"].capture_end()
            pieces.append((state["g"], marker))
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(pool=pieces[0][0].pool(), **kw)

        def cut(tag):
            i = self._tag_to_bucket.get(tag)
            if i is None or any(b == i for _, b in pieces):
                return
            next_piece(i)

        def forward_done(att):
            cap.att = att
            next_piece("fwd")         # piece 0 = forward pass + losses + the generator pass's discriminator forward
        if not self.overlap_backward and self._disc_after is not None:
            cut.tags = [self.seg_arena.buckets[self._disc_after][0]]       # one cut: where a replay launches the discriminator graph
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if pool is not None:
                state["g"].capture_begin(pool=pool, **kw)
            else:
                state["g"].capture_begin(**kw)
            try:
                cap.att = self._phase_segmentor(cap.sx, cap.sys, cap.out, cap.disc, hooks=cut if (self.overlap_backward or getattr(cut, "tags", None)) else False,
                                                between=forward_done)
            finally:
                state["g"].capture_end()
        pieces.append((state["g"], None))
        cur.wait_stream(side)
        cap.seg_graphs = pieces
        if self._disc_after is not None and not any(b == self._disc_after for _, b in pieces):
            # the stage mark behind which a replay starts the discriminator's graph on its own stream never fired during capture
            # (a network without that stage): without this, _replay would silently fall back to a serial discriminator graph
            import warnings
            warnings.warn(f"TrainStep: the stage mark {self._disc_after_tag!r} (OCTA_DISC_AFTER) did not fire during capture; the discriminator's "
                          f"graph is launched right behind the forward graph instead")
            self._disc_after = None
        return pieces[0][0]

    @property
    def _graphs(self):          # bench.py's roofline leg swaps this to record an eager step
        return self._caps or None

    @_graphs.setter
    def _graphs(self, v):
        self._caps = v or {}

    def _eager_static(self, cap: _Capture):
        out: Dict[str, Tensor] = {}
        if self.adversarial:
            cap.feed.refill()
        att = self._phase_segmentor(cap.sx, cap.sys, out, cap.disc)
        self.seg_arena.all_reduce_begin(self.world, self._comm(), self.grad_comm_dtype)
        self._phase_discriminator(att, cap.sreal, out, cap.disc)
        self.seg_arena.all_reduce_end(self.world, self._comm())
        self._phase_seg_update()
        if self.adversarial:
            self.disc_arena.all_reduce(self.world)
        self._phase_finish(cap.disc)
        return out

    def _load_static(self, cap: _Capture, x, ys, real_pyramid):
        if x is not cap.sx:
            cap.sx.copy_(x, non_blocking=True)
        if ys is not cap.sys:
            cap.sys.copy_(ys, non_blocking=True)
        if cap.sreal is not None and real_pyramid is not None:
            for dst, src in zip(cap.sreal, real_pyramid):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)

    # ------------------------------------------------------------------ launch-path choice
    def _snapshot(self):
        arenas = [a for a in (self.seg_arena, self.disc_arena) if a is not None]
        mods = [self.seg] + list(self.discs.values())
        bufs, seen = [], set()
        for m in mods:
            for b in m.buffers():
                if id(b) not in seen:
                    seen.add(id(b))
                    bufs.append((b, b.detach().clone()))
        ls = None if self.ls_state is None else self.ls_state.clone()
        return ([(a, a.p.clone(), a.m.clone(), a.v.clone(), a.step_count) for a in arenas], bufs, torch.get_rng_state(), ls)

    def _restore(self, snap):
        arenas, bufs, rng, ls = snap
        if ls is not None:
            self.ls_state.copy_(ls)
        with torch.no_grad():
            for a, p, m, v, sc in arenas:
                a.p.copy_(p); a.m.copy_(m); a.v.copy_(v)
                a.step_count = sc
                F_.repack_all(a.params)
            for b, c in bufs:
                b.copy_(c)
        torch.set_rng_state(rng)

    def autotune_launch(self, x, ys, real_pyramid=None, rounds: int = 3, steps: int = 6) -> str:
        """Pick the faster launch path for THIS process on THIS host: hipGraph replay needs almost no host time but pays a few
        microseconds of dependency handling per node and stalls when the box is busy; launching the ~1400 kernels from Python
        costs the host ~33 ms/step.  Both paths run the identical step on the same arenas; `rounds` alternating timings of
        `steps` real steps each, the medians decide.  The parameters, Adam moments and step counters, every module buffer and
        the CPU generator are snapshotted before and restored after, so the call has no training side effect."""
        import time
        cap = self._caps.get(int(x.shape[-1]))
        if cap is None:
            self.launch_timing = None
            return self.launch
        snap = self._snapshot()
        t = {"graph": [], "eager": []}
        for _ in range(rounds):
            for mode in ("graph", "eager"):
                self.launch = mode
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    self(x, ys, real_pyramid)
                torch.cuda.synchronize()
                t[mode].append((time.perf_counter() - t0) / steps)
        med = {m: sorted(v)[len(v) // 2] for m, v in t.items()}
        if _dist_on(self.world):
            # every rank takes the same decision (the slowest rank's timings): the two paths issue the same collectives in the
            # same order, but a job is easier to reason about when all ranks launch the same way
            tt = torch.tensor([med["graph"], med["eager"]], dtype=torch.float64, device=x.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med = {"graph": float(tt[0]), "eager": float(tt[1])}
        # Prefer the replayed path unless eager launches are clearly faster: a replay costs the host < 2 ms per step, Python
        # launches ~30 ms - right at the device time, so they are the first to suffer on a busy host.  (Replays used to
        # stall for ~3 step times at a stretch; the cause was on the CPU side - OpenMP workers spinning after a parallel
        # tensor copy starved the HIP runtime's helper threads - and is gone, see InstanceNoise.draw / DESIGN.md.)
        self.launch = "eager" if med["eager"] < 0.97 * med["graph"] else "graph"
        self.launch_timing = med
        self._restore(snap)
        torch.cuda.synchronize()
        return self.launch

    def _replay(self, cap: _Capture, x, ys, real_pyramid):
        self._load_static(cap, x, ys, real_pyramid)
        if self.adversarial:
            cap.feed.refill()
        g1, g2, g2b, g3 = cap.graphs
        cur = torch.cuda.current_stream()
        if self._probed_for != cur.cuda_stream:
            # the side streams must run beside THIS stream (overlapping_stream): what the eager warm-up steps drew was probed against the capture's
            # side stream, not against the stream the graphs replay on
            self._probed_for = cur.cuda_stream
            self._disc_stream = None
            if _dist_on(self.world):
                self._comm_stream = overlapping_stream(cur)
        comm = self._comm()
        started: List[int] = []
        side_d = self.concurrent_disc and self.adversarial
        d_done = None
        # the segmentor phase in pieces: the forward graph, then the backward pass cut where a gradient bucket completes: bucket k
        # leaves for RCCL as soon as piece k has been enqueued, the next piece (the rest of the backward pass) runs meanwhile
        for g, marker in cap.seg_graphs:
            g.replay()
            if marker == ("fwd" if self._disc_after is None else self._disc_after):
                if side_d:
                    # the discriminator's step beside the backward pass: it reads the attention maps and the spectral-norm state the
                    # forward graph left, and writes only its own gradient arena
                    if self._disc_stream is None:
                        # default priority (a high-priority stream gave the whole gain back: 28.45 -> 29.2 ms), on a hardware queue of its own
                        self._disc_stream = overlapping_stream(cur, also=[comm] if _dist_on(self.world) else ())
                    self._disc_stream.wait_stream(cur)
                    with torch.cuda.stream(self._disc_stream):
                        g2.replay()
                        d_done = torch.cuda.Event()
                        d_done.record(self._disc_stream)
            if marker is not None and marker != "fwd":
                self.seg_arena.all_reduce_bucket_async(self.world, comm, marker, self.grad_comm_dtype)
                started.append(marker)
        # what has not left yet (everything, without the bucket cuts), in completion order on the comm stream
        self.seg_arena.all_reduce_begin(self.world, comm, self.grad_comm_dtype, skip=started)
        seg_done = None
        if _dist_on(self.world):
            seg_done = torch.cuda.Event()
            seg_done.record(comm)
        if d_done is not None:
            cur.wait_event(d_done)
        else:
            g2.replay()
        if self.adversarial:
            # the discriminator's gradients follow the segmentor's on the comm stream and travel while the segmentor's Adam runs
            self.disc_arena.all_reduce_begin(self.world, comm, None)
        if seg_done is not None:
            wd = getattr(self, "_wait_diag", None)
            if wd is not None:                  # diagnosis: how long the main stream sits in front of Adam waiting for the last bucket
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record()
            torch.cuda.current_stream().wait_event(seg_done)
            if wd is not None:
                w1.record()
                wd.append((w0, w1))
        g2b.replay()
        if self.adversarial:
            self.disc_arena.all_reduce_end(self.world, comm)
        g3.replay()
        F_.bump_weight_epoch()      # the weights moved behind the pack cache's back
        F_.bump_param_epoch()
        return cap.out

    def comm_diagnosis(self, x, ys, real_pyramid=None, steps: int = 5) -> Dict:
        """Self-diagnosis of the gradient exchange (run AFTER a timed region, it records events around every collective):
        `steps` further steps with, per gradient bucket, the time from "handed to RCCL on the comm stream" to "complete", and the
        time the main stream waits in front of the segmentor's Adam for the last bucket (the EXPOSED part of the exchange).
        Also counts the ranks the communicator really spans (an all-reduce of ones).  Every rank must call it."""
        if not _dist_on(self.world):
            return {"world_size": self.world, "note": "single process: no collective is issued"}
        dev = self.seg_arena.g.device
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        self.seg_arena.diag, self._wait_diag = [], []
        if self.disc_arena is not None:
            self.disc_arena.diag = []
        try:
            for _ in range(steps):
                self(x, ys, real_pyramid)
            torch.cuda.synchronize()
            per = collections.defaultdict(list)
            for idx, e0, e1 in self.seg_arena.diag:
                per[idx].append(e0.elapsed_time(e1))
            waits = [w0.elapsed_time(w1) for w0, w1 in self._wait_diag]
            dsc = [e0.elapsed_time(e1) for _, e0, e1 in (self.disc_arena.diag if self.disc_arena is not None else [])]
        finally:
            self.seg_arena.diag = self._wait_diag = None
            if self.disc_arena is not None:
                self.disc_arena.diag = None
        es = 4 if self.grad_comm_dtype in (None, torch.float32) else 2
        med = lambda v: sorted(v)[len(v) // 2] if v else None     # noqa: E731
        return {
            "world_size": dist.get_world_size(), "ranks_in_communicator": int(round(float(ones.item()))), "backend": dist.get_backend(),
            "grad_comm_dtype": "fp32" if es == 4 else str(self.grad_comm_dtype).replace("torch.", ""),
            "launch": self.launch, "overlap_backward": bool(self.overlap_backward), "steps": steps,
            "buckets": [{"tag": tag, "mbytes": round((hi - lo) * es / 1e6, 1), "rccl_ms_median": med(per.get(i, [])),
                         "gbps_algorithmic": (round((hi - lo) * es / 1e6 / med(per[i]), 1) if per.get(i) and med(per[i]) > 0 else None)}
                        for i, (tag, lo, hi) in enumerate(self.seg_arena.buckets)],
            "discriminator_allreduce_ms_median": med(dsc),
            "exposed_wait_before_adam_ms_median": med(waits),
        }

    def close(self):
        """Leave the fused-training mode (per-parameter gradients, immediate BatchNorm counters)."""
        F_.set_grad_sink(False)
        F_.defer_wgrads(False)
        F_.set_splitk_workspace(None)
        F_.set_wgrad_fold_workspace(None)
        F_.clear_mark_hooks()
        defer_bn_counters(False)
        for d in self.discs.values():
            d.rng_feed = None
        self._caps = {}
