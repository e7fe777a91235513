"""The training step of the hot path (a17: absent from the reference, whose OctaScribbleNet.forward
raises -- models/octa.py:59-60; defined in SURVEY.md 3.5):

    att, agg, _ = net.segmentor(x);  p = softmax(agg, 1)
    L_seg = WPCE(p, ys) [+ Dice(p, ys)] + kl_w * InterlayerDivergence([p, *att]) + adv_w * LSGen(D(att))
    L_seg.backward(); Adam(segmentor)
    L_d = LSDisc(D(real_pyramid), D([a.detach() for a in att])); L_d.backward(); Adam(discriminator)

Data parallel: one process per GPU, gradients live in two flat fp32 arenas (segmentor,
discriminator) that are all-reduced over RCCL (torch.distributed backend "nccl" on ROCm) and then
consumed by ONE fused Adam launch each.  BatchNorm statistics and the WPCE class weights are
per-replica, like DistributedDataParallel on the reference would be.
"""
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import functional as F_
from ._lib import lib
from .layers import defer_bn_counters, flush_bn_counters


FORCE_ALLREDUCE = os.environ.get("OCTA_DIST_ALWAYS") == "1"     # exercise the RCCL path with a single rank (tests)


class FlatArena:
    """Flat fp32 storage for the grad-bearing parameters of a module, their gradients and the Adam
    moments.  Parameters keep their logical shapes AND strides (channels-last conv weights stay so)."""

    def __init__(self, params: Sequence[nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            n = (n + 3) // 4 * 4            # 16-byte aligned slots
            offs.append(n)
            n += p.numel()
        self.numel = (n + 3) // 4 * 4
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
        self.step_count = 0

    def zero_grad(self):
        self.g.zero_()

    def needs_comm(self, world: int) -> bool:
        return world > 1 or (FORCE_ALLREDUCE and dist.is_available() and dist.is_initialized())

    def all_reduce(self, world: int):
        if self.needs_comm(world):
            dist.all_reduce(self.g, op=dist.ReduceOp.SUM)

    def all_reduce_begin(self, world: int, comm_stream):
        """Start the all-reduce on `comm_stream` once the gradients written on the current stream are complete;
        the caller overlaps independent work and calls all_reduce_end() before consuming the gradients."""
        if not self.needs_comm(world):
            return
        cur = torch.cuda.current_stream()
        comm_stream.wait_stream(cur)
        with torch.cuda.stream(comm_stream):
            dist.all_reduce(self.g, op=dist.ReduceOp.SUM)

    def all_reduce_end(self, world: int, comm_stream):
        if self.needs_comm(world):
            torch.cuda.current_stream().wait_stream(comm_stream)

    def adam(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, dyn: Optional[Tensor] = None):
        """One fused Adam launch over the arena.  `dyn` (device, 2 floats) carries the bias corrections when the
        launch is captured in a hipGraph; the caller then advances step_count / dyn itself (set_dyn)."""
        if dyn is None:
            self.step_count += 1
        lib().octa_adam_step(self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.numel, lr, betas[0], betas[1],
                             eps, weight_decay, max(self.step_count, 1), grad_scale, None if dyn is None else dyn.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
        # one launch refreshes every cached packed conv operand of THIS network; nothing else went stale, so the global
        # weight epoch is left alone (bumping it here made the other network's operands look stale: ~170 redundant
        # per-weight pack launches were captured into every replayed step)
        F_.repack_all(self.params)

    def advance_dyn(self, ring, betas):
        self.step_count += 1
        h = ring.slot()
        h[0] = 1.0 - betas[0] ** self.step_count
        h[1] = (1.0 - betas[1] ** self.step_count) ** 0.5
        ring.push()


def mask_pyramid(mask: Tensor, levels: int = 5) -> List[Tensor]:
    """Real multi-scale pyramid for the discriminator: nearest down-sampling by 2**i (contract of
    discriminator/blocks.py:114-125; views, no copy)."""
    return [mask[:, :, ::2 ** i, ::2 ** i] for i in range(levels)]


class _PinnedRing:
    """Rotating pinned staging buffers for small host->device updates issued while the GPU still runs
    earlier steps: a slot is only rewritten after the async copy that read it has completed (event)."""

    def __init__(self, shape, device, slots=4):
        self.host = [torch.zeros(shape, pin_memory=True) for _ in range(slots)]
        self.events = [None] * slots
        self.dev = torch.zeros(shape, device=device)
        self.i = 0

    def slot(self) -> Tensor:
        ev = self.events[self.i]
        if ev is not None:
            ev.synchronize()
        return self.host[self.i]

    def push(self):
        self.dev.copy_(self.host[self.i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.i] = ev
        self.i = (self.i + 1) % len(self.host)


class _RngFeed:
    """The CPU random draws of the three discriminator calls of a step (InstanceNoise plane then LabelNoise
    uniform, per call, from the GLOBAL CPU generator exactly like the reference: blocks.py:149-170), staged in
    pinned memory and copied to static device buffers so that a captured hipGraph can consume them."""

    def __init__(self, disc, device, calls=3):
        self.has_noise, self.has_label = disc._has_noise, disc._has_label_noise
        self.noise_mod = disc.stack_0[0] if self.has_noise else None
        self.label_mod = disc.out[2] if self.has_label else None
        hw = self.noise_mod.size if self.has_noise else (1, 1)
        self.calls = calls
        self.noise_ring = _PinnedRing((calls,) + tuple(hw), device)
        self.sign_ring = _PinnedRing((calls,), device)
        self.noise_dev, self.sign_dev = self.noise_ring.dev, self.sign_ring.dev
        self.sign_dev.fill_(1.0)
        self.i_noise = self.i_sign = 0

    def refill(self):
        nh, sh = self.noise_ring.slot(), self.sign_ring.slot()
        for c in range(self.calls):
            if self.has_noise:
                nh[c].copy_(self.noise_mod.draw())
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


class TrainStep:
    def __init__(self, net: nn.Module, lr: float = 1e-4, lr_disc: Optional[float] = None, betas=(0.9, 0.999), compute_dtype=torch.bfloat16,
                 adversarial: bool = True, use_dice: bool = True, kl_weight: float = 0.1, adv_weight: float = 0.1):
        self.net = net
        self.seg, self.disc = net.segmentor, getattr(net, "discriminator", None)
        self.adversarial = adversarial and self.disc is not None
        self.use_dice, self.kl_weight, self.adv_weight = use_dice, kl_weight, adv_weight
        self.lr, self.lr_disc, self.betas = lr, lr_disc or lr, betas
        self.seg.compute_dtype = compute_dtype
        if self.disc is not None:
            self.disc.compute_dtype = compute_dtype
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # linear_head_* never receive a gradient on this path: keep them out of the arena
        seg_params = [p for n, p in self.seg.named_parameters() if not n.startswith("linear_head_")]
        self.seg_arena = FlatArena(seg_params)
        self.disc_arena = FlatArena(list(self.disc.parameters())) if self.adversarial else None
        F_.set_grad_sink(True)
        F_.defer_wgrads(True)        # weight gradients are queued per stage and run as batched launches (functional.flush_wgrads)
        defer_bn_counters(True)
        self._graphs = None
        self._comm_stream = None
        self.launch = "graph"        # after capture(): "graph" replays the hipGraphs, "eager" launches the same step from Python

    # ------------------------------------------------------------------ the three phases of a step
    # (split at the two gradient all-reduces so that the collectives stay OUTSIDE any captured graph)
    def _phase_segmentor(self, x, ys, out):
        self.seg_arena.zero_grad()
        F_.ZERO_SLAB.begin(x.device)          # one clear for every small fp32 accumulator of the step
        att, agg, _ = self.seg(x)
        l = F_.wpce_dice(agg, ys, from_logits=True)
        loss = l[0] + l[1] if self.use_dice else l[0]
        out["wpce"], out["dice"] = l[0].detach(), l[1].detach()
        if self.adversarial:
            p = F_.class_softmax(agg)
            kl = F_.interlayer_kl([p, *att], [1] * len(att))[0]
            # the generator pass only needs dL/d(att) through D: D's own weight gradients of this pass are discarded
            # (zeroed before D's step, here and in the reference), so they are not computed at all
            for q in self.disc_arena.params:
                q.requires_grad_(False)
            g_adv = F_.lsgan_generator(self.disc(att))
            loss = loss + self.kl_weight * kl + self.adv_weight * g_adv
            out["kl"], out["g_adv"] = kl.detach(), g_adv.detach()
        try:
            loss.backward()
            F_.flush_wgrads()
        finally:
            if self.adversarial:
                for q in self.disc_arena.params:
                    q.requires_grad_(True)
        out["loss_seg"] = loss.detach()
        F_.ZERO_SLAB.end()
        flush_bn_counters()
        return [a.detach() for a in att]

    def _phase_discriminator(self, att, real_pyramid, out):
        """The discriminator's forward/backward.  It depends on the segmentor step only through the (detached)
        attention maps, not on the reduced segmentor gradients, so it runs WHILE those are being all-reduced."""
        if self.adversarial:
            self.disc_arena.zero_grad()
            d_real = self.disc(real_pyramid)
            d_fake = self.disc(att)
            l_d = F_.lsgan_discriminator(d_real, d_fake)
            l_d.backward()
            F_.flush_wgrads()
            out["loss_disc"] = l_d.detach()

    def _phase_seg_update(self, dyn=None):
        self.seg_arena.adam(self.lr, self.betas, grad_scale=1.0 / self.world, dyn=dyn)

    def _phase_finish(self, dyn=None):
        if self.adversarial:
            self.disc_arena.adam(self.lr_disc, self.betas, grad_scale=1.0 / self.world, dyn=dyn)

    def _comm(self):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        return self._comm_stream

    def __call__(self, x: Tensor, ys: Tensor, real_pyramid: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
        if self.adversarial and real_pyramid is None:
            raise ValueError("the adversarial step needs the real mask pyramid")
        if self._graphs is not None and self.launch == "graph":
            return self._replay(x, ys, real_pyramid)
        # launch == "eager" (or nothing captured): the step launched kernel by kernel from Python.  Same arenas, step counters
        # and global CPU generator as the replayed path, so the two can be mixed freely.
        out: Dict[str, Tensor] = {}
        feed = getattr(self.disc, "rng_feed", None) if self.disc is not None else None
        if feed is not None:
            self.disc.rng_feed = None          # captured earlier: this path draws the discriminator's noise inline, like the reference
        try:
            att = self._phase_segmentor(x, ys, out)
            self.seg_arena.all_reduce_begin(self.world, self._comm())      # 286 MB over xGMI, hidden behind the D step
            self._phase_discriminator(att, real_pyramid, out)
            self.seg_arena.all_reduce_end(self.world, self._comm())
            self._phase_seg_update()
            if self.adversarial:
                self.disc_arena.all_reduce(self.world)
            self._phase_finish()
        finally:
            if feed is not None:
                self.disc.rng_feed = feed
        return out

    # ------------------------------------------------------------------ hipGraph capture / replay
    def capture(self, x: Tensor, ys: Tensor, real_pyramid: Optional[Sequence[Tensor]] = None, warmup: int = 2):
        """Capture the step into three hipGraphs (torch.cuda.CUDAGraph = hipGraph on ROCm) around the two
        all-reduces.  Inputs are copied into static buffers on every call; the CPU random draws of the
        discriminator are staged through _RngFeed; Adam's bias corrections come from device memory."""
        dev = x.device
        self._sx, self._sys = x.clone(), ys.clone()
        self._sreal = [r.contiguous().clone() for r in real_pyramid] if real_pyramid is not None else None
        if self.adversarial:
            self._feed = _RngFeed(self.disc, dev)
            self.disc.rng_feed = self._feed
        self._dyn = [_PinnedRing((2,), dev) for _ in range(2)]
        self._dyn_dev = [r.dev for r in self._dyn]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):             # warm-up off the default stream, as graph capture requires
            F_.set_conv_autotune(True)            # the first warm-up step measures the fwd / dgrad kernel choice per layer shape
            try:
                for _ in range(warmup):
                    self._eager_static()
            finally:
                F_.set_conv_autotune(False)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._out: Dict[str, Tensor] = {}
        # the warm-up steps ended with Adam + repack_all: every cached operand is current, so no per-weight pack launch
        # is captured (a stale cache here would put ~180 redundant pack kernels into every replay)
        if self.adversarial:
            self._feed.rewind()
        # "thread_local": RCCL's watchdog thread keeps polling the events of the warm-up all-reduces with hipEventQuery while this
        # thread captures; under the default global mode that query is an error and the watchdog aborts the process
        mode = "thread_local"
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode=mode):
            self._att = self._phase_segmentor(self._sx, self._sys, self._out)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode=mode):
            self._phase_discriminator(self._att, self._sreal, self._out)
        g2b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2b, pool=g1.pool(), capture_error_mode=mode):
            self._phase_seg_update(dyn=self._dyn_dev[0])
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, pool=g1.pool(), capture_error_mode=mode):
            self._phase_finish(dyn=self._dyn_dev[1])
        self._graphs = (g1, g2, g2b, g3)
        return self

    def _eager_static(self):
        out: Dict[str, Tensor] = {}
        if self.adversarial:
            self._feed.refill()
        self.seg_arena.advance_dyn(self._dyn[0], self.betas)
        att = self._phase_segmentor(self._sx, self._sys, out)
        self.seg_arena.all_reduce_begin(self.world, self._comm())
        self._phase_discriminator(att, self._sreal, out)
        self.seg_arena.all_reduce_end(self.world, self._comm())
        self._phase_seg_update(dyn=self._dyn_dev[0])
        if self.adversarial:
            self.disc_arena.advance_dyn(self._dyn[1], self.betas)
            self.disc_arena.all_reduce(self.world)
        self._phase_finish(dyn=self._dyn_dev[1])
        return out

    def _load_static(self, x, ys, real_pyramid):
        if x is not self._sx:
            self._sx.copy_(x, non_blocking=True)
        if ys is not self._sys:
            self._sys.copy_(ys, non_blocking=True)
        if self._sreal is not None and real_pyramid is not None:
            for dst, src in zip(self._sreal, real_pyramid):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)

    def autotune_launch(self, x, ys, real_pyramid=None, rounds: int = 3, steps: int = 4) -> str:
        """Pick the faster launch path for THIS process on THIS host: hipGraph replay needs almost no host time but pays a few
        microseconds of dependency handling per node and stalls when the box is busy; launching the ~1100 kernels from Python
        costs the host ~35 ms/step, which is enough to keep an otherwise quiet GPU fed (measured a steady 39.7 ms/step eager
        against 39.6-66 ms/step replayed across boxes; with a slower or shared host core the order flips).  Both paths run the
        identical step on the same arenas; `rounds` alternating timings of `steps` real training steps each, the medians decide."""
        import time
        if self._graphs is None:
            return self.launch
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
        self.launch = "eager" if med["eager"] < med["graph"] else "graph"
        self.launch_timing = med
        return self.launch

    def _replay(self, x, ys, real_pyramid):
        self._load_static(x, ys, real_pyramid)
        if self.adversarial:
            self._feed.refill()
        g1, g2, g2b, g3 = self._graphs
        self.seg_arena.advance_dyn(self._dyn[0], self.betas)
        if self.adversarial:
            self.disc_arena.advance_dyn(self._dyn[1], self.betas)
        g1.replay()
        self.seg_arena.all_reduce_begin(self.world, self._comm())
        g2.replay()                                  # D step overlaps the segmentor gradient all-reduce
        self.seg_arena.all_reduce_end(self.world, self._comm())
        g2b.replay()
        if self.adversarial:
            self.disc_arena.all_reduce(self.world)
        g3.replay()
        F_.bump_weight_epoch()      # the weights moved behind the pack cache's back
        return self._out

    def close(self):
        """Leave the fused-training mode (per-parameter gradients, immediate BatchNorm counters)."""
        F_.set_grad_sink(False)
        F_.defer_wgrads(False)
        F_.clear_mark_hooks()
        defer_bn_counters(False)
        if self.disc is not None:
            self.disc.rng_feed = None
        self._graphs = None
