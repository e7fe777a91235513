"""Host side of the hot path: raw wrappers over the C ABI and the torch.autograd.Functions built
on them.  PyTorch provides device memory, streams and the autograd tape only; every FLOP below
runs in libocta_hip.so.  Activations are torch tensors of logical shape (B, C, H, W) whose MEMORY
is NHWC (channels-last) with a per-pixel stride `ld` that is a multiple of 8.
"""
import ctypes
import os
import threading
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import (ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, OCTA_BF16, OCTA_F16, OCTA_F32, ConvDesc,
                   OctaError, WgradJob, lib)

Tensor = torch.Tensor
_I64x4 = ctypes.c_int64 * 4


# ----------------------------------------------------------------------------- basics
def _dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return OCTA_F32
    if d == torch.bfloat16:
        return OCTA_BF16
    if d == torch.float16:
        return OCTA_F16
    raise OctaError(f"unsupported activation dtype {d} (float32, bfloat16 or float16)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def _st():
    """Raw hipStream_t of torch's current stream (this runs ~1100 times per step: the Stream-object route costs 9 us a call)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _require_gpu(t: Tensor):
    if not t.is_cuda:
        raise OctaError("octave_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")


def _strides4(t: Tensor):
    return _I64x4(*t.stride())


def round8(c: int) -> int:
    return (c + 7) // 8 * 8


def nhwc_empty(B: int, C: int, H: int, W: int, dtype, device, zero: bool = False, pad_written: bool = False) -> Tensor:
    """(B,C,H,W)-shaped tensor with NHWC memory, ld = round8(C); padded channels are zeroed unless the caller's kernel
    writes them itself (pad_written)."""
    ld = round8(C)
    mk = torch.zeros if (zero or (ld != C and not pad_written)) else torch.empty
    buf = mk((B, H, W, ld), dtype=dtype, device=device)
    t = buf.permute(0, 3, 1, 2)
    return t if ld == C else t[:, :C]


def nhwc_ld(t: Tensor) -> Optional[int]:
    """Per-pixel stride if `t` (B,C,H,W) is NHWC-strided and chunk-aligned, else None."""
    if t.dim() != 4:
        return None
    B, C, H, W = t.shape
    s = t.stride()
    if C > 1 and s[1] != 1:
        return None
    if W > 1:
        ld = s[3]
    elif H > 1:
        ld = s[2]
    elif B > 1:
        ld = s[0]
    else:
        ld = round8(C)
    if ld < C or ld % 8 != 0:
        return None
    if H > 1 and s[2] != W * ld:
        return None
    if B > 1 and s[0] != H * W * ld:
        return None
    if t.data_ptr() % 16 != 0:
        return None
    return ld


def to_nhwc(x: Tensor, dtype=None, cpad: Optional[int] = None) -> Tensor:
    """Any (B,C,H,W) tensor -> NHWC-strided tensor of `dtype` whose buffer holds >= cpad channels
    (extra channels zero).  No-op when it already qualifies."""
    _require_gpu(x)
    dtype = dtype or x.dtype
    B, C, H, W = x.shape
    cpad = cpad or C
    ld = nhwc_ld(x)
    if ld is not None and x.dtype == dtype and ld >= cpad:
        return x
    src = x if x.dtype == torch.float32 else x.float()
    out = nhwc_empty(B, max(cpad, C), H, W, dtype, x.device, pad_written=True)      # the kernel zero-fills channels C .. ld-1
    base = out if out.shape[1] == C else out[:, :C]
    ldo = round8(max(cpad, C))
    lib().octa_nchw_to_nhwc(_p(src), src.stride(0), src.stride(1), src.stride(2), src.stride(3), _p(base), B, C, H, W,
                            ldo, 0, ldo, _dt(dtype), _st())
    return base


def to_nchw_f32(x: Tensor) -> Tensor:
    """NHWC-strided activation -> dense NCHW fp32 tensor."""
    ld = nhwc_ld(x)
    if ld is None:
        return x.float().contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    lib().octa_nhwc_to_nchw(_p(x), ld, 0, _dt(x), _p(out), B, C, H, W, 0, _st())
    return out


def dense_nhwc(x: Tensor) -> Tensor:
    """NHWC tensor with ld == C (copy when it is a channel slice of a wider buffer)."""
    x = to_nhwc(x)
    B, C, H, W = x.shape
    ld = nhwc_ld(x)
    if ld == C:
        return x
    if C % 8 != 0:
        raise OctaError(f"dense_nhwc: C={C} is not a multiple of 8")
    out = nhwc_empty(B, C, H, W, x.dtype, x.device)
    lib().octa_copy_channels(_p(x), H, W, ld, 0, _p(out), H, W, C, 0, B, C, _dt(x), 0, _st())
    return out


# ----------------------------------------------------------------------------- zero slab
class _ZeroSlab:
    """fp32 scratch slab cleared by ONE launch at the start of a training step.  Ops that need a zero-initialised fp32
    accumulator (split-attention GAP / logit gradients, ...) take a slice instead of launching a zero fill each
    (63 launches per step).  Bump allocation, never reused inside a step; inactive outside a TrainStep."""

    def __init__(self, nfloats: int = 4 << 20):
        self.cap, self.buf, self.off, self.high, self.active, self.zeroed = nfloats, None, 0, 0, False, 0

    def begin(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.cap, dtype=torch.float32, device=device)
            self.high, self.zeroed = 0, self.cap
        else:
            n = self.cap if self.high == 0 else min(self.cap, self.high)
            self.buf[:n].zero_()
            self.zeroed = n          # slices beyond this are NOT clean (an earlier, larger user): take() refuses them
        self.off, self.active = 0, True

    def end(self):
        self.high = max(self.high, self.off)
        self.active = False

    def take(self, shape) -> Optional[Tensor]:
        if not self.active:
            return None
        n = 1
        for v in shape:
            n *= v
        a = (self.off + 15) // 16 * 16
        self.high = max(self.high, min(self.cap, a + n))     # the next begin() clears this far
        if a + n > self.zeroed:
            return None              # the caller's entry point zero-fills its own buffer this once
        self.off = a + n
        return self.buf[a:a + n].view(shape)


ZERO_SLAB = _ZeroSlab()


def _zeroed_f32(shape, device):
    """(tensor, prezeroed flag): a slab slice that is already zero, or an uninitialised tensor the kernel must clear."""
    t = ZERO_SLAB.take(shape)
    if t is not None and t.device == device:
        return t, 1
    return torch.empty(shape, dtype=torch.float32, device=device), 0


# ----------------------------------------------------------------------------- per-call scratch of the conv engine
# The library keeps no device pointer between calls (include/octa_hip.h: octa_conv_desc.ws, octa_conv2d_wgrad_batch's ws): the scratch
# of the 8-wave kernels' tail split and of the weight gradients' partial-tile fold travels with every call.  WHICH buffer a call gets is
# host-side context: a training step enters its phase's buffers here (the discriminator's step, replayed on a second stream, has a
# fold scratch of its own and no tail-split scratch), side branches launch without a tail-split scratch.
_SPLITK_WS = None
_WGRAD_FOLD_WS = None


def set_wgrad_fold_workspace(ws: Optional[Tensor]):
    """fp32 scratch handed to every weight-gradient call from now on (None: none): the single-problem kernels then store private
    partial tiles and one fold launch per batch adds them to the gradient -- no float atomics, deterministic.  Phases that run
    concurrently on two streams enter different buffers."""
    global _WGRAD_FOLD_WS
    if ws is not None and (ws.dtype != torch.float32 or not ws.is_cuda or not ws.is_contiguous() or ws.data_ptr() % 16):
        raise OctaError("set_wgrad_fold_workspace: contiguous, 16-byte aligned fp32 device tensor")
    _WGRAD_FOLD_WS = ws


def set_splitk_workspace(ws: Optional[Tensor]):
    """fp32 scratch handed to every forward / data-gradient conv call from now on (None: none) for the 8-wave kernels' tail split."""
    global _SPLITK_WS
    if ws is not None and (ws.dtype != torch.float32 or not ws.is_cuda or not ws.is_contiguous() or ws.data_ptr() % 16):
        raise OctaError("split-K workspace: contiguous, 16-byte aligned fp32 device tensor")
    _SPLITK_WS = ws


def _fold_ws_args():
    w = _WGRAD_FOLD_WS
    return (None, 0) if w is None else (w.data_ptr(), w.numel() * 4)


def set_deterministic(on: bool):
    """octa_tuning_set(5, .): every cross-workgroup sum of the library in a fixed order (parity tests; slower)."""
    lib().octa_tuning_set(5, 1 if on else 0)


# ----------------------------------------------------------------------------- side branches (second stream)
# A block's shortcut branch (avg-pool -> 1x1 -> BatchNorm of a strided bottleneck; the 1x1 of a decoder block) does not depend on the
# main branch until the final add.  Issued on a second stream between a fork and a join, it runs BESIDE the main branch: eagerly as a
# second HIP queue, inside a captured step as a parallel branch of the hipGraph.  autograd replays the stream of every node's forward in
# its backward and inserts the cross-stream waits itself; what it cannot see is handled here: side-branch launches get no tail-split
# scratch (the main branch's launches are using it), and a gradient parked in a GradHolder travels by event.
_SIDE_BRANCH = os.environ.get("OCTA_SIDE_BRANCH", "1") != "0"
_SIDE_STREAMS = {}
_IN_SIDE = False
_NO_SPLIT = 0


class no_splitk:
    """Conv launches inside this context never use the tail-split scratch (they may run beside launches that do)."""

    def __enter__(self):
        global _NO_SPLIT
        _NO_SPLIT += 1
        return self

    def __exit__(self, *a):
        global _NO_SPLIT
        _NO_SPLIT -= 1


def _set_conv_ws(d):
    """The tail-split scratch of this fwd / dgrad call: the phase's buffer, or none inside a side branch."""
    w = None if _NO_SPLIT else _SPLITK_WS
    if w is None:
        d.ws, d.ws_bytes = None, 0
    else:
        d.ws, d.ws_bytes = w.data_ptr(), w.numel() * 4


class SideBranch:
    """with SideBranch(x) as br: y = shortcut(x)   # on the second stream, behind everything the current stream has enqueued so far
       ... main branch on the current stream ...
       br.join(y)                                   # the current stream waits for the branch before y is used"""

    def __init__(self, after=None):
        # after: an event recorded earlier on the current stream (fork_point()): the branch then waits for THAT point only, not for
        # what the current stream enqueued since (the main branch's first kernels)
        self.after = after
        self.cur = torch.cuda.current_stream()
        key = self.cur.device.index
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.cur.device)
        self.side = _SIDE_STREAMS[key]
        if self.side == self.cur:                          # (a caller that already runs on the side stream: take another one)
            self.side = torch.cuda.Stream(device=self.cur.device)

    def __enter__(self):
        global _IN_SIDE
        if self.after is not None:
            self.side.wait_event(self.after)
        else:
            self.side.wait_stream(self.cur)
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        self._prev, _IN_SIDE = _IN_SIDE, True
        self._ns = no_splitk().__enter__()
        return self

    def __exit__(self, *a):
        global _IN_SIDE
        self._ns.__exit__(*a)
        _IN_SIDE = self._prev
        return self._ctx.__exit__(*a)

    def join(self, *tensors):
        self.cur.wait_stream(self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(self.cur)                  # allocated on the side stream, consumed (and possibly freed) on this one


def fork_point():
    """An event on the current stream, for SideBranch(after=...)."""
    ev = torch.cuda.Event()
    ev.record()
    return ev


def side_branch_ok(x) -> bool:
    return _SIDE_BRANCH and x.is_cuda and not _IN_SIDE


# ----------------------------------------------------------------------------- packed weights
_WEIGHT_EPOCH = 0
_PARAM_EPOCH = 0          # bumped whenever a kernel writes parameters or BatchNorm running statistics through raw pointers
_PACK_CACHE = {}          # (id(param), kind, dtype, groups, pad_to) -> _PackEntry  (persistent output buffers)
_PACK_PLANS = {}          # frozenset of cache keys -> (desc table, prefix, n, total) on the device


class _PackEntry:
    __slots__ = ("tag", "out", "wref", "direct", "kind", "dtype", "groups", "pad_to", "aux")


def bump_weight_epoch():
    """Call after parameters were modified outside autograd's version counter (fused optimiser): every packed
    operand becomes stale until it is refreshed (individually on next use, or all at once by repack_all)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def bump_param_epoch():
    """A kernel moved parameter / running-statistic VALUES behind autograd's version counters (fused Adam, BatchNorm training
    forward).  Packed operands are refreshed by repack_all; value-derived caches that are not (layers._fold_entry: conv+BN
    folded for inference) compare this counter."""
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


def _pack_tag(w: Tensor):
    return (w._version, _WEIGHT_EPOCH, w.data_ptr(), tuple(w.shape), tuple(w.stride()))


def _pack_numel(w: Tensor, kind: str, groups: int, pad_to: int) -> int:
    O, Ig, KH, KW = w.shape
    if kind == "fwd":
        return O * KH * KW * pad_to
    if kind == "dgrad":
        return groups * Ig * KH * KW * pad_to
    if kind == "fwd_dense":
        return O * KH * KW * pad_to
    if kind == "dgrad_dense":
        return groups * Ig * KH * KW * pad_to
    if kind == "dgrad_taps":
        return KH * KW * round8(Ig) * pad_to
    return 4 * Ig * pad_to      # convT: w is (CinT, CoutT, 2, 2)


_PACK_KIND = {"fwd": 0, "dgrad": 1, "convT": 2, "fwd_dense": 3, "dgrad_dense": 4, "dgrad_taps": 5}


def _pack_desc_fill(d, kind: str, w: Tensor, out: Tensor, dtype, groups: int, pad_to: int):
    O, Ig, KH, KW = w.shape
    d.src, d.dst = w.data_ptr(), out.data_ptr()
    d.s_o, d.s_i, d.s_h, d.s_w = w.stride()
    d.dtype, d.KH, d.KW, d.groups, d.pad_to = _dt(dtype), KH, KW, groups, pad_to
    d.kind = _PACK_KIND[kind]
    if kind == "convT":
        d.Cout_g, d.Cin_g = Ig, O
    else:
        d.Cout_g, d.Cin_g = O // groups, Ig


def _packed(w: Tensor, kind: str, dtype, groups: int, pad_to: int) -> Tensor:
    # only nn.Parameters are cached (they persist); identity is checked through a weak reference
    # because both id() and data_ptr() are recycled once a tensor dies
    pre = getattr(w, "_octa_packed", None)       # operands that came with the tensor (a spectral-normalised weight: SpectralNormBatchFn)
    if pre is not None:
        t = pre.get((kind, dtype, groups, pad_to))
        if t is not None:
            return t
    cacheable = isinstance(w, torch.nn.Parameter)
    key = (id(w), kind, dtype, groups, pad_to)
    tag = _pack_tag(w)
    entry = _PACK_CACHE.get(key) if cacheable else None
    if entry is not None and entry.wref() is not w:
        entry = None
    if entry is not None and entry.tag == tag:
        return entry.out
    L = lib()
    wd = w.detach()
    if wd.dtype != torch.float32:
        wd = wd.float()
    O, Ig, KH, KW = wd.shape
    s = wd.stride()
    direct = False
    aux = None
    n = _pack_numel(wd, kind, groups, pad_to)
    out = entry.out if (entry is not None and not entry.direct and entry.out.numel() == n and entry.tag[2:] == tag[2:]) else None
    if kind == "fwd":
        # already [O][KH][KW][Ig] fp32 with Ig % 8 == 0 -> use the parameter storage itself
        if (dtype == torch.float32 and pad_to == Ig and s[1] == 1 and (KW == 1 or s[3] == Ig) and (KH == 1 or s[2] == KW * Ig)
                and s[0] == KH * KW * Ig and wd.data_ptr() % 16 == 0):
            out, direct = wd, True
        else:
            out = out if out is not None else torch.empty((n,), dtype=dtype, device=w.device)
            L.octa_pack_weight_fwd(_p(wd), s[0], s[1], s[2], s[3], _p(out), O, Ig, KH, KW, groups, pad_to, _dt(dtype), _st())
    elif kind == "dgrad":
        out = out if out is not None else torch.empty((n,), dtype=dtype, device=w.device)
        L.octa_pack_weight_dgrad(_p(wd), s[0], s[1], s[2], s[3], _p(out), O, Ig, KH, KW, groups, pad_to, _dt(dtype), _st())
    elif kind == "dgrad_taps":  # rows (kh, kw, ci < round8(Cin)), columns co < pad_to: the tap-major GEMM + col2im data gradient
        out = out if out is not None else torch.empty((n,), dtype=dtype, device=w.device)
        L.octa_pack_weight_dgrad_taps(_p(wd), s[0], s[1], s[2], s[3], _p(out), O, Ig, KH, KW, round8(Ig), pad_to, _dt(dtype), _st())
    elif kind == "convT":      # w: (CinT, CoutT, 2, 2)
        out = out if out is not None else torch.empty((n,), dtype=dtype, device=w.device)
        L.octa_pack_weight_convT(_p(wd), s[0], s[1], s[2], s[3], _p(out), O, Ig, pad_to, _dt(dtype), _st())
    elif kind in ("fwd_dense", "dgrad_dense"):
        # one-entry multi-pack (first use only; afterwards repack_all refreshes it with every other operand)
        from ._lib import PackDesc
        if out is not None and entry.aux is not None:
            tb, pf, tiles = entry.aux          # device-resident descriptor: the refresh is capturable
        else:
            out = torch.empty((n,), dtype=dtype, device=w.device)
            table = (PackDesc * 1)()
            _pack_desc_fill(table[0], kind, wd, out, dtype, groups, pad_to)
            tiles = int(L.octa_pack_tile_count(ctypes.byref(table[0])))
            tb = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(w.device)
            pf = torch.zeros(1, dtype=torch.int64, device=w.device)
        aux = (tb, pf, tiles)
        L.octa_pack_many(_p(tb), _p(pf), 1, tiles, _st())
    else:
        raise ValueError(kind)
    if cacheable:
        if len(_PACK_CACHE) > 8192:
            # evict only entries whose parameter is gone: live entries may be baked into captured hipGraphs (their buffers
            # and the device-side descriptor tables of _PACK_PLANS must stay where they are)
            for k in [k for k, e_ in _PACK_CACHE.items() if e_.wref() is None]:
                del _PACK_CACHE[k]
        e = _PackEntry()
        e.tag, e.out, e.wref, e.direct, e.kind, e.dtype, e.groups, e.pad_to = tag, out, weakref.ref(w), direct, kind, dtype, groups, pad_to
        e.aux = aux
        _PACK_CACHE[key] = e
    return out


def repack_all(params) -> int:
    """Refresh, with ONE launch, every cached packed operand whose source parameter is in `params` (after the fused
    optimiser moved the weights).  The descriptor table lives on the device and is rebuilt only when the set of
    operands changes, so the call is hipGraph-capturable.  Returns the number of operands refreshed."""
    from ._lib import PackDesc
    ids = {id(p) for p in params}
    todo = []
    for key, e in _PACK_CACHE.items():
        w = e.wref()
        if w is None or key[0] not in ids or e.direct:
            if w is not None and key[0] in ids and e.direct:
                e.tag = _pack_tag(w)          # the parameter storage is the operand: always current
            continue
        todo.append((key, e, w))
    if not todo:
        return 0
    sig = frozenset((k, e.out.data_ptr(), w.data_ptr(), tuple(w.stride())) for k, e, w in todo)
    plan = _PACK_PLANS.get(sig)
    if plan is None:
        table = (PackDesc * len(todo))()
        prefix, total = [], 0
        for i, (key, e, w) in enumerate(todo):
            _pack_desc_fill(table[i], e.kind, w, e.out, e.dtype, e.groups, e.pad_to)
            prefix.append(total)
            total += int(lib().octa_pack_tile_count(ctypes.byref(table[i])))
        dev = todo[0][2].device
        tb = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
        pf = torch.tensor(prefix, dtype=torch.int64, device=dev)
        plan = (tb, pf, len(todo), total, [e.wref for _, e, _ in todo])
        if len(_PACK_PLANS) > 16:
            # drop only plans whose parameters are gone: the device tables of the others may be baked into captured hipGraphs
            for k in [k for k, pl in _PACK_PLANS.items() if any(r() is None for r in pl[4])]:
                del _PACK_PLANS[k]
        _PACK_PLANS[sig] = plan
    tb, pf, n, total = plan[:4]
    lib().octa_pack_many(_p(tb), _p(pf), n, total, _st())
    for key, e, w in todo:
        e.tag = _pack_tag(w)
    return n


# ----------------------------------------------------------------------------- launch recorder (bench.py roofline leg)
_RECORD = None


def start_recording():
    """Record every conv-engine launch (kind, descriptor, buffers) of the following calls so that bench.py
    can replay each one between HIP events and attribute time/FLOPs per kernel."""
    global _RECORD
    _RECORD = []


def stop_recording():
    global _RECORD
    r, _RECORD = _RECORD, None
    return r


def _record(kind, d, ptrs, keep):
    if _RECORD is not None:
        dd = ConvDesc()
        ctypes.memmove(ctypes.byref(dd), ctypes.byref(d), ctypes.sizeof(ConvDesc))
        dd.alg_groups = getattr(d, "alg_groups", d.groups)       # densified grouped layers keep their algorithmic group count
        _RECORD.append((kind, dd, ptrs, keep))


# ----------------------------------------------------------------------------- raw conv ops
_ALGO_OVERRIDE = 0      # octa_conv_desc.algo for every descriptor built below (0 = library heuristic; tools/conv8_micro.py sweeps it)


# Per-shape kernel choice for fwd / dgrad (octa_conv_desc.algo), measured once per shape while a training step warms up:
# small layers prefer the 4-wave kernels (occupancy), big ones the 8-wave LDS-DMA kernel; the crossover depends on M, N, K,
# taps and groups in ways a formula only approximates, so it is measured (3 launches per candidate, HIP events).
_AUTOTUNE = False
_ALGO_CACHE = {}


def set_conv_autotune(on: bool):
    global _AUTOTUNE
    _AUTOTUNE = bool(on)


def save_algo_cache(path: str):
    """Persist the measured per-shape kernel choices (JSON) so that a later process -- a profiled run, a restarted job --
    starts with them instead of timing every shape again."""
    import json
    with open(path, "w") as f:
        json.dump([[list(k), v] for k, v in _ALGO_CACHE.items()], f)


def load_algo_cache(path: str) -> int:
    import json
    if not os.path.exists(path):
        return 0
    for k, v in json.load(open(path)):
        _ALGO_CACHE[tuple(k)] = int(v)
    return len(_ALGO_CACHE)


def _choose_algo(kind: str, d, launch) -> int:
    if _ALGO_OVERRIDE:
        return _ALGO_OVERRIDE
    if d.dtype == OCTA_F32:
        return 0
    key = (kind, d.dtype, d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.stride, d.pad, d.groups, d.ldx, d.ldy, d.upshuffle, d.act)
    a = _ALGO_CACHE.get(key)
    if a is not None:
        return a
    if not _AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return 0
    cg = d.cin_g_pad if kind in ("fwd", "fwd_stats") else d.cout_g_pad
    cands = [1, 4, 5, 6]            # heuristic (incl. the 3x3 halo kernels), then the explicit 4-wave tiles
    if kind == "dgrad_gated":
        cands = [4, 5, 6]           # (only the generic kernel's epilogue applies an activation gate)
    elif cg % 64 == 0 and d.KH * d.KW <= 32 and (kind in ("fwd", "fwd_stats") or d.stride == 1):
        cands += [2, 3, 8]          # 8-wave LDS-DMA kernel, both slab orientations; its 4-wave 128x128 form
    if kind == "fwd_stats":
        pass                        # (the resident-weight, pointwise-GEMM and 2-D patch kernels cannot take the statistics along)
    elif kind not in ("dgrad_add", "dgrad_gated") and cg in (32, 64) and d.groups in (1, 2, 4, 8) and d.stride == 1 and ((d.KH == 3 and d.pad == 1) or (d.KH == 1 and d.pad == 0)):
        cands += [7]                # resident-weight persistent kernel (ineligible shapes fall back to the heuristic)
    if kind not in ("fwd_stats", "dgrad_gated") and cg % 64 == 0 and d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad == 0 and d.groups == 1 and os.environ.get("OCTA_NO_PWGEMM") != "1":
        cands += [9, 10, 11]        # persistent pointwise GEMM (pwgemm.hpp): 256x128, 128x256, 128x128 tiles, cross-tile pipelined
    if kind not in ("dgrad_add", "fwd_stats", "dgrad_gated") and cg % 64 == 0 and d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad == 1 and os.environ.get("OCTA_NO_HALO8") != "1":
        cands += [12]               # 8-wave 3x3 kernel, 2-D pixel patch per tile (halo8.hpp)
        if os.environ.get("OCTA_NO_HALO16") != "1":
            cands += [13, 14]       # ... on v_mfma_f32_16x16x32 (halo16.hpp) / with a persistent tile loop (halo16p.hpp)
    best, best_t = cands[0], None
    if len(cands) > 1:
        for c in cands:
            d.algo = c
            launch()
            t = None
            for _ in range(2):                             # the better of two timings of three launches (one noisy sample used to decide ~1 pick in 20)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(); launch(); launch()
                e1.record()
                e1.synchronize()
                tt = e0.elapsed_time(e1)
                t = tt if t is None else min(t, tt)
            if best_t is None or t < best_t * 0.97:        # a challenger must win by 3 %
                best, best_t = c, t
    _ALGO_CACHE[key] = best
    return best


class ConvStats:
    """BatchNorm statistics taken in the producing conv's epilogue (octa_conv2d_fwd_stats): `sums` is a zero-filled fp32
    (replicas, 2, Cout) buffer, `shift` the BatchNorm's running mean (or None).  `fused` is set by the conv launch: False when
    the kernel chosen for this shape cannot do it (3x3 halo / resident-weight kernels) -- BatchNorm then runs its own pass."""
    __slots__ = ("sums", "shift", "replicas", "fused")

    def __init__(self, cout: int, shift: Optional[Tensor], device, replicas: int = 16):
        t = ZERO_SLAB.take((replicas, 2, cout))
        self.sums = t if (t is not None and t.device == device) else torch.zeros((replicas, 2, cout), dtype=torch.float32, device=device)
        self.shift, self.replicas, self.fused = shift, replicas, False


def _launch_fwd(d, x, wp, bias, y, stats: Optional["ConvStats"] = None):
    L, st = lib(), _st()
    px, pw, pb, py = _p(x), _p(wp), _p(bias), _p(y)
    _set_conv_ws(d)                 # (before the tuner: its timing launches are this call's launches)
    if stats is not None:
        # the statistics-fusing launch is its own shape class: only kernels that can fuse are candidates, and they are timed as such
        flag = ctypes.c_int(0)

        def launch_stats():
            L.octa_conv2d_fwd_stats(ctypes.byref(d), px, pw, pb, py, _p(stats.sums), _p(stats.shift), stats.replicas, ctypes.byref(flag), st)
        d.algo = _choose_algo("fwd_stats", d, launch_stats)
        if _AUTOTUNE and not torch.cuda.is_current_stream_capturing():
            stats.sums.zero_()      # (the tuner's launches added into the accumulators)
        launch_stats()
        stats.fused = bool(flag.value)
        return
    d.algo = _choose_algo("fwd", d, lambda: L.octa_conv2d_fwd(ctypes.byref(d), px, pw, pb, py, st))
    L.octa_conv2d_fwd(ctypes.byref(d), px, pw, pb, py, st)


def _launch_dgrad(d, dy, wt, dx, addend=None, gate=None):
    L, st = lib(), _st()
    pdy, pw, pdx = _p(dy), _p(wt), _p(dx)
    _set_conv_ws(d)
    if gate is not None:
        # dx = conv^T(dy) * f'(gate) in the generic kernel's epilogue (small layers: the library picks the tile by size)
        gx, ldg, gact, _ = gate
        pg = _p(gx)
        d.algo = _choose_algo("dgrad_gated", d, lambda: L.octa_conv2d_dgrad_gated(ctypes.byref(d), pdy, pw, pg, ldg, gact, pdx, st))
        L.octa_conv2d_dgrad_gated(ctypes.byref(d), pdy, pw, pg, ldg, gact, pdx, st)
        return
    if addend is not None:
        # dx = conv^T(dy) + addend in the kernel's epilogue (the generic / LDS-DMA kernels; tuned as its own shape class)
        pa, lda = _p(addend), nhwc_ld(addend)
        d.algo = _choose_algo("dgrad_add", d, lambda: L.octa_conv2d_dgrad_add(ctypes.byref(d), pdy, pw, pa, lda, pdx, st))
        L.octa_conv2d_dgrad_add(ctypes.byref(d), pdy, pw, pa, lda, pdx, st)
        return
    d.algo = _choose_algo("dgrad", d, lambda: L.octa_conv2d_dgrad(ctypes.byref(d), pdy, pw, pdx, st))
    L.octa_conv2d_dgrad(ctypes.byref(d), pdy, pw, pdx, st)


def _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups, ldx, ldy, dtype, act=ACT_NONE, upshuffle=0) -> ConvDesc:
    d = ConvDesc()
    d.B, d.H, d.W, d.OH, d.OW = B, H, W, OH, OW
    d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.groups = Cin, Cout, KH, KW, stride, pad, groups
    d.cin_g_pad = round8(Cin // groups)
    d.cout_g_pad = round8(Cout // groups)
    d.ldx, d.xoff, d.ldy, d.yoff = ldx, 0, ldy, 0
    d.dtype, d.act, d.upshuffle = _dt(dtype), act, upshuffle
    d.algo = _ALGO_OVERRIDE
    return d


def _conv_geometry(x: Tensor, w: Tensor, stride: int, pad: int):
    B, Cin, H, W = x.shape
    Cout, Cin_g, KH, KW = w.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return B, Cin, H, W, Cout, Cin_g, KH, KW, OH, OW


def _conv_input(x: Tensor, cin_g_pad: int, groups: int) -> Tuple[Tensor, int]:
    """NHWC view of x whose buffer is readable for the padded channel count."""
    need = cin_g_pad if groups == 1 else x.shape[1]
    x = to_nhwc(x, cpad=need)
    ld = nhwc_ld(x)
    if ld < need:
        x = to_nhwc(x.contiguous(), cpad=need)
        ld = nhwc_ld(x)
    return x, ld


def _densify(groups: int, Cin: int, Cout: int, KH: int, KW: int, stride: int, pad: int, H: int, W: int, dtype) -> int:
    """Small-channel grouped 3x3/s1/p1 layers at high resolution are loader-bound on the grouped gather kernel: each group
    re-reads its 16/32-byte slice of every pixel for 9 taps (and the groups' workgroups fetch every activation line once each).
    Returns the number of adjacent groups to MERGE into one dense block with block-diagonal weights (0 = run as is), so that
    a merged group has 32 input channels and the halo-reuse / resident-weight kernels take the layer (pixel patch loaded once;
    the extra MFMA work is free on these HBM-bound layers): decoder_0's split-attention conv (32 -> 64, groups 4, 400x400)
    becomes ONE dense conv (2-3x faster), decoder_1's (64 -> 128, groups 4, 200x200: 16 -> 32 per group) a conv with two
    groups of 32 -> 64 (round 4).  Algorithmic FLOPs are still counted with the real groups."""
    if groups == 1 or (KH, KW, stride, pad) != (3, 3, 1, 1) or os.environ.get("OCTA_NO_DENSIFY") == "1":
        return 0
    ck = 16 if dtype == torch.float32 else 32
    cg, ng = Cin // groups, Cout // groups
    if not (Cin % ck == 0 and Cout % ck == 0 and H * W >= 128 * 128):
        return 0
    if cg <= 8 and ng <= 16:
        return groups                                   # everything into one dense conv
    if cg == 16 and ng == 32 and groups % 2 == 0 and dtype != torch.float32 and os.environ.get("OCTA_NO_DENSIFY_PAIRS") != "1":
        return 2                                        # pairs: groups / 2 groups of 32 -> 64
    return 0


def raw_conv_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int, pad: int, groups: int, act: int = ACT_NONE,
                 out: Optional[Tensor] = None, stats: Optional["ConvStats"] = None) -> Tensor:
    B, Cin, H, W, Cout, Cin_g, KH, KW, OH, OW = _conv_geometry(x, w, stride, pad)
    if Cin != Cin_g * groups:
        raise OctaError(f"conv2d: input has {Cin} channels, weight expects {Cin_g * groups}")
    x, ldx = _conv_input(x, round8(Cin_g), groups)
    zp = out is None and groups == 1 and Cout % 8 != 0       # the kernel zero-fills the padding channels itself
    y = out if out is not None else nhwc_empty(B, Cout, OH, OW, x.dtype, x.device, pad_written=zp)
    ldy = nhwc_ld(y)
    merge = _densify(groups, Cin, Cout, KH, KW, stride, pad, H, W, x.dtype)
    if merge:
        d = _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups // merge, ldx, ldy, x.dtype, act)
        d.alg_groups = groups
        wp = _packed(w, "fwd_dense", x.dtype, groups, d.cin_g_pad)       # pad_to = merge * Cin/groups: the merged set's dense inner axis
    else:
        d = _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups, ldx, ldy, x.dtype, act)
        d.zero_pad = int(zp)
        wp = _packed(w, "fwd", x.dtype, groups, d.cin_g_pad)
    if stats is not None and (x.dtype == torch.float32 or act != ACT_NONE):
        stats = None
    _launch_fwd(d, x, wp, bias, y, stats)
    _record("fwd", d, (_p(x), _p(wp), _p(bias), _p(y)), (x, wp, bias, y))
    return y


_COL2IM_TAPS = os.environ.get("OCTA_NO_COL2IM_TAPS") is None


def raw_conv_dgrad(dy: Tensor, w: Tensor, xshape, stride: int, pad: int, groups: int, addend: Optional[Tensor] = None, gate=None):
    """Data gradient of a conv.  `addend` (a tensor of x's shape) is summed into the result: in the kernel's epilogue on
    the direct path, as a separate add on the GEMM + col2im / densified ones.
    `gate` = (x, act, channels): x (the conv's saved input) is the output of activation `act` on its first `channels` channels
    (0: all) and the caller wants dx * f'(x); the return value is then (dx, applied) -- applied False where the path that ran has no
    gated epilogue and the caller must fall back to octa_act_bwd in the producer."""
    if gate is not None:
        dx, applied = _raw_conv_dgrad(dy, w, xshape, stride, pad, groups, addend, gate)
        return dx, applied
    return _raw_conv_dgrad(dy, w, xshape, stride, pad, groups, addend, None)[0]


def _raw_conv_dgrad(dy: Tensor, w: Tensor, xshape, stride: int, pad: int, groups: int, addend: Optional[Tensor], gate):
    B, Cin, H, W = xshape
    if addend is not None:
        if tuple(addend.shape) != tuple(xshape):
            raise OctaError(f"raw_conv_dgrad: addend shape {tuple(addend.shape)} != input shape {tuple(xshape)}")
        direct = not (stride > 1 and groups == 1 and (Cin * w.shape[2] * w.shape[3]) % 8 == 0) and \
            not _densify(groups, Cin, w.shape[0], w.shape[2], w.shape[3], stride, pad, H, W, dy.dtype)
        if not direct:
            return raw_conv_dgrad(dy, w, xshape, stride, pad, groups) + addend.to(dy.dtype), False
        addend = to_nhwc(addend, dtype=dy.dtype)
        gate = None                       # (no epilogue takes both)
    Cout, Cin_g, KH, KW = w.shape
    OH, OW = dy.shape[2], dy.shape[3]
    need = round8(Cout // groups) if groups == 1 else Cout
    dy = to_nhwc(dy, cpad=need)
    ldy = nhwc_ld(dy)
    if ldy < need:
        dy = to_nhwc(dy.contiguous(), cpad=need)
        ldy = nhwc_ld(dy)
    col2im = stride > 1 and groups == 1 and (Cin * KH * KW) % 8 == 0
    zp = groups == 1 and Cin % 8 != 0 and not col2im
    dx = nhwc_empty(B, Cin, H, W, dy.dtype, dy.device, pad_written=zp or col2im)      # octa_col2im stores the pad channels too
    merge = _densify(groups, Cin, Cout, KH, KW, stride, pad, H, W, dy.dtype)
    if merge:
        d = _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups // merge, nhwc_ld(dx), ldy, dy.dtype)
        d.alg_groups = groups
        wt = _packed(w, "dgrad_dense", dy.dtype, groups, d.cout_g_pad)
        _launch_dgrad(d, dy, wt, dx)
        _record("dgrad", d, (_p(dy), _p(wt), _p(dx)), (dy, wt, dx))
        return dx, False
    gop = _gate_operand(gate, dy.dtype)
    d = _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups, nhwc_ld(dx), ldy, dy.dtype)
    if col2im and Cin <= 16 and _COL2IM_TAPS:
        # few input channels (the discriminator's 15-channel inputs): tap-major N axis, the fold moves 16-byte channel vectors
        cp = round8(Cin)
        N = KH * KW * cp
        wt = _packed(w, "dgrad_taps", dy.dtype, 1, d.cout_g_pad)
        z = nhwc_empty(B, N, OH, OW, dy.dtype, dy.device)
        dz = _desc(B, OH, OW, OH, OW, d.cout_g_pad, N, 1, 1, 1, 0, 1, ldy, N, dy.dtype)
        _launch_fwd(dz, dy, wt, None, z)
        _record("fwd", dz, (_p(dy), _p(wt), None, _p(z)), (dy, wt, z))
        if gop is not None and gop[1] >= cp and gop[1] % 8 == 0:
            gx, ldg, gact, gch = gop
            lib().octa_col2im_taps_gated(_p(z), N, _p(dx), nhwc_ld(dx), B, H, W, OH, OW, cp, KH, KW, stride, pad, _dt(dy), _p(gx), ldg, gact,
                                         gch if gch > 0 else Cin, _st())
            return dx, True
        lib().octa_col2im_taps(_p(z), N, _p(dx), nhwc_ld(dx), B, H, W, OH, OW, cp, KH, KW, stride, pad, _dt(dy), _st())
        return dx, False
    wt = _packed(w, "dgrad", dy.dtype, groups, d.cout_g_pad)
    d.zero_pad = int(zp)
    if col2im:
        # strided data gradient = dense GEMM Z = dy x W^T (no wasted taps, N = Cin*KH*KW) + col2im fold;
        # the data-grad operand [ci][kh][kw][co] is exactly the [N][K] matrix the 1x1 forward wants
        N = Cin * KH * KW
        z = nhwc_empty(B, N, OH, OW, dy.dtype, dy.device)
        dz = _desc(B, OH, OW, OH, OW, d.cout_g_pad, N, 1, 1, 1, 0, 1, ldy, N, dy.dtype)
        _launch_fwd(dz, dy, wt, None, z)
        _record("fwd", dz, (_p(dy), _p(wt), None, _p(z)), (dy, wt, z))
        lib().octa_col2im(_p(z), N, _p(dx), nhwc_ld(dx), B, H, W, OH, OW, Cin, KH, KW, stride, pad, _dt(dy), _st())
        return dx, False
    if gop is not None and addend is None and (gop[3] == 0 or gop[3] >= Cin) and gop[1] >= Cin:
        _launch_dgrad(d, dy, wt, dx, None, gop)
        return dx, True
    _launch_dgrad(d, dy, wt, dx, addend)
    _record("dgrad", d, (_p(dy), _p(wt), _p(dx)), (dy, wt, dx))
    return dx, False


# ----------------------------------------------------------------------------- deferred weight gradients
# Nothing downstream of a conv's backward reads its weight gradient before the optimiser does, so a training step
# may queue the weight-gradient jobs and flush the queue as a few batched launches (octa_conv2d_wgrad_batch):
# the M-split -- and with it the fp32 reduction traffic -- is then chosen over all queued layers at once, and
# ~60 launches per step disappear.  Only jobs that accumulate into a pre-assigned gradient buffer (gradient sink)
# are queued: autograd never sees their result, so nobody can read it before the flush.
_WGRAD_Q = None
_WGRAD_FLUSH_MIN = int(os.environ.get("OCTA_WGRAD_FLUSH_MIN", "16"))      # swept in situ (4 / 8 / 16 / 30 / all at the end): 16 is 0.1 ms faster but its bigger batches miss L2 more (PMC: 6.6 -> 8.4 GB of fetches per step)
_MARK_HOOKS = []


def defer_wgrads(on: bool):
    """Turn the weight-gradient queue on/off (off flushes what is pending)."""
    global _WGRAD_Q
    if on:
        if _WGRAD_Q is None:
            _WGRAD_Q = []
    else:
        flush_wgrads()
        _WGRAD_Q = None


def pending_wgrads() -> int:
    return len(_WGRAD_Q) if _WGRAD_Q else 0


def flush_wgrads(min_jobs: int = 1) -> int:
    """Launch every queued weight gradient (if at least `min_jobs` are pending).  Jobs are grouped by the kernel family
    the library runs them on, so one call = one family (keeps bench.py's per-kernel attribution clean)."""
    q = _WGRAD_Q
    if not q or len(q) < min_jobs:
        return 0
    L = lib()
    groups = {}
    for job in q:
        groups.setdefault(int(L.octa_wgrad_job_class(ctypes.byref(job[0]))), []).append(job)
    st = _st()
    for cls in sorted(groups):
        jobs = groups[cls]
        if cls == 0 and (_RECORD is not None or len(jobs) == 1):
            # (recording tools want one entry per layer; otherwise the single-problem jobs travel as one batch too, so that their
            # partial tiles are folded by ONE launch: octa_wgrad_fold_workspace)
            for j, keep in jobs:
                j.d.ws, j.d.ws_bytes = _fold_ws_args()
                L.octa_conv2d_wgrad(ctypes.byref(j.d), j.x, j.dy, j.dw, j.dw_strides, j.dbias, st)
                _record("wgrad", j.d, (j.x, j.dy, keep[2], keep[3]), keep)
            continue
        arr = (WgradJob * len(jobs))()
        for i, (j, keep) in enumerate(jobs):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(j), ctypes.sizeof(WgradJob))
        L.octa_conv2d_wgrad_batch(arr, len(jobs), *_fold_ws_args(), st)
        if _RECORD is not None:
            _RECORD.append(("wgrad_batch", arr, len(jobs), [k for _, k in jobs]))
    n = len(q)
    q.clear()
    return n


def add_mark_hook(fn, tags=None):
    """fn(tag) is called (after the weight-gradient flush) whenever the backward pass crosses a stage mark; `tags` (optional):
    only at these marks -- the queue of deferred weight gradients is then flushed completely there (a gradient bucket must be
    whole before it travels) and by the usual batch-size rule at the other marks."""
    _MARK_HOOKS.append((fn, None if tags is None else frozenset(tags)))


def clear_mark_hooks():
    _MARK_HOOKS.clear()


class StageMarkFn(Function):
    """Identity that marks a stage boundary of the network: when the BACKWARD pass crosses it, every layer executed after
    the mark in forward has produced (or queued) its parameter gradients.  The queue of deferred weight gradients is
    flushed there and the registered hooks run (bucketed gradient all-reduce, train.py)."""

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        hooks = [fn for fn, tags in _MARK_HOOKS if tags is None or ctx.tag in tags]
        flush_wgrads(1 if hooks else _WGRAD_FLUSH_MIN)
        for fn in hooks:
            fn(ctx.tag)
        return g, None


def stage_mark(x: Tensor, tag: str) -> Tensor:
    if _WGRAD_Q is None and not _MARK_HOOKS:
        return x
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x
    return StageMarkFn.apply(x, tag)


def raw_conv_wgrad(x: Tensor, dy: Tensor, w: Tensor, stride: int, pad: int, groups: int, dw: Optional[Tensor] = None,
                   dbias: Optional[Tensor] = None, defer: bool = False) -> Tensor:
    """dw += wgrad; when `dbias` (fp32 [Cout]) is given the bias gradient is accumulated by the same kernel.
    defer=True (caller owns dw/dbias as gradient-sink buffers): the job may be queued instead of launched."""
    B, Cin, H, W, Cout, Cin_g, KH, KW, OH, OW = _conv_geometry(x, w, stride, pad)
    x, ldx = _conv_input(x, round8(Cin_g), groups)
    need = round8(Cout // groups) if groups == 1 else Cout
    dy = to_nhwc(dy, dtype=x.dtype, cpad=need)
    ldy = nhwc_ld(dy)
    if dw is None:
        # channels-last storage: the kernel's atomics then land in contiguous runs along Cin (17x the
        # rate of a strided OIHW target; MI355X_MICROARCH.md "Global float atomics")
        if w.numel() <= (1 << 18):
            flat, pz = _zeroed_f32((w.numel(),), w.device)
        else:
            flat, pz = torch.empty((w.numel(),), dtype=torch.float32, device=w.device), 0
        if not pz:
            flat.zero_()
        dw = flat.view(Cout, KH, KW, Cin_g).permute(0, 3, 1, 2)
    d = _desc(B, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad, groups, ldx, ldy, x.dtype)
    j = WgradJob()
    ctypes.memmove(ctypes.byref(j.d), ctypes.byref(d), ctypes.sizeof(ConvDesc))
    j.x, j.dy, j.dw, j.dbias = _p(x), _p(dy), _p(dw), _p(dbias)
    for a, sv in enumerate(dw.stride()):
        j.dw_strides[a] = sv
    keep = (x, dy, tuple(dw.shape), tuple(dw.stride()), dw, dbias)
    if defer and _WGRAD_Q is not None and not _IN_SIDE:
        # (a side branch's x / dy live in the side stream's allocator pool: their weight gradient is launched there and then, not
        # parked for a flush on the main stream)
        _WGRAD_Q.append((j, keep))
        return dw
    L = lib()
    # (a non-deferrable job that is wide enough for the 8-wave kernel could go there as a batch of one: measured 0.07 ms per
    # step for the four 15-channel discriminator convs, at the price of six 30 us launches in that kernel's statistics - not taken)
    d.ws, d.ws_bytes = _fold_ws_args()
    L.octa_conv2d_wgrad(ctypes.byref(d), _p(x), _p(dy), _p(dw), _strides4(dw), _p(dbias), _st())
    _record("wgrad", d, (_p(x), _p(dy), tuple(dw.shape), tuple(dw.stride())), (x, dy))
    return dw


def raw_colsum(t: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """sum over (B,H,W) of an NHWC tensor -> fp32 [C]."""
    t = to_nhwc(t)
    B, C, H, W = t.shape
    if out is None:
        out = torch.zeros((C,), dtype=torch.float32, device=t.device)
    part = torch.empty((int(lib().octa_colsum_workspace_floats(C)),), dtype=torch.float32, device=t.device) if B * H * W >= (1 << 16) else None
    lib().octa_colsum(_p(t), B * H * W, C, nhwc_ld(t), 0, _dt(t), _p(out), _p(part), _st())
    return out


def raw_act_bwd(y: Tensor, dy: Tensor, act: int) -> Tensor:
    y = dense_or_same(y)
    dy = to_nhwc(dy, dtype=y.dtype)
    if nhwc_ld(dy) != nhwc_ld(y):
        dy = _match_ld(dy, nhwc_ld(y))
    B, C, H, W = y.shape
    ld = nhwc_ld(y)
    dx = nhwc_empty(B, C, H, W, y.dtype, y.device, pad_written=True)      # all ld channels are computed (dy's padding is zero)
    lib().octa_act_bwd(_p(y), _p(dy), _p(dx), B * H * W * ld, act, _dt(y), _st())
    return dx


def dense_or_same(t: Tensor) -> Tensor:
    return to_nhwc(t)


def _match_ld(t: Tensor, ld: int) -> Tensor:
    """Copy an NHWC tensor into a fresh buffer with per-pixel stride `ld` (pad zeroed)."""
    B, C, H, W = t.shape
    out = torch.empty((B, H, W, ld), dtype=t.dtype, device=t.device).permute(0, 3, 1, 2)[:, :C]
    src = to_nchw_f32(t)
    lib().octa_nchw_to_nhwc(_p(src), src.stride(0), src.stride(1), src.stride(2), src.stride(3), _p(out), B, C, H, W, ld, 0, ld,
                            _dt(t), _st())
    return out


# ----------------------------------------------------------------------------- raw BN
def _bn_ws(rows: int, C: int, device) -> Tensor:
    n = lib().octa_bn_workspace_floats(rows, C)
    return torch.empty((n,), dtype=torch.float32, device=device)


def raw_bn_fwd(x: Tensor, gamma: Tensor, beta: Tensor, rm: Optional[Tensor], rv: Optional[Tensor], momentum: float, eps: float,
               training: bool, relu: bool, residual: Optional[Tensor] = None, pre_sums: Optional["ConvStats"] = None):
    x = to_nhwc(x)
    B, C, H, W = x.shape
    rows = B * H * W
    L = lib()
    y = nhwc_empty(B, C, H, W, x.dtype, x.device)
    res = to_nhwc(residual, dtype=x.dtype) if residual is not None else None
    # 1 bit per element ReLU mask for the backward pass (read instead of y: 1/16 of the bytes in both backward kernels)
    mask = torch.empty(((rows * (C // (4 if x.dtype == torch.float32 else 8)) + 3) // 4 * 4,), dtype=torch.uint8, device=x.device) if (relu and training) else None
    if training:
        if rows <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        if pre_sums is not None and pre_sums.fused and pre_sums.shift is rm:
            # the producing conv summed its own output: replica merge + apply, no statistics pass over x
            L.octa_bn_train_fwd_sums(_p(x), nhwc_ld(x), 0, _p(pre_sums.sums), pre_sums.replicas, _p(gamma), _p(beta), _p(res),
                                     nhwc_ld(res) if res is not None else 0, 0, _p(y), nhwc_ld(y), 0, rows, C, _dt(x), eps, momentum, int(relu),
                                     _p(mean), _p(invstd), _p(rm), _p(rv), _p(mask), _st())
            if rm is not None:
                bump_param_epoch()
            return y, mean, invstd, x, mask
        L.octa_bn_train_fwd(_p(x), nhwc_ld(x), 0, _p(gamma), _p(beta), _p(res), nhwc_ld(res) if res is not None else 0, 0,
                            _p(y), nhwc_ld(y), 0, rows, C, _dt(x), eps, momentum, int(relu), _p(mean), _p(invstd), _p(rm), _p(rv),
                            _p(mask), _p(_bn_ws(rows, C, x.device)), _st())
        if rm is not None:
            bump_param_epoch()           # running statistics moved behind their version counters
        return y, mean, invstd, x, mask
    mean = rm.float()
    invstd = torch.rsqrt(rv.float() + eps)
    L.octa_bn_apply(_p(x), nhwc_ld(x), 0, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), nhwc_ld(res) if res is not None else 0, 0,
                    _p(y), nhwc_ld(y), 0, rows, C, _dt(x), int(relu), _p(mask), _st())
    return y, mean, invstd, x, mask


def raw_bn_apply_only(x: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, beta: Tensor, relu: bool, residual: Optional[Tensor]) -> Tensor:
    """y = (x - mean) * invstd * gamma + beta [+ residual] [-> relu] with GIVEN per-channel vectors (inference)."""
    x = to_nhwc(x)
    B, C, H, W = x.shape
    y = nhwc_empty(B, C, H, W, x.dtype, x.device)
    res = to_nhwc(residual, dtype=x.dtype) if residual is not None else None
    lib().octa_bn_apply(_p(x), nhwc_ld(x), 0, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), nhwc_ld(res) if res is not None else 0, 0,
                        _p(y), nhwc_ld(y), 0, B * H * W, C, _dt(x), int(relu), None, _st())
    return y


def raw_bn_bwd(dy: Tensor, x: Tensor, y: Optional[Tensor], mean: Tensor, invstd: Tensor, gamma: Tensor, relu: bool,
               want_dres: bool, dgamma: Tensor, dbeta: Tensor, mask: Optional[Tensor] = None):
    B, C, H, W = x.shape
    rows = B * H * W
    dy = to_nhwc(dy, dtype=x.dtype)
    dx = nhwc_empty(B, C, H, W, x.dtype, x.device)
    dres = nhwc_empty(B, C, H, W, x.dtype, x.device) if want_dres else None
    lib().octa_bn_bwd(_p(dy), nhwc_ld(dy), 0, _p(x), nhwc_ld(x), 0, _p(y), nhwc_ld(y) if y is not None else 0, 0, _p(mean), _p(invstd),
                      _p(gamma), _p(dx), nhwc_ld(dx), 0, _p(dres), nhwc_ld(dres) if dres is not None else 0, 0, _p(dgamma), _p(dbeta),
                      rows, C, _dt(x), int(relu), _p(mask), _p(_bn_ws(rows, C, x.device)), _st())
    return dx, dres


# ----------------------------------------------------------------------------- parameter-gradient sink
_GRAD_SINK = False


def set_grad_sink(on: bool):
    """When on, backward kernels accumulate parameter gradients DIRECTLY into an existing fp32
    ``param.grad`` (e.g. a view of a flat gradient arena, zeroed once per step) and return None to
    autograd, instead of materialising a fresh tensor per parameter.  Every weight/bias/affine
    gradient kernel of the library has += semantics, so this is exact."""
    global _GRAD_SINK
    _GRAD_SINK = bool(on)


def _sink(p: Optional[Tensor]) -> Optional[Tensor]:
    if not _GRAD_SINK or p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or not g.is_cuda:
        return None
    return g


def _ret(p: Tensor, buf: Tensor) -> Optional[Tensor]:
    """What backward hands to autograd for parameter p whose gradient was written into buf."""
    return None if (_GRAD_SINK and p.is_leaf and p.grad is buf) else buf


# ============================================================================= autograd Functions
class ActGate:
    """Shared by a conv with a fused activation (the producer) and the ONE op that consumes its output.  The consumer's backward
    multiplies its data gradient by the activation's derivative in its own epilogue (octa_conv2d_dgrad_gated,
    octa_col2im_taps_gated, octa_fullconv_bwd_gated: it holds the activation's output anyway, as its saved input) and sets
    `done`; the producer's backward then skips its derivative kernel (octa_act_bwd).  Only valid when nothing else reads the
    producer's output (the discriminator's chain, blocks.py): a gate is multiplicative, so EVERY contribution would have to be gated."""
    __slots__ = ("done",)

    def __init__(self):
        self.done = False


_ACT_GATES = os.environ.get("OCTA_NO_ACT_GATE") is None


def act_gates_enabled() -> bool:
    return _ACT_GATES


def _gate_operand(gate, dy_dtype):
    """(tensor, ld, act, channels) of a gate request if the consumer's saved input can serve as the gate as it is (NHWC, compute dtype)."""
    if gate is None or not _ACT_GATES:
        return None
    gx, gact, gch = gate
    if gx is None or gx.dtype != dy_dtype or gx.dim() != 4 or gx.stride(1) != 1:
        return None
    return gx, nhwc_ld(gx), int(gact), int(gch)


class GradHolder:
    """Carries the gradient of one consumer of a tensor to the conv that consumes the same tensor, so that the conv's data
    gradient can add it in its epilogue (fan-out gradient sum without autograd's separate add kernel)."""
    __slots__ = ("grad", "consumed", "event", "taken")

    def __init__(self):
        self.grad = None
        self.consumed = False
        self.event = None           # recorded behind the parked gradient: the consumer may run on another stream (side branches)
        self.taken = False          # a backward pass that adds the parked gradient has been wired to this holder (offer_fanout)


class StashGradFn(Function):
    """Identity.  Backward parks the incoming gradient in the holder and reports None; the conv sharing the holder adds it
    to its own data gradient.  If that conv's backward already ran (unexpected graph order), the gradient flows normally."""

    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        h = ctx.holder
        if h.consumed:
            return g, None
        h.grad = g if h.grad is None else h.grad + g
        if _SIDE_BRANCH and g.is_cuda:
            h.event = torch.cuda.Event()
            h.event.record()
        return None, None


def stash_grad(x: Tensor, holder: GradHolder) -> Tensor:
    return StashGradFn.apply(x, holder)


# U-Net skip connections (compose.py:141-147): x_k feeds the decoder's cat and the next encoder stage.  The encoder builds the stage
# first, so it OFFERS a holder before calling the stage; the stage's first pooling op on x_k (the avg_down shortcut's pool, the stem's
# max-pool) takes the offer and adds whatever is parked there in its backward kernel (octa_*pool*_bwd_add); the decoder then parks the
# cat's gradient slice through stash_grad -- only if the offer was taken (holder.taken), otherwise it uses x_k as is.
_FUSE_FANOUT_SKIP = os.environ.get("OCTA_FUSE_FANOUT_SKIP", "1") != "0"
_FANOUT_TLS = threading.local()


def offer_fanout(x: Tensor) -> Optional[GradHolder]:
    """A holder for the gradient of another consumer of `x`, offered to the next pooling op that asks (take_fanout); None when
    nothing would be differentiated."""
    if not (_FUSE_FANOUT_SKIP and torch.is_grad_enabled() and x.requires_grad and x.is_cuda):
        _FANOUT_TLS.offer = None
        return None
    h = GradHolder()
    _FANOUT_TLS.offer = h
    return h


def take_fanout() -> Optional[GradHolder]:
    h = getattr(_FANOUT_TLS, "offer", None)
    _FANOUT_TLS.offer = None
    if h is not None and _IN_SIDE:
        return None                    # (a pool on a side stream: the parked gradient would need an event of its own -- not wired)
    if h is not None:
        h.taken = True
    return h


def withdraw_fanout():
    _FANOUT_TLS.offer = None


def skip_with_fanout(x: Tensor, holder: Optional[GradHolder]) -> Tensor:
    """x as the decoder's cat should consume it: through stash_grad when a pooling backward adds the parked gradient."""
    return stash_grad(x, holder) if (holder is not None and holder.taken) else x


def _fanout_addend(holder: Optional[GradHolder], dy: Tensor, shape):
    """(addend usable by the *_bwd_add kernels or None, gradient to add afterwards with ATen or None)"""
    if holder is None:
        return None, None
    holder.consumed = True
    g, holder.grad = holder.grad, None
    if g is None:
        return None, None
    if holder.event is not None:
        torch.cuda.current_stream().wait_event(holder.event)
    if g.dtype == dy.dtype and tuple(g.shape) == tuple(shape) and nhwc_ld(g) is not None:
        return g, None
    return None, g


class Conv2dFn(Function):
    """nn.Conv2d (+ fused activation).  Weight OIHW-logical fp32 parameter, any strides."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, groups, act, holder=None, stats=None, gate_out=None, gate_in=None):
        _require_gpu(x)
        y = raw_conv_fwd(x, w, bias, stride, pad, groups, act, stats=stats)
        ctx.holder = holder
        ctx.gate_out = gate_out          # ActGate of THIS conv's fused activation (its consumer may apply the derivative)
        ctx.gate_in = gate_in            # (ActGate, act, channels) of the activation that produced x: this conv's dgrad applies it
        ctx.side = _IN_SIDE              # a side-branch conv: its backward launches must not use the tail-split scratch either
        ctx.cfg = (stride, pad, groups, act, tuple(x.shape))
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.w_pre = getattr(w, "_octa_packed", None)       # operands that came with the weight (saved tensors come back as new objects)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        if ctx.w_pre is not None:
            w._octa_packed = ctx.w_pre
        stride, pad, groups, act, xshape = ctx.cfg
        if act != ACT_NONE and not (ctx.gate_out is not None and ctx.gate_out.done):
            dy = raw_act_bwd(y, dy, act)         # (gate done: the consumer's data gradient already carries f'(y))
        addend = None
        if ctx.holder is not None:
            ctx.holder.consumed = True
            addend, ctx.holder.grad = ctx.holder.grad, None
            if addend is not None and ctx.holder.event is not None:
                torch.cuda.current_stream().wait_event(ctx.holder.event)      # parked on another stream (side branches)
        gate = None
        if ctx.gate_in is not None and ctx.gate_in[0] is not None and addend is None:
            gate = (x, ctx.gate_in[1], ctx.gate_in[2])
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.side:
                with no_splitk():
                    dx = raw_conv_dgrad(dy, w, xshape, stride, pad, groups, addend)
            elif gate is not None:
                dx, applied = raw_conv_dgrad(dy, w, xshape, stride, pad, groups, None, gate=gate)
                if applied:
                    ctx.gate_in[0].done = True
            else:
                dx = raw_conv_dgrad(dy, w, xshape, stride, pad, groups, addend)
        dw = db = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        b = ctx.bias_ref
        if want_b:
            db = _sink(b)
            if db is None:
                db = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device)
        if ctx.needs_input_grad[1]:
            sw = _sink(w)
            can_defer = sw is not None and (not want_b or _sink(b) is db)      # every target is a gradient-sink buffer
            dw = _ret(w, raw_conv_wgrad(x, dy, w, stride, pad, groups, sw, db if want_b else None, defer=can_defer))   # bias gradient fused
        elif want_b:
            raw_colsum(dy, db)
        if want_b:
            db = _ret(b, db)
        return dx, dw, db, None, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, groups=1, act=ACT_NONE, grad_holder=None, stats=None, gate_out=None, gate_in=None):
    return Conv2dFn.apply(x, w, bias, stride, pad, groups, act, grad_holder, stats, gate_out, gate_in)


class ConvTranspose2x2Fn(Function):
    """nn.ConvTranspose2d(k=2, s=2) + bias as an up-shuffle GEMM (extra/resnest.py:50)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _require_gpu(x)
        B, Cin, H, W = x.shape
        CinT, CoutT = w.shape[0], w.shape[1]
        if Cin != CinT or tuple(w.shape[2:]) != (2, 2):
            raise OctaError("conv_transpose2x2: weight must be (Cin, Cout, 2, 2)")
        x, y = _conv_transpose2x2_into(x, w, bias, None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        return _conv_transpose2x2_bwd(x, w, ctx.bias_ref if ctx.has_bias else None, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                      ctx.has_bias and ctx.needs_input_grad[2])


def _conv_transpose2x2_into(x, w, bias, y):
    """The up-shuffle GEMM of a ConvTranspose2d(k=2, s=2); `y`: NHWC (B, CoutT, 2H, 2W) destination (may be a channel slice of a wider
    buffer: the kernels take its per-pixel stride), allocated here when None.  Returns (x as launched, y)."""
    B, Cin, H, W = x.shape
    CoutT = w.shape[1]
    x, ldx = _conv_input(x, round8(Cin), 1)
    if y is None:
        y = nhwc_empty(B, CoutT, 2 * H, 2 * W, x.dtype, x.device)
    d = _desc(B, H, W, H, W, Cin, 4 * CoutT, 1, 1, 1, 0, 1, ldx, nhwc_ld(y), x.dtype, ACT_NONE, upshuffle=1)
    wp = _packed(w, "convT", x.dtype, 1, d.cin_g_pad)
    _launch_fwd(d, x, wp, bias, y)
    _record("fwd", d, (_p(x), _p(wp), _p(bias), _p(y)), (x, wp, bias, y))
    return x, y


def _conv_transpose2x2_bwd(x, w, bias, dy, need_x, need_w, need_b):
    # the adjoint is a plain conv k2 s2 p0 from the (2H,2W,CoutT) image to (H,W,CinT) whose OIHW weight is w itself
    dx = raw_conv_fwd(dy, w, None, 2, 0, 1) if need_x else None
    dw = db = None
    if need_w:
        sw = _sink(w)
        dw = _ret(w, raw_conv_wgrad(to_nhwc(dy, dtype=x.dtype), x, w, 2, 0, 1, sw, defer=sw is not None))
    if need_b:
        db = _ret(bias, raw_colsum(dy, _sink(bias)))
    return dx, dw, db


def conv_transpose2x2(x, w, bias):
    return ConvTranspose2x2Fn.apply(x, w, bias)


class UpCatFn(Function):
    """torch.cat((skip, ConvTranspose2d(k=2, s=2)(x)), dim=1) without a crop (compose.py:141-147 with extra/resnest.py:50): the up-shuffle
    GEMM stores straight into its channel slice of the cat buffer (per-pixel stride = the cat's channel count), so the copy of the
    up-sampled half -- 41 / 82 / 82 MB each way at 50 x 50 / 100 x 100 / 200 x 200 of the B = 16, 400 x 400 step -- does not happen.
    The backward pass is CatFn's (zero-copy channel slices) followed by ConvTranspose2x2Fn's."""

    @staticmethod
    def forward(ctx, skip, x, w, bias):
        _require_gpu(x)
        B, Cin, H, W = x.shape
        if Cin != w.shape[0] or tuple(w.shape[2:]) != (2, 2):
            raise OctaError("conv_transpose2x2: weight must be (Cin, Cout, 2, 2)")
        Cb = w.shape[1]
        a = to_nhwc(skip)
        Ca = a.shape[1]
        if tuple(a.shape[2:]) != (2 * H, 2 * W) or Ca % 8 or Cb % 8:
            raise OctaError("upsample_cat: the skip tensor must be (B, 8k, 2H, 2W) and the up-sampled channel count a multiple of 8")
        out = nhwc_empty(B, Ca + Cb, 2 * H, 2 * W, a.dtype, a.device)
        lib().octa_copy_channels(_p(a), 2 * H, 2 * W, nhwc_ld(a), 0, _p(out), 2 * H, 2 * W, Ca + Cb, 0, B, Ca, _dt(a), 0, _st())
        xl, _ = _conv_transpose2x2_into(x, w, bias, out[:, Ca:])
        ctx.save_for_backward(xl, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.Ca = Ca
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        x, w = ctx.saved_tensors
        d = to_nhwc(d)
        Ca = ctx.Ca
        dx, dw, db = _conv_transpose2x2_bwd(x, w, ctx.bias_ref if ctx.has_bias else None, d[:, Ca:], ctx.needs_input_grad[1],
                                            ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3])
        return (d[:, :Ca] if ctx.needs_input_grad[0] else None), dx, dw, db


_FUSE_UPCAT = os.environ.get("OCTA_FUSE_UPCAT", "1") != "0"


def upsample_cat(skip, x, w, bias, Hc=None, Wc=None):
    """cat_crop(skip, conv_transpose2x2(x, w, bias), Hc, Wc); one launch fewer and no copy of the up-sampled half when nothing is
    cropped (OCTA_FUSE_UPCAT=0: the two separate ops, the parity twin)."""
    Hc = skip.shape[2] if Hc is None else Hc
    Wc = skip.shape[3] if Wc is None else Wc
    fits = (_FUSE_UPCAT and x.is_cuda and (Hc, Wc) == tuple(skip.shape[2:]) == (2 * x.shape[2], 2 * x.shape[3])
            and skip.shape[1] % 8 == 0 and w.shape[1] % 8 == 0 and skip.dtype == x.dtype and nhwc_ld(skip) is not None)
    if not fits:
        return cat_crop(skip, conv_transpose2x2(x, w, bias), Hc, Wc)
    return UpCatFn.apply(skip, x, w, bias)


class BatchNormFn(Function):
    """Training/eval BatchNorm2d with fused ReLU and residual add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, momentum, eps, training, relu, residual, pre_sums=None):
        _require_gpu(x)
        y, mean, invstd, xn, mask = raw_bn_fwd(x, gamma, beta, rm, rv, momentum, eps, training, relu, residual, pre_sums)
        ctx.relu, ctx.training, ctx.has_res = relu, training, residual is not None
        ctx.beta_ref = beta
        ctx.save_for_backward(xn, y if (relu and mask is None) else None, mean, invstd, gamma, mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, mean, invstd, gamma, mask = ctx.saved_tensors
        if not ctx.training:
            raise OctaError("BatchNorm backward in eval mode is not part of the hot path")
        beta = ctx.beta_ref
        dgamma = _sink(gamma)
        dbeta = _sink(beta)
        if dgamma is None or dbeta is None:
            dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
        dx, dres = raw_bn_bwd(dy, x, y, mean, invstd, gamma, ctx.relu, ctx.has_res and ctx.needs_input_grad[9], dgamma, dbeta, mask)
        return dx, _ret(gamma, dgamma), _ret(beta, dbeta), None, None, None, None, None, None, dres, None


def batch_norm(x, gamma, beta, rm, rv, momentum=0.1, eps=1e-5, training=True, relu=False, residual=None, pre_sums=None):
    return BatchNormFn.apply(x, gamma, beta, rm, rv, momentum, eps, training, relu, residual, pre_sums)


class MaxPool3s2Fn(Function):
    @staticmethod
    def forward(ctx, x, holder=None):
        ctx.holder = holder
        x = dense_nhwc(x)
        B, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = nhwc_empty(B, C, OH, OW, x.dtype, x.device)
        am = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device)
        lib().octa_maxpool3s2_fwd(_p(x), _p(y), _p(am), B, H, W, C, OH, OW, _dt(x), _st())
        ctx.save_for_backward(am)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy = dense_nhwc(dy)
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        add, late = _fanout_addend(ctx.holder, dy, (B, C, H, W))
        lib().octa_maxpool3s2_bwd_add(_p(dy), _p(am), _p(dx), _p(add), nhwc_ld(add) if add is not None else 0, B, H, W, C, dy.shape[2], dy.shape[3],
                                      _dt(dy), _st())
        return (dx if late is None else dx + late), None


def max_pool3s2(x, fanout: Optional[GradHolder] = None):
    return MaxPool3s2Fn.apply(x, fanout)


class AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, ceil_mode, count_include_pad, holder=None):
        ctx.holder = holder
        x = dense_nhwc(x)
        B, C, H, W = x.shape

        def osz(L):
            num = L + 2 * pad - k
            o = (-(-num // stride) if ceil_mode else num // stride) + 1
            if ceil_mode and (o - 1) * stride >= L + pad:
                o -= 1
            return o
        OH, OW = osz(H), osz(W)
        y = nhwc_empty(B, C, OH, OW, x.dtype, x.device)
        lib().octa_avgpool_fwd(_p(x), _p(y), B, H, W, C, OH, OW, k, stride, pad, int(count_include_pad), _dt(x), _st())
        ctx.cfg = (B, C, H, W, OH, OW, k, stride, pad, int(count_include_pad))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W, OH, OW, k, stride, pad, cip = ctx.cfg
        dy = dense_nhwc(dy)
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        add, late = _fanout_addend(ctx.holder, dy, (B, C, H, W))
        lib().octa_avgpool_bwd_add(_p(dy), _p(dx), _p(add), nhwc_ld(add) if add is not None else 0, B, H, W, C, OH, OW, k, stride, pad, cip,
                                   _dt(dy), _st())
        return (dx if late is None else dx + late), None, None, None, None, None, None


def avg_pool(x, k, stride, pad=0, ceil_mode=False, count_include_pad=True, fanout: Optional[GradHolder] = None):
    return AvgPoolFn.apply(x, k, stride, pad, ceil_mode, count_include_pad, fanout)


class CatFn(Function):
    """torch.cat((a, b), dim=1) followed by an optional bottom/right crop (compose.py:141-147)."""

    @staticmethod
    def forward(ctx, a, b, Hc, Wc):
        a, b = to_nhwc(a), to_nhwc(b, dtype=a.dtype)
        B, Ca, Ha, Wa = a.shape
        Cb = b.shape[1]
        if Ca % 8 or Cb % 8:
            raise OctaError("cat: channel counts must be multiples of 8")
        out = nhwc_empty(B, Ca + Cb, Hc, Wc, a.dtype, a.device)
        L = lib()
        L.octa_copy_channels(_p(a), Ha, Wa, nhwc_ld(a), 0, _p(out), Hc, Wc, Ca + Cb, 0, B, Ca, _dt(a), 0, _st())
        L.octa_copy_channels(_p(b), b.shape[2], b.shape[3], nhwc_ld(b), 0, _p(out[:, Ca:]), Hc, Wc, Ca + Cb, 0, B, Cb, _dt(a), 0, _st())
        ctx.cfg = (Ca, Cb, tuple(a.shape), tuple(b.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        Ca, Cb, sa, sb = ctx.cfg
        d = to_nhwc(d)
        B, _, Hc, Wc = d.shape
        ld = nhwc_ld(d)

        def part(off, C, shape):
            if shape[2] == Hc and shape[3] == Wc:
                return d[:, off:off + C]                       # zero-copy channel slice
            g = nhwc_empty(B, C, shape[2], shape[3], d.dtype, d.device)   # un-crop: zero fill
            lib().octa_copy_channels(_p(d[:, off:off + C]), Hc, Wc, ld, 0, _p(g), shape[2], shape[3], C, 0, B, C, _dt(d), 0, _st())
            return g
        return part(0, Ca, sa), part(Ca, Cb, sb), None, None


def cat_crop(a, b, Hc=None, Wc=None):
    Hc = a.shape[2] if Hc is None else Hc
    Wc = a.shape[3] if Wc is None else Wc
    return CatFn.apply(a, b, Hc, Wc)


class PadBRFn(Function):
    """F.pad(x, (0, pw, 0, ph)) with zeros (compose.py:125-130)."""

    @staticmethod
    def forward(ctx, x, ph, pw):
        x = to_nhwc(x)
        B, C, H, W = x.shape
        out = nhwc_empty(B, C, H + ph, W + pw, x.dtype, x.device)
        lib().octa_copy_channels(_p(x), H, W, nhwc_ld(x), 0, _p(out), H + ph, W + pw, C, 0, B, C, _dt(x), 0, _st())
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        B, C, H, W = ctx.shape
        d = to_nhwc(d)
        g = nhwc_empty(B, C, H, W, d.dtype, d.device)
        lib().octa_copy_channels(_p(d), d.shape[2], d.shape[3], nhwc_ld(d), 0, _p(g), H, W, C, 0, B, C, _dt(d), 0, _st())
        return g, None, None


def pad_bottom_right(x, ph, pw):
    return PadBRFn.apply(x, ph, pw)


def _dense2d(w: Tensor) -> Optional[Tensor]:
    """(O, I, 1, 1) conv weight (or its grad) as a dense [O][I] fp32 view, or None if its memory is not that."""
    O, I = w.shape[0], w.shape[1]
    if w.dtype != torch.float32 or w.stride(1) != 1 or (O > 1 and w.stride(0) != I):
        return None
    return w


def _grad_buf(p: Optional[Tensor], dense2d: bool = False):
    """fp32 accumulation target for parameter p: its pre-assigned .grad (gradient sink) or fresh zeros."""
    if p is None:
        return None, None
    g = _sink(p)
    if g is not None and (not dense2d or _dense2d(g) is not None):
        return g, None
    z = torch.zeros(tuple(p.shape), dtype=torch.float32, device=p.device)
    return z, z


_SPLAT_BWD_MERGE = os.environ.get("OCTA_SPLAT_BWD_MERGE", "1") != "0"      # bn0's backward sums ride along the logits pass (2 passes instead of 3)
# radix softmax backward inside the micro-net backward (octa_splat_bn_bwd_da2 + octa_splat_mlp_bwd_da: no launch of its own, 21 launches fewer per
# step).  Built and parity-tested at the end of round 5, measured +-0 on the replayed step (profiles/r05_ab_splat_softmax_inline.txt: the three extra
# loads and the exp per value cost the micro-net backward what the 4.7 us launch cost): off by default
_SPLAT_SOFTMAX_INLINE = os.environ.get("OCTA_SPLAT_SOFTMAX_INLINE", "0") == "1"


class SplatTailFn(Function):
    """Everything of SplAtConv2d.forward after bn0+relu (extra/resnest.py:106-138), radix 2:
    radix-sum GAP -> fc1 -> bn1 -> relu -> fc2 -> radix softmax -> weighted sum [-> relu].
    One Function so that the backward can run its two streaming passes around the micro-net, which itself is
    two (forward) / four (backward) small exact-fp32 kernels.
    With bn0's parameters (g0 ...; training only) `xr` is the RAW conv output and bn0 + ReLU (resnest.py:100-103) are part of the
    op: batch statistics first (octa_bn_stats), then every streaming kernel recomputes y = relu(bn0(x)) in registers
    (octa_splat_bn_*): y and its gradient never exist in memory."""

    @staticmethod
    def forward(ctx, xr, fc1_w, fc1_b, g1, b1, rm1, rv1, fc2_w, fc2_b, cardinality, momentum, eps, training, relu,
                g0=None, b0=None, rm0=None, rv0=None, momentum0=0.1, eps0=1e-5):
        xr = dense_nhwc(xr)
        B, C2, H, W = xr.shape
        C, HW = C2 // 2, H * W
        inter = fc1_w.shape[0]
        if B > 32:
            raise OctaError(f"split attention: per-GPU batch {B} > 32 is not supported by the fused attention micro-net")
        if training and B < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {(B, inter, 1, 1)}")
        L = lib()
        dev = xr.device
        fused = g0 is not None
        mean0 = invstd0 = None
        if fused:
            if not training:
                raise OctaError("split attention with bn0 on the fly is a training-mode op")
            if B * HW <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(xr.shape)}")
            mean0 = torch.empty((C2,), dtype=torch.float32, device=dev)
            invstd0 = torch.empty_like(mean0)
            L.octa_bn_stats(_p(xr), B * HW, C2, nhwc_ld(xr), 0, _dt(xr), eps0, momentum0, _p(mean0), _p(invstd0), _p(rm0), _p(rv0),
                            _p(_bn_ws(B * HW, C2, dev)), _st())
            if rm0 is not None:
                bump_param_epoch()           # running statistics moved behind their version counters
        gap, pz = _zeroed_f32((B, C), dev)
        if fused:
            L.octa_splat_bn_gap(_p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(gap), B, HW, C, _dt(xr), pz, _st())
        else:
            L.octa_splat_gap(_p(xr), _p(gap), B, HW, C, _dt(xr), pz, _st())
        w1, w2 = _dense2d(fc1_w.detach()), _dense2d(fc2_w.detach())
        if w1 is None:
            w1 = fc1_w.detach().float().contiguous()
        if w2 is None:
            w2 = fc2_w.detach().float().contiguous()
        h1 = torch.empty((B, inter), dtype=torch.float32, device=dev)
        h2 = torch.empty_like(h1)
        mean1 = torch.empty((inter,), dtype=torch.float32, device=dev)
        invstd1 = torch.empty_like(mean1)
        logits = torch.empty((B, C2), dtype=torch.float32, device=dev)
        L.octa_splat_mlp_fwd(_p(gap), _p(w1), _p(fc1_b), _p(g1), _p(b1), _p(rm1), _p(rv1), momentum, eps, int(training), _p(w2), _p(fc2_b),
                             _p(h1), _p(h2), _p(mean1), _p(invstd1), _p(logits), B, C, inter, cardinality, _st())
        out = nhwc_empty(B, C, H, W, xr.dtype, dev)
        if fused:
            L.octa_splat_bn_apply(_p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), B, HW, C, _dt(xr), int(relu), _st())
        else:
            L.octa_splat_apply(_p(xr), _p(logits), _p(out), B, HW, C, _dt(xr), int(relu), _st())
        ctx.cfg = (cardinality, training, relu, B, C, H, W, inter, fused)
        ctx.refs = (fc1_w, fc1_b, g1, b1, fc2_w, fc2_b, g0, b0)
        ctx.save_for_backward(xr, out if relu else None, logits, gap, h1, h2, mean1, invstd1, w1, w2, mean0, invstd0)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        xr, out, logits, gap, h1, h2, mean1, invstd1, w1, w2, mean0, invstd0 = ctx.saved_tensors
        card, training, relu, B, C, H, W, inter, fused = ctx.cfg
        fc1_w, fc1_b, g1, b1, fc2_w, fc2_b, g0, b0 = ctx.refs
        if not training:
            raise OctaError("SplAt backward in eval mode is not part of the hot path")
        HW = H * W
        L = lib()
        dev = xr.device
        dout = dense_nhwc(to_nhwc(dout, dtype=xr.dtype))
        dlogits, pz = _zeroed_f32((B, 2 * C), dev)
        aux = None
        raw_da = False
        if fused and _SPLAT_BWD_MERGE:
            # two passes over (dout, out, x) instead of three: the logits pass also takes bn0's backward sums along (aux), see splat_aag.hip
            aux, pza = _zeroed_f32((B, 8, C), dev)
            # (_SPLAT_SOFTMAX_INLINE: the radix softmax backward is applied by the micro-net backward where it reads the sums -- no launch for it)
            raw_da = _SPLAT_SOFTMAX_INLINE
            logits_pass = L.octa_splat_bn_bwd_da2 if raw_da else L.octa_splat_bn_bwd_logits2
            if pz and pza:
                logits_pass(_p(dout), _p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), _p(dlogits), _p(aux),
                            B, HW, C, _dt(xr), int(relu), 1, _st())
            else:
                if pz:
                    dlogits = torch.empty((B, 2 * C), dtype=torch.float32, device=dev)      # (both buffers are cleared by the entry point)
                logits_pass(_p(dout), _p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), _p(dlogits), _p(aux),
                            B, HW, C, _dt(xr), int(relu), 0, _st())
        elif fused:
            L.octa_splat_bn_bwd_logits(_p(dout), _p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), _p(dlogits), B, HW, C,
                                       _dt(xr), int(relu), pz, _st())
        else:
            L.octa_splat_bwd(_p(dout), _p(xr), _p(logits), _p(out), None, None, _p(dlogits), B, HW, C, _dt(xr), int(relu), 0, pz, _st())
        dw1, r_w1 = _grad_buf(fc1_w, True)
        db1f, r_b1f = _grad_buf(fc1_b)
        dg1, r_g1 = _grad_buf(g1)
        dbe1, r_be1 = _grad_buf(b1)
        dw2, r_w2 = _grad_buf(fc2_w, True)
        db2, r_b2 = _grad_buf(fc2_b)
        dh1 = torch.empty((B, inter), dtype=torch.float32, device=dev)
        dgap, pzg = _zeroed_f32((B, C), dev)
        if raw_da:
            L.octa_splat_mlp_bwd_da(_p(dlogits), _p(logits), _p(gap), _p(w1), _p(w2), _p(h1), _p(h2), _p(mean1), _p(invstd1), _p(g1), _p(dh1), _p(dgap),
                                    _p(dw1), _p(db1f), _p(dg1), _p(dbe1), _p(dw2), _p(db2), B, C, inter, card, pzg, _st())
        else:
            L.octa_splat_mlp_bwd(_p(dlogits), _p(gap), _p(w1), _p(w2), _p(h1), _p(h2), _p(mean1), _p(invstd1), _p(g1), _p(dh1), _p(dgap), _p(dw1),
                                 _p(db1f), _p(dg1), _p(dbe1), _p(dw2), _p(db2), B, C, inter, card, pzg, _st())
        dx = nhwc_empty(B, 2 * C, H, W, xr.dtype, dev)
        r_g0 = r_b0 = None
        if fused:
            dg0, r_g0 = _grad_buf(g0)
            db0, r_b0 = _grad_buf(b0)
            if aux is not None:
                L.octa_splat_bn_bwd_dx2(_p(dout), _p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), _p(dgap), _p(aux), _p(dx),
                                        _p(dg0), _p(db0), _p(_bn_ws(B * HW, 2 * C, dev)), B, HW, C, _dt(xr), int(relu), _st())
            else:
                L.octa_splat_bn_bwd_dx(_p(dout), _p(xr), _p(mean0), _p(invstd0), _p(g0), _p(b0), _p(logits), _p(out), _p(dgap), _p(dx), _p(dg0),
                                       _p(db0), _p(_bn_ws(B * HW, 2 * C, dev)), B, HW, C, _dt(xr), int(relu), _st())
        else:
            L.octa_splat_bwd(_p(dout), None, _p(logits), _p(out), _p(dgap), _p(dx), None, B, HW, C, _dt(xr), int(relu), 1, 0, _st())
        return dx, r_w1, r_b1f, r_g1, r_be1, None, None, r_w2, r_b2, None, None, None, None, None, r_g0, r_b0, None, None, None, None


def splat_tail(xr, fc1_w, fc1_b, g1, b1, rm1, rv1, fc2_w, fc2_b, cardinality, momentum=0.1, eps=1e-5, training=True, relu=False, bn0=None):
    """bn0 = (weight, bias, running_mean, running_var, momentum, eps) of the BatchNorm that precedes the split attention: xr is then
    the raw conv output (training mode)."""
    if bn0 is None:
        return SplatTailFn.apply(xr, fc1_w, fc1_b, g1, b1, rm1, rv1, fc2_w, fc2_b, cardinality, momentum, eps, training, relu)
    return SplatTailFn.apply(xr, fc1_w, fc1_b, g1, b1, rm1, rv1, fc2_w, fc2_b, cardinality, momentum, eps, training, relu, *bn0)


class AagFn(Function):
    """AdversarialAttentionGate.forward (segmentor/blocks.py:38-46) / 1x1 head (compose.py:181)."""

    @staticmethod
    def forward(ctx, x, w, bias, mode):
        x = dense_nhwc(x)
        B, C, H, W = x.shape
        K = w.shape[0]
        w2 = w.detach().reshape(K, C)
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        y = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
        masked = nhwc_empty(B, C, H, W, x.dtype, x.device) if mode == 0 else None
        lib().octa_aag_fwd(_p(x), _p(w2), _p(bias), _p(masked), _p(y), B, H * W, C, K, _dt(x), mode, _st())
        ctx.mode = mode
        ctx.wshape = tuple(w.shape)
        ctx.refs = (w, bias)
        ctx.save_for_backward(x, w2, y if mode == 0 else None)
        if mode == 0:
            return masked, y
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        x, w2, y = ctx.saved_tensors
        B, C, H, W = x.shape
        K = w2.shape[0]
        if ctx.mode == 0:
            dmasked, dy = grads
            if dmasked is None:
                dmasked = torch.zeros_like(x)
            dmasked = dense_nhwc(to_nhwc(dmasked, dtype=x.dtype))
        else:
            dmasked, dy = None, grads[0]
        if dy is not None:
            dy = dy.float().contiguous()
        dx = nhwc_empty(B, C, H, W, x.dtype, x.device)
        wp, bp = ctx.refs
        dw, db = _sink(wp), _sink(bp)
        if dw is None or db is None or dw.stride(0) != C or dw.stride(1) != 1:
            dw = torch.zeros(ctx.wshape, dtype=torch.float32, device=x.device)
            db = torch.zeros((K,), dtype=torch.float32, device=x.device)
        part = torch.empty((int(lib().octa_aag_workspace_floats(C, K)),), dtype=torch.float32, device=x.device)
        lib().octa_aag_bwd(_p(x), _p(w2), _p(y), _p(dmasked), _p(dy), _p(dx), _p(dw), _p(db), B, H * W, C, K, _dt(x), ctx.mode, _p(part), _st())
        return dx, _ret(wp, dw), _ret(bp, db), None


def attention_gate(x, w, bias):
    return AagFn.apply(x, w, bias, 0)


def head_1x1(x, w, bias):
    return AagFn.apply(x, w, bias, 1)


# ----------------------------------------------------------------------------- losses
class WpceDiceFn(Function):
    """out[0] = WeightedPartialCE(manual=True), out[1] = DiceLoss, on probabilities or (fused
    softmax) on logits; segmentor/losses.py:26-61, 70-74."""

    @staticmethod
    def forward(ctx, inp, ys, from_logits, full, reduction_sum):
        _require_gpu(inp)
        if inp.dtype != torch.float32:
            inp = inp.float()
        if ys.dtype != torch.float32:
            ys = ys.float()
        B, K, H, W = inp.shape
        L = lib()
        ws = torch.empty((L.octa_loss_workspace_floats(B, K),), dtype=torch.float32, device=inp.device)
        out = torch.empty((2,), dtype=torch.float32, device=inp.device)
        L.octa_wpce_dice_fwd(_p(inp), _strides4(inp), _p(ys), _strides4(ys), B, K, H, W, int(from_logits), int(full), int(reduction_sum),
                             _p(out), _p(ws), _st())
        ctx.cfg = (int(from_logits), int(full), int(reduction_sum))
        ctx.save_for_backward(inp, ys, ws)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        inp, ys, ws = ctx.saved_tensors
        B, K, H, W = inp.shape
        g = g.float().contiguous()
        din = torch.empty((B, K, H, W), dtype=torch.float32, device=inp.device)
        fl, full, rs = ctx.cfg
        lib().octa_wpce_dice_bwd(_p(inp), _strides4(inp), _p(ys), _strides4(ys), B, K, H, W, fl, full, rs, g.data_ptr(),
                                 g.data_ptr() + 4, _p(ws), _p(din), _strides4(din), _st())
        return din, None, None, None, None


def wpce_dice(inp, ys, from_logits=False, full=False, reduction_sum=False):
    return WpceDiceFn.apply(inp, ys, from_logits, full, reduction_sum)


class ClassSoftmaxFn(Function):
    """nn.Softmax(dim=1) on the (B, classes, H, W) logits (compose.py:192)."""

    @staticmethod
    def forward(ctx, logits):
        _require_gpu(logits)
        if logits.dtype != torch.float32:
            logits = logits.float()
        B, K, H, W = logits.shape
        p = torch.empty((B, K, H, W), dtype=torch.float32, device=logits.device)
        lib().octa_class_softmax_fwd(_p(logits), _strides4(logits), _p(p), B, K, H, W, _st())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    @once_differentiable
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        B, K, H, W = p.shape
        dp = dp.float()
        din = torch.empty_like(p)
        lib().octa_class_softmax_bwd(_p(p), _p(dp), _strides4(dp), _p(din), B, K, H, W, _st())
        return din


def class_softmax(logits):
    return ClassSoftmaxFn.apply(logits)


class InterlayerKLFn(Function):
    """InterlayerDivergence KLD/mean (segmentor/losses.py:111-147) with the nearest up-sampling fused."""

    @staticmethod
    def forward(ctx, weights, stop_gradient, holders, basis, *maps):
        _require_gpu(basis)
        ctx.holders = holders        # per (basis, *maps): GradHolder of another consumer's gradient (added by the backward kernels) or None
        basis = basis.float().contiguous()
        B, K, H, W = basis.shape
        use = [(m.float().contiguous(), float(w)) for m, w in zip(maps, weights) if w != 0]
        shifts = []
        for m, _ in use:
            f = H // m.shape[2]
            if f < 1 or f & (f - 1) or m.shape[2] * f != H or m.shape[3] * f != W:
                raise OctaError(f"interlayer KL: map {tuple(m.shape)} is not a power-of-two reduction of {tuple(basis.shape)}")
            shifts.append(f.bit_length() - 1)
        n = len(use)
        ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m, _ in use])
        sh = (ctypes.c_int * n)(*shifts)
        wt = (ctypes.c_float * n)(*[w for _, w in use])
        wsum = float(sum(weights))
        out = torch.empty((2,), dtype=torch.float32, device=basis.device)
        ws = torch.empty((1024,), dtype=torch.float32, device=basis.device)
        lib().octa_interlayer_kl_fwd(_p(basis), ptrs, sh, wt, n, wsum, B, K, H, W, _p(out), _p(ws), _st())
        ctx.cfg = (shifts, [w for _, w in use], wsum, stop_gradient, [i for i, w in enumerate(weights[:len(maps)]) if w != 0], len(maps))
        ctx.save_for_backward(basis, *[m for m, _ in use])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        basis, *use = ctx.saved_tensors
        shifts, wts, wsum, stop_gradient, idx, nmaps = ctx.cfg
        B, K, H, W = basis.shape
        g0 = g.float().contiguous()
        n = len(use)
        dmaps = [torch.empty_like(m) for m in use]
        ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in use])
        dptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in dmaps])
        sh = (ctypes.c_int * n)(*shifts)
        wt = (ctypes.c_float * n)(*wts)
        dbasis = None if stop_gradient else torch.empty_like(basis)
        # fan-out: what the maps' other consumer (the discriminator's generator pass, whose backward ran first) parked for them
        holders = ctx.holders or [None] * (nmaps + 1)
        parked = [_take_parked(h) for h in holders]

        def addend_of(i, like):
            g_ = parked[i]
            if g_ is not None and like is not None and g_.dtype == torch.float32 and g_.shape == like.shape and g_.is_contiguous():
                parked[i] = None
                return g_
            return None
        badd = addend_of(0, dbasis)
        madds = [addend_of(1 + i, d) for i, d in zip(idx, dmaps)]
        aptrs = (ctypes.c_void_p * n)(*[_p(a) for a in madds])
        lib().octa_interlayer_kl_bwd_add(_p(basis), ptrs, sh, wt, n, wsum, B, K, H, W, _p(g0), _p(dbasis), dptrs, _p(badd), aptrs, _st())
        grads: List[Optional[Tensor]] = [None] * nmaps
        for i, d in zip(idx, dmaps):
            grads[i] = d
        # whatever could not ride as an addend (a map without a KL weight, stop_gradient, another layout) is added / returned here
        if parked[0] is not None:
            dbasis = parked[0] if dbasis is None else dbasis + parked[0]
        for i in range(nmaps):
            if parked[1 + i] is not None:
                grads[i] = parked[1 + i] if grads[i] is None else grads[i] + parked[1 + i]
        return (None, None, None, dbasis, *grads)


def _take_parked(h: Optional["GradHolder"]):
    if h is None:
        return None
    h.consumed = True
    g, h.grad = h.grad, None
    if g is not None and h.event is not None:
        torch.cuda.current_stream().wait_event(h.event)
    return g


_FUSE_KL_FANOUT = os.environ.get("OCTA_FUSE_KL_FANOUT", "1") != "0"


def interlayer_kl(attentions: Sequence[Tensor], weights, stop_gradient=False, holders=None) -> Tensor:
    """returns a (2,) tensor: [loss, nan_flag].  `holders`: one GradHolder (or None) per attention -- the gradient another consumer of
    that map parks there (stash_grad, created AFTER this call so that its backward runs first) is added by the backward kernels."""
    if holders is not None and len(holders) != len(attentions):
        raise OctaError("interlayer_kl: one holder (or None) per attention")
    return InterlayerKLFn.apply(list(weights), stop_gradient, list(holders) if holders is not None else None, attentions[0], *attentions[1:])


class LsganFn(Function):
    @staticmethod
    def forward(ctx, mode, fake, real):
        _require_gpu(fake)
        fake = fake.float().contiguous()
        real = real.float().contiguous() if real is not None else None
        out = torch.empty((1,), dtype=torch.float32, device=fake.device)
        lib().octa_lsgan_fwd(_p(real), _p(fake), real.numel() if real is not None else 0, fake.numel(), mode, _p(out), _st())
        ctx.mode = mode
        ctx.save_for_backward(fake, real)
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        fake, real = ctx.saved_tensors
        g = g.float().contiguous()
        df = torch.empty_like(fake)
        dr = torch.empty_like(real) if real is not None else None
        lib().octa_lsgan_bwd(_p(real), _p(fake), real.numel() if real is not None else 0, fake.numel(), ctx.mode, _p(g), _p(dr), _p(df), _st())
        return None, df, dr


def lsgan_generator(fake):
    return LsganFn.apply(0, fake, None)


class LossCombineFn(Function):
    """total = ((a[0] wa0 + a[1] wa1) + b[0] wb0 + b[1] wb1) + c wc0 and scaled = total x loss scale in ONE launch each way (octa_loss_combine_*): the
    sums of the reference's training step written with tensor operators cost six 4-byte ATen launches forward and a dozen backward (select backward =
    fill + copy, accumulations), all of them on the critical path between the two passes.  Zero-weight slots are skipped.  Returns (total, scaled);
    only `scaled` is differentiable."""

    @staticmethod
    def forward(ctx, weights, scale_dev, scale_host, a, b, c):
        ref = next(t for t in (a, b, c) if t is not None)
        _require_gpu(ref)
        for t, n in ((a, 2), (b, 2), (c, 1)):
            if t is not None and (t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous()):
                raise OctaError("loss_combine: terms are contiguous fp32 tensors of 2, 2 and 1 elements")
        wa0, wa1, wb0, wb1, wc0 = [float(w) for w in weights]
        total = torch.empty((), dtype=torch.float32, device=ref.device)
        scaled = torch.empty((), dtype=torch.float32, device=ref.device)
        lib().octa_loss_combine_fwd(_p(a), _p(b), _p(c), wa0, wa1, wb0, wb1, wc0, _p(scale_dev), float(scale_host), _p(total), _p(scaled), _st())
        ctx.cfg = (wa0, wa1, wb0, wb1, wc0, float(scale_host), a is not None, b is not None, c is not None, None if c is None else c.shape)
        ctx.scale_dev = scale_dev
        ctx.mark_non_differentiable(total)
        return total, scaled

    @staticmethod
    @once_differentiable
    def backward(ctx, _g_total, g):
        wa0, wa1, wb0, wb1, wc0, scale_host, ha, hb, hc, cshape = ctx.cfg
        g = g.float().contiguous()
        buf = torch.empty((5,), dtype=torch.float32, device=g.device)
        da, db, dc = (buf[0:2] if ha else None), (buf[2:4] if hb else None), (buf[4:5] if hc else None)
        lib().octa_loss_combine_bwd(_p(g), wa0, wa1, wb0, wb1, wc0, _p(ctx.scale_dev), scale_host, _p(da), _p(db), _p(dc), _st())
        return None, None, None, da, db, (dc.reshape(cshape) if hc else None)


def loss_combine(a, b, c, weights, scale_dev=None, scale_host=1.0):
    """-> (total, scaled); see LossCombineFn."""
    return LossCombineFn.apply(tuple(weights), scale_dev, scale_host, a, b, c)


_ONES = {}


def one_like_seed(t: Tensor) -> Tensor:
    """A persistent fp32 scalar 1 on t's device: `loss.backward(gradient=one_like_seed(loss))` spares the fill launch of the implicit seed."""
    k = (t.device, t.dtype)
    if k not in _ONES:
        _ONES[k] = torch.ones((), dtype=t.dtype, device=t.device)
    return _ONES[k]


def lsgan_discriminator(real, fake):
    return LsganFn.apply(1, fake, real)


# ----------------------------------------------------------------------------- discriminator pieces
class NoiseClipFn(Function):
    """InstanceNoise (+clip) on an NCHW fp32 map, emitting the NHWC activation (blocks.py:149-154)."""

    @staticmethod
    def forward(ctx, y, noise, dtype, clip=True):
        _require_gpu(y)
        y = y.float()
        B, C, H, W = y.shape
        out = nhwc_empty(B, C, H, W, dtype, y.device, pad_written=True)       # the kernel writes all round8(C) channels
        mask = torch.empty((B, C, H, W), dtype=torch.uint8, device=y.device)
        lib().octa_noise_clip_fwd(_p(y), _strides4(y), _p(noise), _p(out), _p(mask), B, C, H, W, round8(C), round8(C), _dt(dtype), int(bool(clip)), _st())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        (mask,) = ctx.saved_tensors
        B, C, H, W = mask.shape
        d = to_nhwc(d)
        g = torch.empty((B, C, H, W), dtype=torch.float32, device=d.device)
        lib().octa_noise_clip_bwd(_p(d), nhwc_ld(d), _p(mask), _p(g), B, C, H, W, _dt(d), _st())
        return g, None, None, None


class NoiseClipS2dFn(Function):
    """InstanceNoise (+clip) written SPACE-TO-DEPTH for the k4 s2 p1 conv behind it (blocks.py:42-46): (B, C, H, W) fp32 ->
    NHWC (B, 4C, H/2+1, W/2+1) with channel (dy*2+dx)*C + c of pixel (Y, X) = the value at (2Y+dy-1, 2X+dx-1), zero outside the
    image.  That conv is then a k2 s1 p0 conv (s2d_weight): same sums, no channel padding (2 -> 8 channels in the plain layout:
    4x the bytes and 4x the MFMA K of the discriminator's first layer)."""

    @staticmethod
    def forward(ctx, y, noise, dtype, clip=True):
        _require_gpu(y)
        y = y.float()
        B, C, H, W = y.shape
        out = nhwc_empty(B, 4 * C, H // 2 + 1, W // 2 + 1, dtype, y.device, pad_written=True)
        mask = torch.empty((B, C, H, W), dtype=torch.uint8, device=y.device)
        lib().octa_noise_clip_s2d_fwd(_p(y), _strides4(y), _p(noise), _p(out), _p(mask), B, C, H, W, nhwc_ld(out), _dt(dtype), int(bool(clip)), _st())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        (mask,) = ctx.saved_tensors
        B, C, H, W = mask.shape
        d = to_nhwc(d)
        g = torch.empty((B, C, H, W), dtype=torch.float32, device=d.device)
        lib().octa_noise_clip_s2d_bwd(_p(d), nhwc_ld(d), _p(mask), _p(g), B, C, H, W, _dt(d), _st())
        return g, None, None, None


def s2d_weight(w: Tensor) -> Tensor:
    """(O, C, 4, 4) weight of a k4 s2 p1 conv -> (O, 4C, 2, 2) weight of the equivalent k2 s1 p0 conv on NoiseClipS2dFn's output:
    w2[o, (dy*2+dx)*C + c, i, j] = w[o, c, 2i+dy, 2j+dx].  Plain tensor ops: autograd carries the gradient back to w."""
    O, C, KH, KW = w.shape
    if (KH, KW) != (4, 4):
        raise OctaError("s2d_weight: 4 x 4 kernels only")
    return w.reshape(O, C, 2, 2, 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(O, 4 * C, 2, 2)


class ToNhwcFn(Function):
    """NCHW fp32 map -> NHWC activation of `dtype` (buffer padded to a multiple of 8 channels)."""

    @staticmethod
    def forward(ctx, y, dtype):
        out = to_nhwc(y, dtype=dtype)
        if out is y:
            out = y.clone()
        ctx.shape = tuple(y.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        return to_nchw_f32(to_nhwc(d)), None


class DiscCatFn(Function):
    """torch.cat((s, y), dim=1) for the discriminator (blocks.py:124): s is the NHWC squeeze output,
    y a user-facing NCHW fp32 map; the result lives in a buffer padded to a multiple of 8."""

    @staticmethod
    def forward(ctx, s, y, inplace=False):
        s = to_nhwc(s)
        B, Cs, H, W = s.shape
        Cy = y.shape[1]
        C = Cs + Cy
        ld = round8(C)
        L = lib()
        lds = nhwc_ld(s)
        yf = y.float()
        if inplace and lds == ld and s.storage_offset() % 8 == 0 and s.stride() == (H * W * ld, 1, W * ld, ld):
            # s (the squeeze conv's own output, read again only by that conv's activation backward, channels < Cs) lives in a buffer
            # of the padded width: the map goes into its pad channels Cs .. Cs+Cy-1 (the rest re-zeroed) and the result is a wider
            # view of the same storage.  No copy of s.
            out = torch.as_strided(s, (B, C, H, W), (H * W * ld, 1, W * ld, ld))
        else:
            out = nhwc_empty(B, C, H, W, s.dtype, s.device, pad_written=True)
            if lds == ld and s.storage_offset() % 8 == 0:
                # same padded width (pad channels zero): ONE 16-byte-chunk copy of all ld channels, then the map overwrites
                # channels Cs .. Cs+Cy-1 and zero-fills the rest
                L.octa_copy_channels(_p(s), H, W, lds, 0, _p(out), H, W, ld, 0, B, ld, _dt(s), 0, _st())
            else:
                s32 = to_nchw_f32(s)
                L.octa_nchw_to_nhwc(_p(s32), s32.stride(0), s32.stride(1), s32.stride(2), s32.stride(3), _p(out), B, Cs, H, W, ld, 0, Cs, _dt(s), _st())
        L.octa_nchw_to_nhwc(_p(yf), yf.stride(0), yf.stride(1), yf.stride(2), yf.stride(3), _p(out), B, Cy, H, W, ld, Cs, ld - Cs, _dt(s), _st())
        ctx.cfg = (Cs, Cy)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        Cs, Cy = ctx.cfg
        d = to_nhwc(d)
        B, C, H, W = d.shape
        ld = nhwc_ld(d)
        ds = d[:, :Cs]
        dy = None
        if ctx.needs_input_grad[1]:            # the generator pass; the discriminator's own step feeds detached maps
            dy = torch.empty((B, Cy, H, W), dtype=torch.float32, device=d.device)
            lib().octa_nhwc_to_nchw(_p(d), ld, Cs, _dt(d), _p(dy), B, Cy, H, W, 0, _st())
        return ds, dy, None


class SpectralNormFn(Function):
    """weight = weight_orig / sigma with one power iteration in training (blocks.py:105-108)."""

    @staticmethod
    def forward(ctx, w, u, v, training, eps):
        _require_gpu(w)
        wd = w.detach().contiguous()
        Cout = wd.shape[0]
        K = wd.numel() // Cout
        sigma = torch.empty((1,), dtype=torch.float32, device=w.device)
        wsn = torch.empty_like(wd)
        ws, pz = _zeroed_f32((K + Cout,), w.device)
        uv = torch.empty((Cout + K,), dtype=torch.float32, device=w.device)      # the u, v THIS forward used (later forwards move them on)
        lib().octa_spectral_norm_fwd(_p(wd), _p(u), _p(v), Cout, K, int(training), eps, _p(sigma), _p(wsn), _p(ws), _p(uv), pz, _st())
        ctx.save_for_backward(wsn, uv, sigma)
        ctx.w_ref = w
        return wsn

    @staticmethod
    @once_differentiable
    def backward(ctx, dwsn):
        wsn, uv, sigma = ctx.saved_tensors
        Cout = wsn.shape[0]
        K = wsn.numel() // Cout
        dwsn = dwsn.float()
        khw = 0
        if not dwsn.is_contiguous():
            if dwsn.dim() == 4 and dwsn.is_contiguous(memory_format=torch.channels_last):
                khw = dwsn.shape[2] * dwsn.shape[3]
            else:
                dwsn = dwsn.contiguous()
        ws, pz = _zeroed_f32((1,), wsn.device)
        sw = _sink(ctx.w_ref)
        if sw is not None and sw.is_contiguous():
            # gradient sink: += straight into weight_orig's pre-assigned gradient (no temporary, no accumulation kernel of autograd's)
            lib().octa_spectral_norm_bwd(_p(dwsn), _p(wsn), _p(uv), _p(uv[Cout:]), _p(sigma), Cout, K, _p(sw), _p(ws), 1, pz, khw, _st())
            return None, None, None, None, None
        dw = torch.empty_like(wsn)
        lib().octa_spectral_norm_bwd(_p(dwsn), _p(wsn), _p(uv), _p(uv[Cout:]), _p(sigma), Cout, K, _p(dw), _p(ws), 0, pz, khw, _st())
        return dw, None, None, None, None


class SpectralNormBatchFn(Function):
    """SpectralNormFn for several independent layers at once (the four spectral-norm convs of a discriminator call): three
    launches forward and two backward for all of them instead of three / two per layer.  apply(training, eps, w0, u0, v0, w1, u1,
    v1, ...) -> (w0 / sigma0, w1 / sigma1, ...)."""

    @staticmethod
    def forward(ctx, training, eps, pack_dtype, *wuv):
        from ._lib import SnJob
        n = len(wuv) // 3
        if n < 1 or n > 8 or len(wuv) != 3 * n:
            raise OctaError("spectral_norm_batch: 1..8 (weight, u, v) triples")
        ws_, keep, outs = [], [], []
        jobs = (SnJob * n)()
        dev = wuv[0].device
        sizes = []
        for i in range(n):
            w, u, v = wuv[3 * i:3 * i + 3]
            _require_gpu(w)
            wd = w.detach().contiguous()
            Cout = wd.shape[0]
            K = wd.numel() // Cout
            sizes.append((Cout, K))
        tot = sum(K + Cout for Cout, K in sizes)
        ws_all, pz = _zeroed_f32((tot,), dev)              # one zero-initialised slice for every layer's v accumulator
        off = 0
        for i in range(n):
            w, u, v = wuv[3 * i:3 * i + 3]
            Cout, K = sizes[i]
            wd = w.detach().contiguous()
            sigma = torch.empty((1,), dtype=torch.float32, device=dev)
            wsn = torch.empty_like(wd)
            uv = torch.empty((Cout + K,), dtype=torch.float32, device=dev)
            ws = ws_all[off:off + K + Cout]
            off += K + Cout
            j = jobs[i]
            j.w, j.u, j.v, j.sigma, j.w_sn, j.ws, j.uv_saved, j.Cout, j.K = _p(wd), _p(u), _p(v), _p(sigma), _p(wsn), _p(ws), _p(uv), Cout, K
            if pack_dtype is not None and wd.dim() == 4 and _SN_PACK:
                # the conv that consumes w_sn wants its packed operands (forward; tap-major data gradient for the few-channel strided
                # layers): written by the launch that writes w_sn instead of two pack launches per layer and call
                _, Cin, KH, KW = wd.shape
                cp, op = round8(Cin), round8(Cout)
                pf = torch.empty((Cout * KH * KW * cp,), dtype=pack_dtype, device=dev)
                j.packed_fwd, j.KH, j.KW, j.Cin, j.pack_dtype = _p(pf), KH, KW, Cin, _dt(pack_dtype)
                pre = {("fwd", pack_dtype, 1, cp): pf}
                if Cin <= 16 and _COL2IM_TAPS and training:
                    pd = torch.empty((KH * KW * cp * op,), dtype=pack_dtype, device=dev)
                    j.packed_dgrad_taps = _p(pd)
                    pre[("dgrad_taps", pack_dtype, 1, op)] = pd
                wsn._octa_packed = pre
            keep.append((wd, sigma, uv))
            outs.append(wsn)
        if not pz:
            ws_all.zero_()
        lib().octa_spectral_norm_fwd_batch(jobs, n, int(training), eps, 1, _st())
        ctx.n = n
        ctx.w_refs = [wuv[3 * i] for i in range(n)]
        ctx.sizes = sizes
        ctx.save_for_backward(*[t for (wd, sigma, uv), wsn in zip(keep, outs) for t in (wsn, uv, sigma)])
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dwsns):
        from ._lib import SnBwdJob
        n = ctx.n
        saved = ctx.saved_tensors
        live = [i for i in range(n) if dwsns[i] is not None]
        grads = [None] * n
        if live:
            jobs = (SnBwdJob * len(live))()
            dots, pz = _zeroed_f32((len(live),), saved[0].device)
            if not pz:
                dots.zero_()
            keep = []
            for q, i in enumerate(live):
                wsn, uv, sigma = saved[3 * i:3 * i + 3]
                Cout, K = ctx.sizes[i]
                dwsn = dwsns[i].float()
                khw = 0
                if not dwsn.is_contiguous():
                    if dwsn.dim() == 4 and dwsn.is_contiguous(memory_format=torch.channels_last):
                        khw = dwsn.shape[2] * dwsn.shape[3]
                    else:
                        dwsn = dwsn.contiguous()
                sw = _sink(ctx.w_refs[i])
                if sw is not None and sw.is_contiguous():
                    dw, acc = sw, 1                      # gradient sink: += straight into weight_orig's pre-assigned gradient
                else:
                    dw, acc = torch.empty_like(wsn), 0
                    grads[i] = dw
                j = jobs[q]
                j.dw_sn, j.w_sn, j.u, j.v, j.sigma, j.dw, j.ws = _p(dwsn), _p(wsn), _p(uv), _p(uv[Cout:]), _p(sigma), _p(dw), _p(dots[q:q + 1])
                j.Cout, j.K, j.accumulate, j.dwsn_khw = Cout, K, acc, khw
                keep.append((dwsn, dw))
            lib().octa_spectral_norm_bwd_batch(jobs, len(live), 1, _st())
        out = [None, None, None]
        for i in range(n):
            out += [grads[i], None, None]
        return tuple(out)


_SN_PACK = os.environ.get("OCTA_SN_PACK", "1") != "0"


def spectral_norm_batch(triples, training: bool, eps: float, pack_dtype=None):
    """[(weight_orig, u, v), ...] -> [weight / sigma, ...] with one power iteration each in training (blocks.py:105-108).
    pack_dtype: the compute dtype of the convs that will consume the weights -- their packed operands are then produced by the same
    launch (they travel on the returned tensors as `_octa_packed`, which functional._packed honours)."""
    flat = [t for tr in triples for t in tr]
    return list(SpectralNormBatchFn.apply(training, eps, pack_dtype, *flat))


class FullConvFn(Function):
    """Conv2d whose kernel covers the whole map (blocks.py:68-72) = one dot product per sample."""

    @staticmethod
    def forward(ctx, x, w, bias, sign, sign_dev=None, gate_in=None):
        ctx.gate_in = gate_in            # (ActGate, act) of the activation that produced x (the last tanh): applied in this op's backward
        x = dense_nhwc(x)
        B, C, H, W = x.shape
        if tuple(w.shape) != (1, C, H, W):
            raise OctaError(f"full conv: weight {tuple(w.shape)} does not cover input {tuple(x.shape)}")
        wp = _packed(w, "fwd", torch.float32, 1, C)          # (h, w, c) order; cached per parameter, refreshed with the other operands
        out, pz = _zeroed_f32((B, 1), x.device)
        lib().octa_fullconv_fwd(_p(x), _p(wp), _p(bias), _p(out), B, H * W * C, _dt(x), float(sign), _p(sign_dev), pz, _st())
        ctx.sign = float(sign)
        ctx.has_bias = bias is not None
        ctx.refs = (w, bias)
        ctx.save_for_backward(x, wp, sign_dev)
        return out                                   # (inside a TrainStep this is a zero-slab slice: valid until the phase ends, which is as long as the loss needs it)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, wp, sign_dev = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = dout.float().contiguous()
        dx = nhwc_empty(B, C, H, W, x.dtype, x.device)
        w, bias = ctx.refs
        sw, sb = _sink(w), _sink(bias)
        gact = 0
        if ctx.gate_in is not None and ctx.gate_in[0] is not None and _ACT_GATES:
            gact = int(ctx.gate_in[1])
            ctx.gate_in[0].done = True
        if sw is not None and sw.is_contiguous() and (not ctx.has_bias or sb is not None):
            # gradient sinks: += straight into the (1, C, H, W) parameter gradient and the bias gradient
            lib().octa_fullconv_bwd_gated(_p(x), _p(wp), _p(dout), _p(dx), _p(sw), _p(sb), B, H * W * C, _dt(x), ctx.sign, _p(sign_dev), C, gact, _st())
            return dx, None, None, None, None, None
        dw = torch.zeros((1, C, H, W), dtype=torch.float32, device=x.device)
        db = torch.zeros((1,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        lib().octa_fullconv_bwd_gated(_p(x), _p(wp), _p(dout), _p(dx), _p(dw), _p(db), B, H * W * C, _dt(x), ctx.sign, _p(sign_dev), C, gact, _st())
        return dx, dw, db, None, None, None


# ----------------------------------------------------------------------------- the rest of the public surface
class InterlayerJSDFn(Function):
    """InterlayerDivergence, JSD branch (segmentor/losses.py:154-169) with the nearest up-sampling fused."""

    @staticmethod
    def forward(ctx, weights, stop_gradient, eps, basis, *maps):
        _require_gpu(basis)
        basis = basis.float().contiguous()
        B, K, H, W = basis.shape
        use = [(m.float().contiguous(), float(w)) for m, w in zip(maps, weights) if w != 0]
        shifts = []
        for m, _ in use:
            f = H // m.shape[2]
            if f < 1 or f & (f - 1) or m.shape[2] * f != H or m.shape[3] * f != W:
                raise OctaError(f"interlayer JSD: map {tuple(m.shape)} is not a power-of-two reduction of {tuple(basis.shape)}")
            shifts.append(f.bit_length() - 1)
        n = len(use)
        ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m, _ in use])
        sh = (ctypes.c_int * n)(*shifts)
        wt = (ctypes.c_float * n)(*[w for _, w in use])
        out = torch.empty((2,), dtype=torch.float32, device=basis.device)
        ws = torch.empty((1024,), dtype=torch.float32, device=basis.device)
        lib().octa_interlayer_jsd_fwd(_p(basis), ptrs, sh, wt, n, float(eps), B, K, H, W, _p(out), _p(ws), _st())
        ctx.cfg = (shifts, [w for _, w in use], float(eps), stop_gradient, [i for i, w in enumerate(weights[:len(maps)]) if w != 0], len(maps))
        ctx.save_for_backward(basis, *[m for m, _ in use])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        basis, *use = ctx.saved_tensors
        shifts, wts, eps, stop_gradient, idx, nmaps = ctx.cfg
        B, K, H, W = basis.shape
        g0 = g.float().contiguous()
        n = len(use)
        dmaps = [torch.empty_like(m) for m in use]
        ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in use])
        dptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in dmaps])
        sh = (ctypes.c_int * n)(*shifts)
        wt = (ctypes.c_float * n)(*wts)
        dbasis = None if stop_gradient else torch.empty_like(basis)
        gq = torch.empty_like(basis)
        lib().octa_interlayer_jsd_bwd(_p(basis), ptrs, sh, wt, n, eps, B, K, H, W, _p(g0), _p(dbasis), _p(gq), dptrs, _st())
        grads: List[Optional[Tensor]] = [None] * nmaps
        for i, d in zip(idx, dmaps):
            grads[i] = d
        return (None, None, None, dbasis, *grads)


def interlayer_jsd(attentions: Sequence[Tensor], weights, stop_gradient=False, eps=1e-12) -> Tensor:
    """returns a (2,) tensor: [loss, nan_flag]."""
    return InterlayerJSDFn.apply(list(weights), stop_gradient, eps, attentions[0], *attentions[1:])


class PixelCEFn(Function):
    """WeightedPartialCE's nn.CrossEntropyLoss (mode 0, 2 classes) / nn.BCEWithLogitsLoss (mode 1, 1 class) branches
    (segmentor/losses.py:40-60) on the masked scores y_hat * ys."""

    @staticmethod
    def forward(ctx, y_hat, ys, full, mode):
        _require_gpu(y_hat)
        y_hat, ys = y_hat.float(), ys.float()
        B, K, H, W = y_hat.shape
        out = torch.empty((2,), dtype=torch.float32, device=y_hat.device)
        ws = torch.empty((1024,), dtype=torch.float32, device=y_hat.device)
        lib().octa_pixel_ce_fwd(_p(y_hat), _strides4(y_hat), _p(ys), _strides4(ys), B, K, H, W, int(full), mode, _p(out), _p(ws), _st())
        ctx.cfg = (int(full), mode)
        ctx.save_for_backward(y_hat, ys)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y_hat, ys = ctx.saved_tensors
        B, K, H, W = y_hat.shape
        g = g.float().contiguous().view(1)
        din = torch.empty((B, K, H, W), dtype=torch.float32, device=y_hat.device)
        lib().octa_pixel_ce_bwd(_p(y_hat), _strides4(y_hat), _p(ys), _strides4(ys), B, K, H, W, ctx.cfg[0], ctx.cfg[1], _p(g), _p(din), _st())
        return din, None, None, None


def pixel_ce(y_hat, ys, full=False, mode=0):
    return PixelCEFn.apply(y_hat, ys, full, mode)


class Abs1mFn(Function):
    """|1 - x| (LabelNoise mode 'label', discriminator/blocks.py:172-177)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xf = x.float().contiguous()
        out = torch.empty_like(xf)
        lib().octa_abs1m(_p(xf), None, _p(out), xf.numel(), _st())
        ctx.save_for_backward(xf)
        ctx.dtype = x.dtype
        return out.to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (xf,) = ctx.saved_tensors
        gf = g.float().contiguous()
        out = torch.empty_like(xf)
        lib().octa_abs1m(_p(xf), _p(gf), _p(out), xf.numel(), _st())
        return out.to(ctx.dtype)


def predict_sigmoid(logits: Tensor) -> Tensor:
    """nn.Sigmoid()(agg_map) (segmentor/compose.py:193) -> dense fp32 NCHW."""
    _require_gpu(logits)
    logits = logits.float()
    B, K, H, W = logits.shape
    out = torch.empty((B, K, H, W), dtype=torch.float32, device=logits.device)
    lib().octa_predict_map(_p(logits), _strides4(logits), B, K, H, W, 0, _p(out), None, _st())
    return out


def predict_one_hot(logits: Tensor) -> Tensor:
    """rearrange(F.one_hot(torch.argmax(agg_map, dim=1)), 'b h w c -> b c h w') (segmentor/compose.py:195): int64, first maximum
    wins, and -- like F.one_hot without num_classes -- the channel count is 1 + the largest class that occurs."""
    _require_gpu(logits)
    logits = logits.float()
    B, K, H, W = logits.shape
    out = torch.empty((B, K, H, W), dtype=torch.int64, device=logits.device)
    mx = torch.zeros((1,), dtype=torch.int32, device=logits.device)
    lib().octa_predict_map(_p(logits), _strides4(logits), B, K, H, W, 1, _p(out), _p(mx), _st())
    return out[:, :int(mx.item()) + 1]      # F.one_hot itself synchronises on the maximum


def dice_coefficient(pred: Tensor, target: Tensor, eps: float = 1e-12) -> Tensor:
    """Dice coefficient 2 |A.B| / (|A| + |B|) per (sample, class) -> (B, K) fp32 (the metric the paper reports; one launch)."""
    _require_gpu(pred)
    pred, target = pred.float(), target.float()
    B, K, H, W = pred.shape
    terms = torch.empty((B, K, 2), dtype=torch.float32, device=pred.device)
    lib().octa_dice_terms(_p(pred), _strides4(pred), _p(target), _strides4(target), B, K, H, W, _p(terms), _st())
    return 2.0 * terms[..., 0] / (terms[..., 1] + eps)


class AdaptiveAvgPoolFn(Function):
    """nn.AdaptiveAvgPool2d on a dense NCHW fp32 map (classification head, segmentor/compose.py:89)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        _require_gpu(x)
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        lib().octa_adaptive_avgpool(_p(x), None, _p(y), B * C, H, W, OH, OW, _st())
        ctx.cfg = (B, C, H, W, OH, OW)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W, OH, OW = ctx.cfg
        dy = dy.float().contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        lib().octa_adaptive_avgpool(None, _p(dy), _p(dx), B * C, H, W, OH, OW, _st())
        return dx, None, None


def adaptive_avg_pool(x, output_size):
    OH, OW = (output_size, output_size) if isinstance(output_size, int) else output_size
    return AdaptiveAvgPoolFn.apply(x, OH, OW)


def synth_octa_batch(B: int, H: int, W: int, seed: int, device, vessel: bool = False):
    """Device-side synthetic batch: x (B,3,H,W), scribbles ys (B,2,H,W), dense real mask (B,2,H,W), fp32 NCHW."""
    x = torch.empty((B, 3, H, W), dtype=torch.float32, device=device)
    ys = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
    real = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
    _require_gpu(x)
    lib().octa_synth_octa(int(seed), B, H, W, int(vessel), _p(x), _p(ys), _p(real), _st())
    return x, ys, real


def mask_pyramid_dense(mask: Tensor, levels: int = 5) -> List[Tensor]:
    """The discriminator's real pyramid (nearest down-sampling by 2**i; contract of discriminator/blocks.py:114-125) as DENSE
    tensors written by one launch (the strided-view form costs a gather per level and per use)."""
    _require_gpu(mask)
    mask = mask.float().contiguous()
    B, C, H, W = mask.shape
    outs = [mask] + [torch.empty((B, C, H >> l, W >> l), dtype=torch.float32, device=mask.device) for l in range(1, levels)]
    ptrs = (ctypes.c_void_p * levels)(*[o.data_ptr() for o in outs])
    lib().octa_mask_pyramid(_p(mask), ptrs, levels, B * C, H, W, _st())
    return outs
