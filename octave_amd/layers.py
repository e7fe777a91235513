"""nn.Module leaves whose forward runs on libocta_hip.so.  They subclass the torch modules only to
inherit parameter registration, initialisation and state_dict layout (identical keys/shapes to the
reference); no ATen compute kernel is called in their forward."""
import os
import weakref

import torch
from torch import nn

from . import functional as F_
from ._lib import ACT_NONE, ACT_RELU


# num_batches_tracked bookkeeping: per-call `add_(1)` is one tiny ATen launch per BatchNorm (93 per step).
# A training loop may defer them: forwards then only enlist their counter and flush_bn_counters() bumps all
# of them with ONE multi-tensor add.  state_dict()/eval semantics are unchanged after the flush.
_DEFER_COUNTERS = False
_PENDING_COUNTERS = []


def defer_bn_counters(on: bool):
    global _DEFER_COUNTERS
    _DEFER_COUNTERS = bool(on)
    if not on:
        flush_bn_counters()


def flush_bn_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


def bump_counter(t):
    if _DEFER_COUNTERS:
        _PENDING_COUNTERS.append(t)
    else:
        t.add_(1)


class Conv2d(nn.Conv2d):
    """nn.Conv2d with forward on the MFMA implicit-GEMM engine."""

    def __init__(self, *args, act: int = ACT_NONE, **kwargs):
        super().__init__(*args, **kwargs)
        if self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("octave_amd.Conv2d: square kernel/stride/padding only")
        if self.dilation != (1, 1) or self.padding_mode != "zeros":
            raise NotImplementedError("octave_amd.Conv2d: dilation 1 and zero padding only")
        self.act = act

    def forward(self, x, grad_holder=None, stats=None, gate_out=None, gate_in=None):
        return F_.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, self.act, grad_holder, stats, gate_out, gate_in)


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(kernel_size=2, stride=2) as an up-shuffle GEMM."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, **kwargs)
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0) or self.groups != 1:
            raise NotImplementedError("octave_amd.ConvTranspose2d: kernel 2, stride 2, no padding only")

    def forward(self, x):
        return F_.conv_transpose2x2(x, self.weight, self.bias)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (batch statistics in training, running-stat update) with optional fused
    ReLU / residual add, selected per call."""

    def forward(self, x, relu: bool = False, residual=None, pre_sums=None):
        training = self.training or (self.running_mean is None)
        mom = 0.0 if self.momentum is None else self.momentum
        if training and self.track_running_stats and self.num_batches_tracked is not None:
            if self.momentum is None:
                self.num_batches_tracked.add_(1)
                mom = 1.0 / float(self.num_batches_tracked)
            else:
                bump_counter(self.num_batches_tracked)
        return F_.batch_norm(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                             self.running_var if self.track_running_stats else None, mom, self.eps, training, relu, residual, pre_sums)


class ReLU(nn.Module):
    """Placeholder that keeps nn.Sequential indices identical to the reference.  The ReLU itself is
    fused into the producing kernel (BatchNorm apply / split-attention apply), so this is identity."""

    def __init__(self, inplace: bool = False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return x


def use_channels_last_weights(module: nn.Module) -> nn.Module:
    """Re-lay every 4-D conv weight of `module` in channels-last MEMORY (logical OIHW shape, values
    and state_dict unchanged).  The weight-gradient kernel then accumulates into contiguous runs,
    and the fp32 forward operand is the parameter storage itself.  Called after initialisation so
    that seeded initialisers consume the generator exactly like the reference."""
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, (Conv2d, ConvTranspose2d)):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


# ----------------------------------------------------------------------------- inference: BatchNorm folded into the conv
# Eval-mode conv -> BatchNorm (-> ReLU) is ONE conv launch: the running statistics are folded into the weights
# (w' = w * gamma / sqrt(var + eps) per output channel) and into a bias (b' = (b - mean) * gamma / sqrt(var + eps) + beta)
# when the operand is packed; the folded parameters are rebuilt only when a weight, an affine parameter or a running
# statistic changes (version counters).  Training mode and grad-enabled eval go through the unfused layers.
# BatchNorm statistics in the producing conv's epilogue: built, parity-tested, measured at break-even in the step (DESIGN.md 9.3:
# the epilogue costs the 25-40 us pointwise kernels what the separate statistics pass cost) -> off unless OCTA_FUSE_BN_STATS=1
_FUSE_BN_STATS = os.environ.get("OCTA_FUSE_BN_STATS", "0") == "1"
_FUSE_BN_MIN_BYTES = int(float(os.environ.get("OCTA_FUSE_BN_STATS_MIN_MB", "16")) * (1 << 20))   # output tensors below this keep the two-launch small-tensor BatchNorm
_FOLD_CACHE = {}
_IDENT = {}


def _fold_entry(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    key = (id(conv.weight), id(bn.running_mean))
    w, cb = conv.weight, conv.bias
    # _PARAM_EPOCH: the fused Adam and the BatchNorm training forward write weights / running statistics through raw pointers
    tag = (w._version, F_._WEIGHT_EPOCH, F_._PARAM_EPOCH, w.data_ptr(), bn.running_mean._version, bn.running_var._version, bn.weight._version, bn.bias._version,
           None if cb is None else cb._version, bn.eps)
    e = _FOLD_CACHE.get(key)
    if e is not None and e[0] == tag and e[3]() is w and e[4]() is bn:
        return e[1], e[2]
    with torch.no_grad():
        scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        wf = (w.detach().float() * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
        base = -bn.running_mean.float() if cb is None else (cb.detach().float() - bn.running_mean.float())
        bf = (base * scale + bn.bias.float()).contiguous()
    # a Parameter object so that the pack cache of functional.py keys on its identity (packed once per fold)
    wp = nn.Parameter(wf, requires_grad=False)
    _FOLD_CACHE[key] = (tag, wp, bf, weakref.ref(w), weakref.ref(bn))
    return wp, bf


def _identity_stats(C: int, device):
    k = (C, str(device))
    t = _IDENT.get(k)
    if t is None:
        t = (torch.zeros(C, device=device), torch.ones(C, device=device))
        _IDENT[k] = t
    return t


def _conv_out_bytes(conv, x) -> int:
    k, s_, p_ = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    oh, ow = (x.shape[2] + 2 * p_ - k) // s_ + 1, (x.shape[3] + 2 * p_ - k) // s_ + 1
    return x.shape[0] * conv.out_channels * oh * ow * x.element_size()


def conv_bn(conv: "Conv2d", bn: "BatchNorm2d", x, relu: bool = False, residual=None, grad_holder=None):
    """bn(conv(x)) [+ residual] [-> relu].  Training: the two fused-statistics layers.  Inference (eval mode, no grad): one conv
    launch with the BatchNorm folded into its packed weights and bias (+ one add/ReLU pass when there is a residual).
    `grad_holder` (functional.GradHolder): the conv's data gradient adds the gradient parked there by functional.stash_grad."""
    if bn.training or bn.running_mean is None or torch.is_grad_enabled():
        if (_FUSE_BN_STATS and bn.training and bn.track_running_stats and bn.running_mean is not None and conv.act == ACT_NONE
                and x.dtype in (torch.bfloat16, torch.float16) and x.is_cuda and _conv_out_bytes(conv, x) >= _FUSE_BN_MIN_BYTES):
            # 16-bit training: the conv sums its own output for the BatchNorm (per-channel sum / sum of squares around the
            # running mean, in its epilogue); BatchNorm is then a replica merge + the apply launch
            st = F_.ConvStats(conv.out_channels, bn.running_mean, x.device)
            y = conv(x, grad_holder, st)
            return bn(y, relu=relu, residual=residual, pre_sums=st)
        return bn(conv(x, grad_holder) if grad_holder is not None else conv(x), relu=relu, residual=residual)
    wf, bf = _fold_entry(conv, bn)
    act = ACT_RELU if (relu and residual is None) else ACT_NONE
    if conv.act != ACT_NONE:
        raise NotImplementedError("conv_bn: the conv already has a fused activation")
    y = F_.conv2d(x, wf, bf, conv.stride[0], conv.padding[0], conv.groups, act)
    if residual is None:
        return y
    zero, one = _identity_stats(y.shape[1], y.device)
    # (y - 0) * 1 * 1 + 0 + residual -> relu: the BatchNorm-apply kernel with identity statistics is the add(+ReLU) pass
    return F_.raw_bn_apply_only(y, zero, one, one, zero, relu, residual)


class Linear(nn.Linear):
    """nn.Linear as a 1x1 conv on the MFMA engine (classification heads, segmentor/compose.py:85,97): same parameters/keys."""

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("octave_amd.Linear: (B, features) input")
        y = F_.conv2d(x.view(x.shape[0], x.shape[1], 1, 1), self.weight.view(self.out_features, self.in_features, 1, 1), self.bias)
        return F_.to_nchw_f32(y).view(x.shape[0], self.out_features)


class AdaptiveAvgPool2d(nn.Module):
    """nn.AdaptiveAvgPool2d on dense fp32 NCHW maps (classification head)."""

    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        if F_.nhwc_ld(x) is not None and x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 1:
            x = F_.to_nchw_f32(x)
        return F_.adaptive_avg_pool(x, self.output_size)
