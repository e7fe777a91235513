"""nn.Module leaves whose forward runs on libocta_hip.so.  They subclass the torch modules only to
inherit parameter registration, initialisation and state_dict layout (identical keys/shapes to the
reference); no ATen compute kernel is called in their forward."""
import torch
from torch import nn

from . import functional as F_
from ._lib import ACT_NONE


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

    def forward(self, x):
        return F_.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, self.act)


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

    def forward(self, x, relu: bool = False, residual=None):
        training = self.training or (self.running_mean is None)
        mom = 0.0 if self.momentum is None else self.momentum
        if training and self.track_running_stats and self.num_batches_tracked is not None:
            if self.momentum is None:
                self.num_batches_tracked.add_(1)
                mom = 1.0 / float(self.num_batches_tracked)
            else:
                bump_counter(self.num_batches_tracked)
        return F_.batch_norm(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                             self.running_var if self.track_running_stats else None, mom, self.eps, training, relu, residual)


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
