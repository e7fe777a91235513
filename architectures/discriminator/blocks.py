"""Mirror of reference architectures/discriminator/blocks.py: ``DiscriminatorBlock`` (ref :12-133),
``InstanceNoise`` (:135-154), ``LabelNoise`` (:156-185) -- the multi-scale LS-GAN critic.

Parity contract kept from the reference: every training-mode call draws, in this
order, one ``torch.normal(size=(H, W))`` plane and one ``FloatTensor(1).uniform_`` from the GLOBAL
CPU generator, and advances the four spectral-norm (u, v) pairs by one power iteration.
"""
import os
from typing import List, Literal, Sequence

import torch
from torch import Size, Tensor, nn
from torch.nn.init import kaiming_normal_, xavier_uniform_

from architectures.utils import rand_uniform
from octave_amd import functional as F_
from octave_amd._lib import ACT_LEAKY02, ACT_SIGMOID, ACT_TANH
from octave_amd.layers import Conv2d


_SN_BATCH = os.environ.get("OCTA_SN_BATCH", "1") != "0"      # one batched power iteration per discriminator call
_S2D_CONV0 = os.environ.get("OCTA_S2D_CONV0", "1") != "0"    # first conv on the space-to-depth image of its input (round 5)


class SpectralConv2d(Conv2d):
    """Conv2d under torch.nn.utils.spectral_norm(n_power_iterations=1): parameters ``weight_orig``,
    buffers ``weight_u`` / ``weight_v`` (same state_dict keys as the reference's hook form)."""

    def __init__(self, *args, eps: float = 1e-12, **kwargs):
        super().__init__(*args, **kwargs)
        w = self.weight
        del self._parameters['weight']
        self.register_parameter('weight_orig', nn.Parameter(w.data))
        h, wdt = w.shape[0], w[0].numel()
        with torch.no_grad():   # same draws as SpectralNorm.apply: u ~ N(0,1)^h, v ~ N(0,1)^w, normalised
            u = nn.functional.normalize(w.new_empty(h).normal_(0, 1), dim=0, eps=eps)
            v = nn.functional.normalize(w.new_empty(wdt).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)
        self.sn_eps = eps

    def forward(self, x, w=None, gate_out=None, gate_in=None):
        """`w`: the normalised weight when the caller ran this layer's power iteration together with other layers'
        (functional.spectral_norm_batch); None: run it here.  gate_out / gate_in: functional.ActGate."""
        if w is None:
            w = F_.SpectralNormFn.apply(self.weight_orig, self.weight_u, self.weight_v, self.training, self.sn_eps)
        return F_.conv2d(x, w, self.bias, self.stride[0], self.padding[0], self.groups, self.act, None, None, gate_out, gate_in)


class InstanceNoise(nn.Module):

    def __init__(self, input_shape: Size, mean: float, std: float, clipping: bool, is_training: bool):
        """Gaussian noise addition: ONE (H, W) plane per call, broadcast over batch and channels."""
        super().__init__()
        self.mean = mean
        self.std = std
        self.clipping = clipping
        self.size = (input_shape[2], input_shape[3])
        self.is_training = is_training
        self.compute_dtype = None

    def draw(self, out: Tensor = None) -> Tensor:
        """One (H, W) plane from the global CPU generator, like ref :150.  `out` (e.g. a slot of pinned staging memory)
        receives the draw directly: copying a 160k-element CPU tensor afterwards runs as an OpenMP parallel region, and the
        idle OpenMP workers then spin for ~100-200 ms on every core the process may use - which starved the HIP runtime's
        helper threads and stalled queued hipGraph replays (train.py, DESIGN.md 'launch path')."""
        if out is not None:
            return torch.normal(mean=self.mean, std=self.std, size=self.size, out=out)
        return torch.normal(mean=self.mean, std=self.std, size=self.size)

    def forward(self, x: Tensor, noise_dev: Tensor = None, fed: bool = False, s2d: bool = False):
        """`fed=True`: the (H, W) plane was drawn by the caller (same CPU-generator order) and already lives on
        the device (hipGraph replay); otherwise it is drawn here exactly like the reference.  `s2d`: emit the space-to-depth
        layout the k4 s2 p1 conv behind this module reads as a k2 s1 conv (functional.NoiseClipS2dFn)."""
        if fed:
            nz = noise_dev if self.is_training else None
        else:
            noise = self.draw()                  # drawn even when not added (ref :150-151)
            nz = noise.to(x.device, non_blocking=True) if self.is_training else None
        dtype = self.compute_dtype or torch.float32
        if s2d:
            return F_.NoiseClipS2dFn.apply(x, nz, dtype, self.clipping)
        return F_.NoiseClipFn.apply(x, nz, dtype, self.clipping)


class LabelNoise(nn.Module):

    def __init__(self, prob: float = 0.1, mode: Literal['sign', 'label'] = 'sign'):
        """Label noise: with probability `prob` the whole logits tensor changes sign."""
        super().__init__()
        self.prob = prob
        self.mode = mode

    def draw_sign(self) -> float:
        return -1.0 if bool(rand_uniform() < self.prob) else 1.0

    def forward(self, x: Tensor):
        if self.mode == 'sign':                      # ref :165-170
            return x * self.draw_sign()
        if self.mode == 'label':                     # ref :172-177: |1 - x| with probability prob (one CPU uniform per call)
            return F_.Abs1mFn.apply(x) if bool(rand_uniform() < self.prob) else x
        raise NotImplementedError


class DiscriminatorBlock(nn.Module):

    def __init__(self, input_shape: Size, is_training: bool, depth: int = 3, num_filters: int = 64, instance_noise: bool = True,
                 label_noise: bool = True):
        """Multi-scale discriminator (design: Valvano et al.).  forward(y): y[0] is the full-resolution
        class map, y[i] the map at 1/2**i; returns (B, 1) logits."""
        super().__init__()
        self.num_filters = num_filters
        self.is_training = is_training
        self.depth = depth
        self.compute_dtype = None
        self.rng_feed = None       # optional: object with next_noise()/next_sign() returning DEVICE tensors (octave_amd.train)
        self.input_hw = (int(input_shape[2]), int(input_shape[3]))
        in_channels = input_shape[1]
        modules = []
        if instance_noise:
            modules.append(InstanceNoise(input_shape=input_shape, is_training=is_training, mean=.0, std=.2, clipping=True))
        conv_0 = Conv2d(in_channels, num_filters, kernel_size=4, stride=2, padding=1, act=ACT_LEAKY02)
        kaiming_normal_(conv_0.weight, nonlinearity='leaky_relu')
        modules.append(conv_0)
        modules.append(nn.LeakyReLU(negative_slope=0.2))        # fused into conv_0's epilogue; kept for indices
        self.stack_0 = nn.Sequential(*modules)
        self._has_noise = instance_noise

        squeeze_stack, spectral_stack = dict(), dict()
        for i in range(self.depth):
            squeeze, spectral = self._discriminator(
                in_channels=num_filters * (2 ** i), num_squeeze_filters=13, num_fake_channels=in_channels,
                num_sn_filters=num_filters * 2 * (2 ** i), sn_kernel_size=4, num_sn_stride=2, sn_padding=1)
            squeeze_stack[f'squeeze_{i}'] = squeeze
            spectral_stack[f'spectral_{i}'] = spectral
        self.squeeze_dict = nn.ModuleDict(squeeze_stack)
        self.spectral_dict = nn.ModuleDict(spectral_stack)
        h, w = [int(i) // (2 ** (self.depth + 1)) for i in input_shape[2:]]
        fc = nn.Conv2d(num_filters * (2 ** self.depth), out_channels=1, kernel_size=(h, w), stride=1)   # parameters only
        xavier_uniform_(fc.weight)
        modules = [fc, nn.Flatten()]
        if label_noise:
            modules.append(LabelNoise(0.1, 'sign'))
        self.out = nn.Sequential(*modules)
        self._has_label_noise = label_noise

    def share_body_with(self, other: "DiscriminatorBlock") -> "DiscriminatorBlock":
        """Mixed-resolution training (BASELINE config 5): the reference's block is tied to ONE resolution -- its full-extent
        head conv (ref :68-72) and the InstanceNoise plane (ref :42) are sized by `input_shape` -- so a second resolution
        needs a second block.  This makes `self` use `other`'s conv stack (first conv, squeeze and spectral-norm convs incl.
        their u / v buffers) so that only the head `out.0` is resolution-specific.  Returns self."""
        if (self.depth, self.num_filters, self._has_noise) != (other.depth, other.num_filters, other._has_noise):
            raise ValueError("share_body_with: the two discriminators must have the same depth / width / noise configuration")
        i = 1 if self._has_noise else 0
        self.stack_0[i] = other.stack_0[i]
        self.squeeze_dict = other.squeeze_dict
        self.spectral_dict = other.spectral_dict
        return self

    def _discriminator(self, in_channels, num_squeeze_filters: int, num_fake_channels: int, num_sn_filters: int, sn_kernel_size: int,
                       num_sn_stride: int, sn_padding: int):
        squeezed = nn.Sequential(Conv2d(in_channels, num_squeeze_filters, kernel_size=1, stride=1, act=ACT_SIGMOID), nn.Sigmoid())
        spectral = nn.Sequential(
            SpectralConv2d(num_squeeze_filters + num_fake_channels, num_sn_filters, kernel_size=sn_kernel_size, stride=num_sn_stride,
                           padding=sn_padding, act=ACT_TANH),
            nn.Tanh())
        return squeezed, spectral

    def forward(self, y: Sequence[Tensor]):
        dtype = self.compute_dtype or torch.float32
        feed = self.rng_feed
        # every activation of the chain has exactly one consumer: that consumer's data gradient applies the activation's derivative
        # in its own epilogue (functional.ActGate) and the 9 derivative launches of a backward pass disappear
        gated = F_.act_gates_enabled() and torch.is_grad_enabled()
        g_prev = F_.ActGate() if gated else None
        a_prev = ACT_LEAKY02
        if self._has_noise:
            self.stack_0[0].compute_dtype = dtype
            c0 = self.stack_0[1]
            # the first conv (k4 s2 p1 on 2 channels) as a k2 s1 conv on the space-to-depth image of its input: 8 real channels
            # per pixel instead of 2 padded to 8 (a quarter of the input bytes and of the MFMA K; its data gradient is an ordinary
            # stride-1 one instead of a tap GEMM + fold)
            s2d = (_S2D_CONV0 and y[0].is_cuda and c0.kernel_size == (4, 4) and c0.stride == (2, 2) and c0.padding == (1, 1) and c0.groups == 1
                   and (4 * c0.in_channels) % 8 == 0 and y[0].shape[2] % 2 == 0 and y[0].shape[3] % 2 == 0)
            s = self.stack_0[0](y[0], feed.next_noise(), True, s2d) if feed is not None else self.stack_0[0](y[0], None, False, s2d)
            if s2d:
                s = F_.conv2d(s, F_.s2d_weight(c0.weight), c0.bias, 1, 0, 1, c0.act, None, None, g_prev, None)
            else:
                s = c0(s, gate_out=g_prev)
        else:
            s = self.stack_0[0](F_.ToNhwcFn.apply(y[0], dtype), gate_out=g_prev)
        # the spectral-norm convs' power iterations (independent of each other and of the activations): one batched call
        sn = [self.spectral_dict[f'spectral_{i}'][0] for i in range(self.depth)]
        wn = F_.spectral_norm_batch([(m.weight_orig, m.weight_u, m.weight_v) for m in sn], sn[0].training, sn[0].sn_eps, dtype) \
            if (_SN_BATCH and 1 <= self.depth <= 8 and all(m.training == sn[0].training and m.sn_eps == sn[0].sn_eps for m in sn)) else [None] * self.depth
        for i in range(self.depth):
            try:
                g_sq = F_.ActGate() if gated else None
                sq = self.squeeze_dict[f'squeeze_{i}'][0]
                s = sq(s, gate_out=g_sq, gate_in=(g_prev, a_prev, 0))
                s = F_.DiscCatFn.apply(s, y[i + 1], True)      # the map goes into the squeeze output's pad channels
                g_prev, a_prev = (F_.ActGate() if gated else None), ACT_TANH
                s = sn[i](s, wn[i], gate_out=g_prev, gate_in=(g_sq, ACT_SIGMOID, sq.out_channels))
            except Exception as e:
                raise Exception(f'Exception raised in depth = {i}') from e
        fc = self.out[0]
        g_last = (g_prev, a_prev) if self.depth > 0 else (g_prev, ACT_LEAKY02)
        if self._has_label_noise and feed is not None:
            return F_.FullConvFn.apply(s, fc.weight, fc.bias, 1.0, feed.next_sign(), g_last)
        sign = self.out[2].draw_sign() if self._has_label_noise else 1.0
        return F_.FullConvFn.apply(s, fc.weight, fc.bias, sign, None, g_last)

    def predict(self, y: List[Tensor]):
        return self.forward(y)
