"""ResNeSt-50 encoder blocks and U-Net decoder blocks on libocta_hip.so.

Mirror of reference architectures/extra/resnest.py for the hot path: ``SplAtConv2d`` (ref :57-138),
``Bottleneck`` (:170-267), ``ResNet`` (:277-449, resnest50 configuration), ``ResNestDecoder``
(:18-43), ``Upsampling`` (:46-54), ``resnest50`` (:451-459).  Same constructor arguments, attribute
names and state_dict keys.  The unreachable options of the reference (DropBlock, rectified conv,
radix 1, dilation) raise NotImplementedError instead of being reproduced.
"""
import math
import os

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from octave_amd import functional as F_
from octave_amd.layers import BatchNorm2d, Conv2d, ConvTranspose2d, ReLU, bump_counter, conv_bn

BN_MOMENTUM = 0.1
_FUSE_FANOUT = os.environ.get("OCTA_FUSE_FANOUT", "1") != "0"     # Bottleneck: shortcut gradient added in conv1's data-gradient epilogue
_FUSE_SPLAT_BN0 = os.environ.get("OCTA_FUSE_SPLAT_BN0", "1") != "0"    # SplAtConv2d (training): bn0 + ReLU recomputed inside the split-attention kernels
# Shortcut branches on a second stream (functional.SideBranch): built and parity-tested in round 4, measured on one box alternating
# (profiles/r04_ab_env.txt): encoder shortcuts 26.05 -> 26.0-26.3 ms per step (no gain: HBM-bound kernels beside HBM-bound kernels),
# decoder shortcut 1x1s 26.05 -> 27.4 (the 1x1 competes with the one-workgroup-per-CU 3x3 kernel for CUs) -- OFF by default.
_SIDE_SHORTCUT = os.environ.get("OCTA_SIDE_SHORTCUT", "0") == "1"          # Bottleneck: avg-down shortcut on a second stream
_SIDE_SHORTCUT_DEC = os.environ.get("OCTA_SIDE_SHORTCUT_DEC", "0") == "1"  # ResNestDecoder: shortcut 1x1 on a second stream
_FUSE_FANOUT_DEC = os.environ.get("OCTA_FUSE_FANOUT_DEC", "1") != "0"     # ResNestDecoder: 3x3 gradient added in the shortcut conv's data gradient


class SplAtConv2d(nn.Module):
    """Split-attention conv, radix 2: grouped 3x3 -> bn0 -> relu -> [radix-sum GAP -> fc1 -> bn1 ->
    relu -> fc2 -> radix softmax -> weighted sum] (the bracket is one fused op, ``F_.splat_tail``).
    As in the reference the radix softmax views fc2's output as (B, radix, C) with no cardinality
    transpose (ref :125)."""

    def __init__(self, in_channels, channels, kernel_size, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, bias=True,
                 radix=2, reduction_factor=4, rectify=False, rectify_avg=False, norm_layer=None, dropblock_prob=0.0, **kwargs):
        super().__init__()
        if radix != 2 or rectify or dropblock_prob > 0.0 or norm_layer is None:
            raise NotImplementedError("SplAtConv2d: only radix=2 with a norm layer, no rectify/dropblock (hot-path configuration)")
        padding = _pair(padding)
        inter_channels = max(in_channels * radix // reduction_factor, 32)
        self.radix = radix
        self.cardinality = groups
        self.channels = channels
        self.dropblock_prob = dropblock_prob
        self.rectify = False
        self.rectify_avg = rectify_avg
        self.conv = Conv2d(in_channels, channels * radix, kernel_size, _pair(stride), padding, _pair(dilation), groups=groups * radix, bias=bias,
                           **kwargs)
        self.use_bn = True
        self.bn0 = BatchNorm2d(channels * radix)
        self.relu = ReLU(inplace=True)
        self.fc1 = Conv2d(channels, inter_channels, 1, groups=self.cardinality)
        self.bn1 = BatchNorm2d(inter_channels)
        self.fc2 = Conv2d(inter_channels, channels * radix, 1, groups=self.cardinality)

    def forward(self, x, relu_after: bool = False):
        bn0, bn1 = self.bn0, self.bn1
        training = bn1.training
        # training: bn0 + ReLU are recomputed inside the split-attention kernels from the raw conv output (functional.SplatTailFn):
        # no BatchNorm-apply pass, no stored activation / gradient of the 2C-channel tensor
        fuse0 = (_FUSE_SPLAT_BN0 and training and bn0.training and bn0.momentum is not None and x.is_cuda and self.conv.act == 0)
        if fuse0:
            x = self.conv(x)
            if bn0.track_running_stats and bn0.num_batches_tracked is not None:
                bump_counter(bn0.num_batches_tracked)
            pre = (bn0.weight, bn0.bias, bn0.running_mean if bn0.track_running_stats else None,
                   bn0.running_var if bn0.track_running_stats else None, bn0.momentum, bn0.eps)
        else:
            x = conv_bn(self.conv, bn0, x, relu=True)
            pre = None
        if training and bn1.num_batches_tracked is not None:
            bump_counter(bn1.num_batches_tracked)
        return F_.splat_tail(x, self.fc1.weight, self.fc1.bias, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                             self.fc2.weight, self.fc2.bias, self.cardinality, bn1.momentum, bn1.eps, training, relu_after, bn0=pre)


class Bottleneck(nn.Module):
    """ResNeSt bottleneck (ref :170-267): 1x1 -> bn -> relu -> SplAt -> [avd 3x3 avgpool] -> 1x1 -> bn
    -> (+ shortcut) -> relu, the last three fused into one BatchNorm-apply kernel."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, radix=1, cardinality=1, bottleneck_width=64, avd=False,
                 avd_first=False, dilation=1, is_first=False, rectified_conv=False, rectify_avg=False, norm_layer=None,
                 dropblock_prob=0.0, last_gamma=False):
        super().__init__()
        if radix != 2 or rectified_conv or dropblock_prob > 0.0 or dilation != 1 or avd_first:
            raise NotImplementedError("Bottleneck: resnest50 configuration only (radix 2, no dropblock/rectify/dilation/avd_first)")
        group_width = int(planes * (bottleneck_width / 64.)) * cardinality
        self.conv1 = Conv2d(inplanes, group_width, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(group_width)
        self.dropblock_prob = dropblock_prob
        self.radix = radix
        self.avd = avd and (stride > 1 or is_first)
        self.avd_first = avd_first
        if self.avd:
            self.avd_stride = stride
            stride = 1
        self.conv2 = SplAtConv2d(group_width, group_width, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                                 groups=cardinality, bias=False, radix=radix, norm_layer=BatchNorm2d)
        self.conv3 = Conv2d(group_width, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        if last_gamma:
            nn.init.zeros_(self.bn3.weight)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.dilation = dilation
        self.stride = stride

    def forward(self, x):
        # x feeds conv1 AND the shortcut: the shortcut's gradient is parked in `h` and conv1's data gradient adds it in its
        # epilogue (one read) instead of autograd summing the two in a separate kernel (three passes over the tensor)
        h = F_.GradHolder() if (_FUSE_FANOUT and torch.is_grad_enabled() and x.requires_grad) else None
        side = self.downsample is not None and _SIDE_SHORTCUT and self.training and F_.side_branch_ok(x)
        ev = F_.fork_point() if side else None          # x is ready here: the shortcut branch depends on nothing later
        out = conv_bn(self.conv1, self.bn1, x, relu=True, grad_holder=h)
        # (the stash node is created AFTER conv1's: autograd runs it first in the backward pass, so conv1 finds the parked gradient)
        xs = F_.stash_grad(x, h) if h is not None else x
        br = None
        if side:
            # the shortcut (avg-pool -> 1x1 -> BatchNorm) on the second stream, beside conv1 .. conv3 (functional.SideBranch)
            br = F_.SideBranch(after=ev)
            with br:
                residual = self.downsample(xs)
        out = self.conv2(out)
        if self.avd:
            out = F_.avg_pool(out, 3, self.avd_stride, 1)
        if br is not None:
            br.join(residual)
        else:
            residual = self.downsample(xs) if self.downsample is not None else xs
        return conv_bn(self.conv3, self.bn3, out, relu=True, residual=residual)


class _AvgDown(nn.Module):
    """nn.AvgPool2d(k, k, ceil_mode=True, count_include_pad=False) of the avg_down shortcut (ref :383-387);
    parameter-free, so it occupies index 0 of the downsample Sequential like the reference's."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x):
        if self.k == 1:
            return x
        # (a stage's first block pools the previous stage's output, which the decoder's cat consumes as well: the pool's backward
        # kernel adds the cat's gradient slice when the encoder offered a holder for it -- functional.offer_fanout)
        return F_.avg_pool(x, self.k, self.k, 0, ceil_mode=True, count_include_pad=False, fanout=F_.take_fanout())


class _ConvBN(nn.Sequential):
    """Sequential whose forward threads the fused-ReLU flag into its trailing BatchNorm."""

    def forward(self, x):
        return conv_bn(self[1], self[2], self[0](x))


class ResNet(nn.Module):
    """ResNeSt backbone container (ref :277-449) for the deep-stem / avg_down / avd variant."""

    def __init__(self, block, layers, radix=1, groups=1, bottleneck_width=64, num_classes=1000, dilated=False, dilation=1,
                 deep_stem=False, stem_width=64, avg_down=False, rectified_conv=False, rectify_avg=False, avd=False,
                 avd_first=False, final_drop=0.0, dropblock_prob=0, last_gamma=False, norm_layer=BatchNorm2d):
        super().__init__()
        if not deep_stem or not avg_down or dilated or dilation != 1 or rectified_conv or final_drop > 0.0 or dropblock_prob:
            raise NotImplementedError("ResNet: resnest50 configuration only (deep stem, avg_down, no dilation/dropout)")
        self.cardinality = groups
        self.bottleneck_width = bottleneck_width
        self.inplanes = stem_width * 2
        self.avg_down = avg_down
        self.last_gamma = last_gamma
        self.radix = radix
        self.avd = avd
        self.avd_first = avd_first
        self.conv1 = nn.Sequential(
            Conv2d(3, stem_width, kernel_size=3, stride=2, padding=1, bias=False),
            BatchNorm2d(stem_width),
            ReLU(inplace=True),
            Conv2d(stem_width, stem_width, kernel_size=3, stride=1, padding=1, bias=False),
            BatchNorm2d(stem_width),
            ReLU(inplace=True),
            Conv2d(stem_width, stem_width * 2, kernel_size=3, stride=1, padding=1, bias=False),
        )
        self.bn1 = BatchNorm2d(self.inplanes)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool3s2()
        self.layer1 = self._make_layer(block, 64, layers[0], is_first=False)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)   # dropped by the U-Net (compose.py:40-73)
        for m in self.modules():                                   # same init rule as ref :368-374
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, is_first=True):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _ConvBN(_AvgDown(stride), Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=1, bias=False),
                                 BatchNorm2d(planes * block.expansion))
        kw = dict(radix=self.radix, cardinality=self.cardinality, bottleneck_width=self.bottleneck_width, avd=self.avd,
                  avd_first=self.avd_first, norm_layer=BatchNorm2d, last_gamma=self.last_gamma)
        layers = [block(self.inplanes, planes, stride, downsample=downsample, dilation=1, is_first=is_first, **kw)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=1, **kw))
        return nn.Sequential(*layers)

    def stem(self, x):
        c = self.conv1
        x = conv_bn(c[0], c[1], x, relu=True)
        x = conv_bn(c[3], c[4], x, relu=True)
        return conv_bn(c[6], self.bn1, x, relu=True)

    def forward(self, x):
        raise NotImplementedError("the classification forward of ResNet is off the hot path; the U-Net consumes its stages")


class MaxPool3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (ref :340)."""
    kernel_size, stride, padding = 3, 2, 1

    def forward(self, x):
        return F_.max_pool3s2(x, F_.take_fanout())       # (the stem output also feeds decoder_1's cat: see _AvgDown)


class ResNestDecoder(nn.Module):
    """relu( BN(1x1(x)) + relu(SplAt(relu(BN(3x3(x))))) ) (ref :18-43)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Sequential(
            Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False),
            BatchNorm2d(out_channels),
            ReLU(inplace=True),
            SplAtConv2d(out_channels, out_channels, kernel_size=3, padding=1, stride=1, groups=2, radix=2, norm_layer=BatchNorm2d),
            ReLU(inplace=True),
        )
        self.downsample = nn.Sequential(
            Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False),
            BatchNorm2d(out_channels),
        )
        self.relu = ReLU(inplace=True)

    def forward(self, x):
        c = self.conv
        if _FUSE_FANOUT_DEC and self.training and torch.is_grad_enabled() and x.requires_grad:
            # x feeds the 3x3 AND the 1x1 shortcut conv.  The 1x1 runs first here, so its backward runs last and its data
            # gradient adds the 3x3's (parked by stash_grad) in its epilogue: no separate gradient-sum kernel
            h = F_.GradHolder()
            if _SIDE_SHORTCUT_DEC and F_.side_branch_ok(x):
                # the shortcut 1x1 on the second stream, beside the 3x3 -> BatchNorm -> split-attention chain
                br = F_.SideBranch()
                with br:
                    ds = self.downsample[0](x, h)
                out = conv_bn(c[0], c[1], F_.stash_grad(x, h), relu=True)
                out = c[3](out, relu_after=True)
                br.join(ds)
                return self.downsample[1](ds, relu=True, residual=out)
            ds = self.downsample[0](x, h)
            out = conv_bn(c[0], c[1], F_.stash_grad(x, h), relu=True)
            out = c[3](out, relu_after=True)
            return self.downsample[1](ds, relu=True, residual=out)
        out = conv_bn(c[0], c[1], x, relu=True)
        out = c[3](out, relu_after=True)
        # residual branch: BN(1x1(x)) + out, then ReLU -- one fused BatchNorm-apply (training) / folded conv + add (inference)
        return conv_bn(self.downsample[0], self.downsample[1], x, relu=True, residual=out)


class Upsampling(nn.Module):
    """ConvTranspose2d(k=2, s=2) (ref :46-54)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.up = ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)

    def forward(self, x):
        return self.up(x)

    def cat_after(self, skip, x):
        """torch.cat((skip, self(x)), dim=1) (compose.py:141-147 without a crop): the transposed conv stores into its slice of the cat."""
        hooked = any(getattr(m, a, None) for m in (self, self.up)
                     for a in ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks"))
        gm = torch.nn.modules.module
        hooked = hooked or bool(getattr(gm, "_global_forward_hooks", None)) or bool(getattr(gm, "_global_forward_pre_hooks", None))
        if hooked:                       # a module hook must see this module's own output: the two separate ops
            return F_.cat_crop(skip, self(x))
        return F_.upsample_cat(skip, x, self.up.weight, self.up.bias)


def resnest50(pretrained=False, **kwargs):
    """ResNeSt-50 (ref :451-459).  `model_path` points at resnest50-528c19ca.pth when pretrained."""
    model = ResNet(Bottleneck, [3, 4, 6, 3], radix=2, groups=1, bottleneck_width=64, deep_stem=True, stem_width=32, avg_down=True,
                   avd=True, avd_first=False)
    model_path = kwargs.get('model_path', './models/resnest50-528c19ca.pth')
    if pretrained:
        model.load_state_dict(torch.load(model_path))
    return model
