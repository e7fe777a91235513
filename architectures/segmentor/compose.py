"""Mirror of reference architectures/segmentor/compose.py: ``ResnestUNet`` (ref :12-230), ``ResnestUnetParallelHead``
(ref :233-362) and ``ResnestUnetParallelHeadAttentionGate`` (ref :365-527).

ResNeSt-50 stages as the U-Net encoder, five ConvTranspose/ResNestDecoder/attention-gate decoder
levels, a 1x1 head.  ``forward`` returns ``(attentions, agg_map, x_4)`` exactly like ref :100-187
(attentions finest first; class maps are dense fp32 NCHW; x_4 is an NHWC-strided activation).
Activations flow NHWC in ``compute_dtype`` (None = dtype of the input: float32, bfloat16 or float16).
``predict`` post-processing (softmax / sigmoid / one-hot) and the classification heads run on HIP
kernels as well; in eval mode (no grad) every conv -> BatchNorm pair is one conv launch with the
running statistics folded into its packed weights (octave_amd.layers.conv_bn).
Encoder gating (reference default off; it changes the return arity) is not built.
"""
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from architectures.extra.resnest import ResNestDecoder, Upsampling, resnest50
from architectures.segmentor.blocks import AdversarialAttentionGate, GlobalAveragePooling2D
from octave_amd import functional as F_
from octave_amd.layers import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Linear, ReLU, conv_bn, use_channels_last_weights
from octave_amd._lib import ACT_RELU


def _activation_dtype(module, x: Tensor):
    return module.compute_dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32)


def _encode(net, x: Tensor):
    """Stem + the four ResNeSt stages shared by the three U-Nets (ref :100-132 / :292-311 / :440-459): returns
    x_0_0, x_1, x_2, x_3 (padded to even size), x_4 and the padding flags."""
    mark = F_.stage_mark
    conv1, bn1 = net.encoder_0_1_2[0], net.encoder_0_1_2[1]
    x = conv_bn(conv1[0], conv1[1], x, relu=True)
    x = conv_bn(conv1[3], conv1[4], x, relu=True)
    x_0_0 = conv_bn(conv1[6], bn1, x, relu=True)
    # x_0_0 .. x_3 each feed the next encoder stage AND a decoder's cat.  The stage's first pooling op on the tensor (the stem's
    # max-pool, the avg_down shortcut's pool) takes the holder offered here and adds the cat's gradient slice in its backward
    # kernel; the decoders get the tensor through stash_grad, which parks that slice (no separate gradient-sum launch).
    try:
        h0 = F_.offer_fanout(x_0_0)
        x_0_1 = net.encoder_0_2_2(x_0_0)
        F_.withdraw_fanout()
        x_1 = net.encoder_1(mark(x_0_1, "encoder_1"))
        h1 = F_.offer_fanout(x_1)
        x_2 = net.encoder_2(mark(x_1, "encoder_2"))
        h2 = F_.offer_fanout(x_2)
        x_3 = net.encoder_3(mark(x_2, "encoder_3"))
        pad_h, pad_w = x_3.shape[2] % 2, x_3.shape[3] % 2
        if pad_h or pad_w:                                   # ref :125-130
            x_3 = F_.pad_bottom_right(x_3, pad_h, pad_w)
        h3 = F_.offer_fanout(x_3)
        x_4 = net.encoder_4(mark(x_3, "encoder_4"))
    finally:
        F_.withdraw_fanout()                                 # (an offer nobody took -- or an exception -- must not reach another network's pool)
    return (F_.skip_with_fanout(x_0_0, h0), F_.skip_with_fanout(x_1, h1), F_.skip_with_fanout(x_2, h2), F_.skip_with_fanout(x_3, h3),
            x_4, pad_h, pad_w)


def _check_input(x: Tensor, who: str):
    if x.dim() != 4 or x.shape[2] % 16 or x.shape[3] % 16:
        raise ValueError(f"{who} needs (B, 3, H, W) input with H and W multiples of 16, got {tuple(x.shape)}")


def _stack_heads(a: Tensor, b: Tensor) -> Tensor:
    """rearrange([agg_map, agg_map_c], 'k b c h w -> k b c h w') (ref :346, :513)."""
    return torch.stack((a, b), dim=0)


def _predict_stacked(agg: Tensor, method: str) -> Tensor:
    """predict() of the parallel-head nets on the (2, B, C, H, W) stack (ref :348-358): softmax / argmax over dim 2."""
    k, B, C, H, W = agg.shape
    flat = agg.reshape(k * B, C, H, W)
    if method == 'softmax':
        return F_.class_softmax(flat).view(k, B, C, H, W)
    if method == 'sigmoid':
        return F_.predict_sigmoid(flat).view(k, B, C, H, W)
    if method == 'one-hot':
        oh = F_.predict_one_hot(flat)
        return oh.reshape(k, B, oh.shape[1], H, W)
    if method == 'original':
        return agg
    raise ValueError(method)


class ResnestUNet(nn.Module):

    def __init__(self, num_classes: int, pretrain: bool, weight_path: str = None, gating_level: int = 4, encoder_gating: bool = False):
        super().__init__()
        if encoder_gating:
            raise NotImplementedError("encoder_gating=True is off the hot path (reference default is False; SURVEY.md 2d)")
        resnest = resnest50(pretrained=pretrain, model_path=weight_path)
        self.gating_level = gating_level
        self.encoder_gating = encoder_gating
        self.compute_dtype: Optional[torch.dtype] = None

        # Depth 0 (registration order follows ref :40-79 so state_dict order matches)
        self.encoder_0_1_2 = nn.Sequential(resnest.conv1, resnest.bn1, resnest.relu)
        self.encoder_0_2_2 = resnest.maxpool
        self.upsampling_0 = Upsampling(64, 64)
        self.decoder_0 = ResNestDecoder(64, 32)
        self.aag_0 = AdversarialAttentionGate(32, num_classes)
        # Depth 1
        self.encoder_1 = resnest.layer1
        self.upsampling_1 = Upsampling(256, 64)
        self.decoder_1 = ResNestDecoder(128, 64)
        self.aag_1 = AdversarialAttentionGate(64, num_classes)
        # Depth 2
        self.encoder_2 = resnest.layer2
        self.aag_2 = AdversarialAttentionGate(256, num_classes)
        self.upsampling_2 = Upsampling(512, 256)
        self.decoder_2 = ResNestDecoder(512, 256)
        # Depth 3
        self.encoder_3 = resnest.layer3
        self.upsampling_3 = Upsampling(1024, 512)
        self.decoder_3 = ResNestDecoder(1024, 512)
        self.aag_3 = AdversarialAttentionGate(512, num_classes)
        # Depth 4
        self.encoder_4 = resnest.layer4
        self.upsampling_4 = Upsampling(2048, 1024)
        self.decoder_4 = ResNestDecoder(2048, 1024)
        self.aag_4 = AdversarialAttentionGate(1024, num_classes)

        self.fc = Conv2d(in_channels=32, out_channels=num_classes, kernel_size=1, stride=1)

        # classification heads: parameters exist for state_dict parity, never touched by forward
        # (they receive no gradient in the reference either)
        self.linear_head_emb = nn.Sequential(GlobalAveragePooling2D(), Linear(2048, num_classes))
        self.linear_head_dec = nn.Sequential(
            AdaptiveAvgPool2d((32, 32)), Conv2d(in_channels=num_classes, out_channels=64, kernel_size=7, act=ACT_RELU), ReLU(inplace=True),
            BatchNorm2d(num_features=64), Conv2d(in_channels=64, out_channels=512, kernel_size=7, act=ACT_RELU), ReLU(inplace=True),
            BatchNorm2d(num_features=512), GlobalAveragePooling2D(), Linear(512, num_classes))
        use_channels_last_weights(self)

    def forward(self, x: Tensor):
        _check_input(x, "ResnestUNet")
        x = F_.to_nhwc(x, dtype=_activation_dtype(self, x), cpad=8)
        # F_.stage_mark is an identity: it only tells the training step where, in the BACKWARD pass, a stage's parameter
        # gradients are complete (deferred weight-gradient flush, gradient buckets); a no-op outside a TrainStep
        mark = F_.stage_mark
        x_0_0, x_1, x_2, x_3, x_4, pad_h, pad_w = _encode(self, x)

        attentions = []
        # Bottom-Up
        d_4 = self.upsampling_4(x_4)
        d_4 = F_.cat_crop(x_3, d_4, x_3.shape[2] - pad_h, x_3.shape[3] - pad_w)   # cat + crop, ref :141-147
        d_4 = self.decoder_4(mark(d_4, "decoder_4"))
        if self.gating_level >= 4:
            d_4, y_4 = self.aag_4(d_4)
            attentions.append(y_4)
        d_3 = self.decoder_3(mark(self.upsampling_3.cat_after(x_2, d_4), "decoder_3"))
        if self.gating_level >= 3:
            d_3, y_3 = self.aag_3(d_3)
            attentions.append(y_3)
        d_2 = self.decoder_2(mark(self.upsampling_2.cat_after(x_1, d_3), "decoder_2"))
        if self.gating_level >= 2:
            d_2, y_2 = self.aag_2(d_2)
            attentions.append(y_2)
        d_1 = self.decoder_1(mark(self.upsampling_1.cat_after(x_0_0, d_2), "decoder_1"))
        if self.gating_level >= 1:
            d_1, y_1 = self.aag_1(d_1)
            attentions.append(y_1)
        d_0 = self.decoder_0(mark(self.upsampling_0(d_1), "decoder_0"))
        if self.gating_level >= 0:
            d_0, y_0 = self.aag_0(d_0)
            attentions.append(y_0)
        agg_map = F_.head_1x1(d_0, self.fc.weight, self.fc.bias)
        attentions.reverse()
        return tuple(attentions), agg_map, x_4

    def predict(self, x: Tensor, method: Literal['softmax', 'one-hot', 'original', 'sigmoid'] = 'softmax'):
        """ref :189-199; the post-processing of the (B, classes, H, W) logits is one HIP launch per method."""
        attentions, agg_map, _ = self.forward(x)
        if method == 'softmax':
            predicate = F_.class_softmax(agg_map)
        elif method == 'sigmoid':
            predicate = F_.predict_sigmoid(agg_map)
        elif method == 'one-hot':
            predicate = F_.predict_one_hot(agg_map)
        elif method == 'original':
            predicate = agg_map
        else:
            raise ValueError(method)
        return attentions, predicate

    def classification_predict(self, x: Tensor, method: Literal['softmax', 'sigmoid'],
                               mode: Literal['classic', 'classic-gating', 'ae-squash', 'ae-extract']):
        """ref :201-230.  'classic': linear head on the pooled bottleneck x_4; 'ae-squash': pooled soft prediction;
        'ae-extract': the small conv head on the soft prediction.  'classic-gating' needs encoder gating (not built)."""
        if mode == 'classic-gating':
            raise ValueError(f'{mode} is not valid if `encoder_gating` is not enabled.')
        att, predicate, latent = self.forward(x)
        predicate = F_.class_softmax(predicate)
        if mode == 'classic':
            emb = self.linear_head_emb(latent)
        elif mode == 'ae-squash':
            emb = GlobalAveragePooling2D()(predicate)
        elif mode == 'ae-extract':
            h = self.linear_head_dec
            e = h[1](h[0](predicate))                 # adaptive pool -> conv7 (+ReLU fused)
            e = h[4](h[3](e))                         # BN -> conv7 (+ReLU fused)
            emb = h[8](h[7](h[6](e)))                 # BN -> global pool -> linear
        else:
            raise NotImplementedError
        emb4 = emb.view(emb.shape[0], emb.shape[1], 1, 1)
        if method == 'softmax':
            class_pred = F_.class_softmax(emb4).view_as(emb)
        elif method == 'sigmoid':
            class_pred = F_.predict_sigmoid(emb4).view_as(emb)
        else:
            raise NotImplementedError
        return class_pred, att, predicate


class _ParallelHeadBase(nn.Module):
    """Shared construction of the two dual-head U-Nets (e.g. ROSE SVC / DVC): the main decoder path plus a second, shallow
    decoder branch (depth 1 and 0) fed by encoder_1, each with its own 1x1 head.  Registration order follows the reference
    so that state_dict keys and order match."""

    def _build(self, num_classes: int, pretrain: bool, weight_path: Optional[str], gates: bool):
        resnest = resnest50(pretrained=pretrain, model_path=weight_path)
        self.compute_dtype: Optional[torch.dtype] = None
        self.encoder_0_1_2 = nn.Sequential(resnest.conv1, resnest.bn1, resnest.relu)
        self.encoder_0_2_2 = resnest.maxpool
        self.upsampling_0 = Upsampling(64, 64)
        self.decoder_0 = ResNestDecoder(64, 32)
        if gates:
            self.aag_0 = AdversarialAttentionGate(32, num_classes)
        self.encoder_1 = resnest.layer1
        self.upsampling_1 = Upsampling(256, 64)
        self.decoder_1 = ResNestDecoder(128, 64)
        if gates:
            self.aag_1 = AdversarialAttentionGate(64, num_classes)
        self.encoder_2 = resnest.layer2
        self.upsampling_2 = Upsampling(512, 256)
        self.decoder_2 = ResNestDecoder(512, 256)
        if gates:
            self.aag_2 = AdversarialAttentionGate(256, num_classes)
        self.encoder_3 = resnest.layer3
        self.upsampling_3 = Upsampling(1024, 512)
        self.decoder_3 = ResNestDecoder(1024, 512)
        if gates:
            self.aag_3 = AdversarialAttentionGate(512, num_classes)
        self.encoder_4 = resnest.layer4
        self.upsampling_4 = Upsampling(2048, 1024)
        self.decoder_4 = ResNestDecoder(2048, 1024)
        if gates:
            self.aag_4 = AdversarialAttentionGate(1024, num_classes)
        self.upsampling_1_c = Upsampling(256, 64)
        self.decoder_1_c = ResNestDecoder(128, 64)
        if gates:
            self.aag_1_c = AdversarialAttentionGate(64, num_classes)
        self.upsampling_0_c = Upsampling(64, 64)
        self.decoder_0_c = ResNestDecoder(64, 32)
        if gates:
            self.aag_0_c = AdversarialAttentionGate(32, num_classes)
        self.fc = Conv2d(in_channels=32, out_channels=num_classes, kernel_size=1, stride=1)
        self.fc_c = Conv2d(in_channels=32, out_channels=num_classes, kernel_size=1, stride=1)
        use_channels_last_weights(self)

    def _run(self, x: Tensor, level: int):
        """level = gating level (-1: no gates).  Returns (attentions, attentions_c, agg_map, agg_map_c)."""
        _check_input(x, type(self).__name__)
        x = F_.to_nhwc(x, dtype=_activation_dtype(self, x), cpad=8)
        x_0_0, x_1, x_2, x_3, x_4, pad_h, pad_w = _encode(self, x)
        att, att_c = [], []

        def gate(name, d, thr, lst, strict=False):
            on = hasattr(self, name) and (level > thr if strict else level >= thr)
            if on:
                d, y = getattr(self, name)(d)
                lst.append(y)
            return d
        d_4 = self.decoder_4(F_.cat_crop(x_3, self.upsampling_4(x_4), x_3.shape[2] - pad_h, x_3.shape[3] - pad_w))
        d_4 = gate("aag_4", d_4, 3, att, strict=True)          # ref :473: `> 3`
        d_3 = gate("aag_3", self.decoder_3(self.upsampling_3.cat_after(x_2, d_4)), 3, att)
        d_2 = gate("aag_2", self.decoder_2(self.upsampling_2.cat_after(x_1, d_3)), 2, att)
        d_1 = gate("aag_1", self.decoder_1(self.upsampling_1.cat_after(x_0_0, d_2)), 1, att)
        d_0 = gate("aag_0", self.decoder_0(self.upsampling_0(d_1)), 0, att)
        d_1_c = gate("aag_1_c", self.decoder_1_c(self.upsampling_1_c.cat_after(x_0_0, x_1)), 1, att_c)
        d_0_c = gate("aag_0_c", self.decoder_0_c(self.upsampling_0_c(d_1_c)), 0, att_c)
        att.reverse()
        att_c.reverse()
        agg = F_.head_1x1(d_0, self.fc.weight, self.fc.bias)
        agg_c = F_.head_1x1(d_0_c, self.fc_c.weight, self.fc_c.bias)
        return tuple(att), tuple(att_c), agg, agg_c


class ResnestUnetParallelHead(_ParallelHeadBase):
    """ref :233-362: two 1x1 heads, no attention gates; forward returns the (2, B, C, H, W) logits stack."""

    def __init__(self, num_classes: int, pretrain: bool, weight_path: str = None):
        super().__init__()
        self._build(num_classes, pretrain, weight_path, gates=False)

    def forward(self, x) -> Tensor:
        _, _, agg, agg_c = self._run(x, -1)
        return _stack_heads(agg, agg_c)

    def predict(self, x: Tensor, method: Literal['softmax', 'sigmoid', 'one-hot', 'original'] = 'softmax'):
        return _predict_stacked(self.forward(x), method)


class ResnestUnetParallelHeadAttentionGate(_ParallelHeadBase):
    """ref :365-527: the dual-head net with attention gates on both decoder branches; forward returns
    ((attentions, attentions_c), (2, B, C, H, W) logits).  The constructor keyword keeps the reference's spelling."""

    def __init__(self, num_classes: int, pretrain: bool, weight_path: str = None, gating_leveL: int = 3):
        super().__init__()
        self.gating_level = gating_leveL
        self._build(num_classes, pretrain, weight_path, gates=True)

    def forward(self, x):
        att, att_c, agg, agg_c = self._run(x, self.gating_level)
        return (att, att_c), _stack_heads(agg, agg_c)

    def predict(self, x: Tensor, method: Literal['softmax', 'sigmoid', 'one-hot', 'original'] = 'softmax'):
        attentions, agg = self.forward(x)
        return attentions, _predict_stacked(agg, method)
