"""Mirror of reference architectures/segmentor/compose.py: ``ResnestUNet`` (ref :12-230).

ResNeSt-50 stages as the U-Net encoder, five ConvTranspose/ResNestDecoder/attention-gate decoder
levels, a 1x1 head.  ``forward`` returns ``(attentions, agg_map, x_4)`` exactly like ref :100-187
(attentions finest first; class maps are dense fp32 NCHW; x_4 is an NHWC-strided activation).
Activations flow NHWC in ``compute_dtype`` (None = dtype of the input: float32 or bfloat16).
The parallel-head variants (ref :233-527) and encoder gating (default off) are off the hot path.
"""
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from architectures.extra.resnest import ResNestDecoder, Upsampling, resnest50
from architectures.segmentor.blocks import AdversarialAttentionGate, GlobalAveragePooling2D
from octave_amd import functional as F_
from octave_amd.layers import Conv2d, use_channels_last_weights


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
        self.linear_head_emb = nn.Sequential(GlobalAveragePooling2D(), nn.Linear(2048, num_classes))
        self.linear_head_dec = nn.Sequential(
            nn.AdaptiveAvgPool2d((32, 32)), nn.Conv2d(in_channels=num_classes, out_channels=64, kernel_size=7), nn.ReLU(inplace=True),
            nn.BatchNorm2d(num_features=64), nn.Conv2d(in_channels=64, out_channels=512, kernel_size=7), nn.ReLU(inplace=True),
            nn.BatchNorm2d(num_features=512), GlobalAveragePooling2D(), nn.Linear(512, num_classes))
        use_channels_last_weights(self)

    def _encode_stem(self, x):
        conv1, bn1 = self.encoder_0_1_2[0], self.encoder_0_1_2[1]
        x = conv1[1](conv1[0](x), relu=True)
        x = conv1[4](conv1[3](x), relu=True)
        return bn1(conv1[6](x), relu=True)

    def forward(self, x: Tensor):
        if x.dim() != 4 or x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"ResnestUNet needs (B, 3, H, W) input with H and W multiples of 16, got {tuple(x.shape)}")
        dtype = self.compute_dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        x = F_.to_nhwc(x, dtype=dtype, cpad=8)
        # Top-Down
        # F_.stage_mark is an identity: it only tells the training step where, in the BACKWARD pass, a stage's parameter
        # gradients are complete (deferred weight-gradient flush, gradient buckets); a no-op outside a TrainStep
        mark = F_.stage_mark
        x_0_0 = self._encode_stem(x)
        x_0_1 = self.encoder_0_2_2(x_0_0)
        x_1 = self.encoder_1(mark(x_0_1, "encoder_1"))
        x_2 = self.encoder_2(mark(x_1, "encoder_2"))
        x_3 = self.encoder_3(mark(x_2, "encoder_3"))
        pad_h, pad_w = x_3.shape[2] % 2, x_3.shape[3] % 2
        if pad_h or pad_w:                                   # ref :125-130
            x_3 = F_.pad_bottom_right(x_3, pad_h, pad_w)
        x_4 = self.encoder_4(mark(x_3, "encoder_4"))

        attentions = []
        # Bottom-Up
        d_4 = self.upsampling_4(x_4)
        d_4 = F_.cat_crop(x_3, d_4, x_3.shape[2] - pad_h, x_3.shape[3] - pad_w)   # cat + crop, ref :141-147
        d_4 = self.decoder_4(mark(d_4, "decoder_4"))
        if self.gating_level >= 4:
            d_4, y_4 = self.aag_4(d_4)
            attentions.append(y_4)
        d_3 = self.decoder_3(mark(F_.cat_crop(x_2, self.upsampling_3(d_4)), "decoder_3"))
        if self.gating_level >= 3:
            d_3, y_3 = self.aag_3(d_3)
            attentions.append(y_3)
        d_2 = self.decoder_2(mark(F_.cat_crop(x_1, self.upsampling_2(d_3)), "decoder_2"))
        if self.gating_level >= 2:
            d_2, y_2 = self.aag_2(d_2)
            attentions.append(y_2)
        d_1 = self.decoder_1(mark(F_.cat_crop(x_0_0, self.upsampling_1(d_2)), "decoder_1"))
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
        """ref :189-199.  The post-processing of the (B, classes, H, W) logits is host-level glue."""
        attentions, agg_map, _ = self.forward(x)
        if method == 'softmax':
            predicate = F_.class_softmax(agg_map)
        elif method == 'sigmoid':
            predicate = torch.sigmoid(agg_map)
        elif method == 'one-hot':
            predicate = torch.nn.functional.one_hot(torch.argmax(agg_map, dim=1)).permute(0, 3, 1, 2)
        elif method == 'original':
            predicate = agg_map
        else:
            raise ValueError(method)
        return attentions, predicate

    def classification_predict(self, *args, **kwargs):
        raise NotImplementedError("classification heads are off the hot path (SURVEY.md 2c)")
