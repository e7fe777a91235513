"""Mirror of reference architectures/segmentor/losses.py for the hot path: ``WeightedPartialCE``
(ref :11-61), ``DiceLoss`` (:64-74), ``InterlayerDivergence`` (:90-172).  Each forward is one fused
reduction on the GPU (ref: 10-30 ATen launches each).  ``ImageMseLoss``/``CELoss`` (thin wrappers over
torch losses, off the hot path) are not provided.
"""
import logging
from typing import Literal, Optional, Sequence

import torch
from torch import Tensor, nn

from octave_amd import functional as F_

logger = logging.getLogger("octave_amd")


class WeightedPartialCE(nn.Module):

    def __init__(self, num_classes, eps=1e-12, manual: bool = False):
        """Weighted partial cross entropy over scribble pixels (manual=True form, the one
        ``OctaScribbleNet`` builds, models/octa.py:52).  `y_hat` are class probabilities."""
        super().__init__()
        self.num_classes = num_classes
        self.eps = eps
        self.manual = manual

    def forward(self, y_hat: Tensor, ys: Tensor, ignore_bg: bool = False, reduction: Literal['mean', 'sum'] = 'mean', **kwargs) -> Tensor:
        assert y_hat.shape[1] == ys.shape[1], 'Number of class mismatch.'
        if not self.manual or self.num_classes == 1:
            raise NotImplementedError("WeightedPartialCE: only manual=True with num_classes >= 2 is on the hot path")
        if reduction not in ('mean', 'sum'):
            raise ValueError(reduction)
        if ignore_bg:
            ys[:, 0] = 0        # in place on the caller's tensor, like ref :29-30
        return F_.wpce_dice(y_hat, ys, from_logits=False, full=bool(kwargs.get('full', False)), reduction_sum=(reduction == 'sum'))[0]

    def forward_logits(self, logits: Tensor, ys: Tensor, **kwargs) -> Tensor:
        """Same loss on raw logits with the caller's nn.Softmax(dim=1) fused into the kernel."""
        return F_.wpce_dice(logits, ys, from_logits=True, full=bool(kwargs.get('full', False)))[0]


class DiceLoss(nn.Module):

    def __init__(self, eps: float = 1e-12):
        super().__init__()
        self.eps = eps

    def forward(self, input: Tensor, target: Tensor):
        return F_.wpce_dice(input, target, from_logits=False)[1]


class InterlayerDivergence(nn.Module):

    def __init__(self, mode: Literal['mean', 'sum'] = 'mean', eps: float = 1e-12, upscaling_mode: Literal['nn', 'deconv'] = 'nn',
                 stop_gradient: bool = False, divergence: Literal['KLD', 'JSD'] = 'KLD'):
        """KL divergence between the finest prediction (attentions[0]) and the nearest-up-sampled
        coarser attention maps.  check_nan (default True) keeps the reference's NaN raise (ref
        :140-142) at the price of one host sync; set it False inside a training loop."""
        super().__init__()
        assert mode in ['mean', 'sum'], f'mode {mode} is not exists/implemented.'
        self.mode = mode
        self.eps = eps
        self.stop_gradient = stop_gradient
        self.divergence = divergence
        self.check_nan = True

    def forward(self, attentions: Sequence[Tensor], weights: Optional[list] = None) -> Tensor:
        n = len(attentions) - 1
        if weights is None:
            weights = [1] * n
        elif len(weights) != n:
            weights = weights[:len(attentions)]      # truncate, ref :121-123
        if self.divergence == 'KLD':
            if self.mode == 'sum':
                raise NotImplementedError('Not implemented yet.')
            out = F_.interlayer_kl(attentions, weights, self.stop_gradient)
            if self.check_nan and bool(out[1] != 0):
                logger.error('Divergence is NaN')
                raise Exception('Divergence is NaN')
            return out[0]
        elif self.divergence == 'JSD':
            raise NotImplementedError("JSD branch is off the hot path (SURVEY.md 8f rank 4)")
        raise NotImplementedError(f'Invalid divergence type / Not implemented: {self.divergence}')
