"""Mirror of reference architectures/segmentor/losses.py for the hot path: ``WeightedPartialCE``
(ref :11-61), ``DiceLoss`` (:64-74), ``InterlayerDivergence`` (:90-172).  Each forward is one fused
reduction on the GPU (ref: 10-30 ATen launches each).  ``ImageMseLoss``/``CELoss`` (thin wrappers over
torch losses, off the hot path) are not provided.  All branches of the three classes run on HIP kernels: the manual
WPCE, its nn.CrossEntropyLoss / nn.BCEWithLogitsLoss forms, KLD and JSD.
"""
import logging
from typing import Literal, Optional, Sequence

import torch
from torch import Tensor, nn

from octave_amd import functional as F_

logger = logging.getLogger("octave_amd")


def _check_eps(eps: float, who: str) -> float:
    """The HIP loss kernels carry the reference's default smoothing constant (1e-12) as a literal; any other value would be
    silently ignored, so it is refused instead."""
    if float(eps) != 1e-12:
        raise NotImplementedError(f"{who}: eps={eps} is not supported by the HIP loss kernels (compiled for the reference default 1e-12)")
    return float(eps)


class WeightedPartialCE(nn.Module):

    def __init__(self, num_classes, eps=1e-12, manual: bool = False):
        """Weighted partial cross entropy over scribble pixels (manual=True form, the one
        ``OctaScribbleNet`` builds, models/octa.py:52).  `y_hat` are class probabilities."""
        super().__init__()
        self.num_classes = num_classes
        self.eps = _check_eps(eps, "WeightedPartialCE")
        self.manual = manual

    def forward(self, y_hat: Tensor, ys: Tensor, ignore_bg: bool = False, reduction: Literal['mean', 'sum'] = 'mean', **kwargs) -> Tensor:
        assert y_hat.shape[1] == ys.shape[1], 'Number of class mismatch.'
        if reduction not in ('mean', 'sum'):
            raise ValueError(reduction)
        if ignore_bg:
            ys[:, 0] = 0        # in place on the caller's tensor, like ref :29-30
        full = bool(kwargs.get('full', False))
        if self.num_classes == 1:
            # ref :48-49: nn.BCEWithLogitsLoss()(y_hat * ys, ys); with manual=False the reference first drops every channel of a
            # one-channel ys (ys[:, 1:]) and fails on the shape -- the same ValueError is raised here
            if not self.manual:
                raise ValueError(f"Target size (torch.Size([0])) must be the same as input size ({tuple(y_hat.permute(0, 2, 3, 1).reshape(-1, 1).shape)})")
            return F_.pixel_ce(y_hat, ys, full=full, mode=1)
        if not self.manual:
            # ref :58: nn.CrossEntropyLoss()(masked scores, long(ys[:, 1:])) -- defined for exactly two classes (for more the
            # flattened target has (C-1) entries per pixel and the reference raises on the batch-size mismatch)
            if y_hat.shape[1] != 2:
                n = y_hat.shape[0] * y_hat.shape[2] * y_hat.shape[3]
                raise ValueError(f"Expected input batch_size ({n}) to match target batch_size ({n * (y_hat.shape[1] - 1)}).")
            return F_.pixel_ce(y_hat, ys, full=full, mode=0)
        return F_.wpce_dice(y_hat, ys, from_logits=False, full=full, reduction_sum=(reduction == 'sum'))[0]

    def forward_logits(self, logits: Tensor, ys: Tensor, **kwargs) -> Tensor:
        """Same loss on raw logits with the caller's nn.Softmax(dim=1) fused into the kernel."""
        return F_.wpce_dice(logits, ys, from_logits=True, full=bool(kwargs.get('full', False)))[0]


class DiceLoss(nn.Module):

    def __init__(self, eps: float = 1e-12):
        super().__init__()
        self.eps = _check_eps(eps, "DiceLoss")

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
        self.eps = _check_eps(eps, "InterlayerDivergence(KLD)") if divergence == 'KLD' else float(eps)     # the JSD kernels take eps as an argument
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
            # ref :154-169 (no NaN check and no `mode` there either)
            return F_.interlayer_jsd(attentions, weights, self.stop_gradient, self.eps)[0]
        raise NotImplementedError(f'Invalid divergence type / Not implemented: {self.divergence}')
