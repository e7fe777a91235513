"""Mirror of reference architectures/segmentor/blocks.py for the hot path:
``AdversarialAttentionGate`` (ref :12-46) and ``GlobalAveragePooling2D`` (ref :349-354).
The propagation/aggregation baselines of that file are off the hot path."""
from typing import Tuple

from torch import Tensor, nn

from octave_amd import functional as F_
from octave_amd.layers import Conv2d


class AdversarialAttentionGate(nn.Module):
    """1x1 conv to `out_channels` class maps, channel softmax, and the Hadamard product of the input
    with the summed foreground maps -- one fused kernel (2 <= out_channels <= 4).

    outputs: (masked_x, y_hat); y_hat is a dense fp32 (B, classes, H, W) tensor."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv1 = Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=(1, 1), stride=1)
        self.softmax = nn.Softmax(dim=1)   # attribute kept for API parity; the softmax is fused

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        masked_x, y_hat = F_.attention_gate(x, self.conv1.weight, self.conv1.bias)
        return masked_x, y_hat


class GlobalAveragePooling2D(nn.Module):
    """Spatial mean -> (B, C) (ref :349-354), used by the classification heads."""

    def __init__(self):
        super().__init__()

    def forward(self, x: Tensor):
        if F_.nhwc_ld(x) is not None and x.shape[1] > 1 and x.stride(1) == 1:
            x = F_.to_nchw_f32(x)                     # NHWC activation -> dense fp32 NCHW
        return F_.adaptive_avg_pool(x, 1).flatten(1)  # one launch: the (1, 1) adaptive pool is the spatial mean
