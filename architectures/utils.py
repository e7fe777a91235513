"""Mirror of reference architectures/utils.py (only what the hot path uses)."""
import math
from typing import Optional

import torch
from torch import Tensor


def get_same_padding_conv(input_size: int, kernel_size: int, stride: int):
    """Padding giving n_out == n_in for a square kernel (reference utils.py:8-13)."""
    return math.ceil(((stride * (input_size - 1)) - input_size + kernel_size) / 2)


def get_same_padding_transpose(input_size: int, kernel_size: int, stride: int):
    """reference utils.py:16-18."""
    return (stride - (input_size * (1 - stride)) + kernel_size) // 2


def rand_uniform(x: Optional[Tensor] = None):
    """One U[0,1) draw from the global CPU generator, as reference utils.py:20-22 (the order in
    which the CPU generator is consumed is part of the parity contract)."""
    rand = torch.FloatTensor(1).uniform_(0, 1)
    # the reference ends with `.type_as(x)`.  For a CPU `x` that is reproduced; for a device tensor the draw stays on the host:
    # its only use is a Python-side comparison (`rand < prob`), and shipping it to the device just to read it back costs a
    # host-device round trip per discriminator call
    if x is not None and not x.is_cuda:
        rand = rand.type_as(x)
    return rand
