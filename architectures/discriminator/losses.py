"""Mirror of reference architectures/discriminator/losses.py (LS-GAN objectives, ref :6-24)."""
from torch import Tensor, nn

from octave_amd import functional as F_


class LSDiscriminatorialLoss(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, y_real: Tensor, y_fake: Tensor):
        return F_.lsgan_discriminator(y_real, y_fake)


class LSGeneratorLoss(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, y_fake: Tensor):
        return F_.lsgan_generator(y_fake)
