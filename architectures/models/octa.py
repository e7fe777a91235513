"""Mirror of reference architectures/models/octa.py: ``OctaScribbleNet`` (ref :14-60) -- the container
that owns the segmentor, the discriminator and the loss modules.  Like the reference, ``forward`` is
not implemented (ref :59-60): the training step is ``octave_amd.train.TrainStep``."""
from logging import warning as warn
from typing import Any, Dict, Optional

from torch import Size, Tensor, nn

from architectures.discriminator.blocks import DiscriminatorBlock
from architectures.discriminator.losses import LSDiscriminatorialLoss, LSGeneratorLoss
from architectures.segmentor.compose import ResnestUNet
from architectures.segmentor.losses import DiceLoss, WeightedPartialCE


class OctaScribbleNet(nn.Module):

    def __init__(self, raw_input_shape: Size, mask_input_shape: Size, is_training: bool, pretrian: bool,
                 weight_path: str = 'resnest50-528c19ca.pth', num_classes: int = 2, num_filters: int = 64, instance_noise: bool = True,
                 label_noise: bool = True, segmentor_gating_level: int = 4, discriminator_depth: int = 4, encoder_gating: bool = False,
                 weakly_supervise: bool = True):
        """ScribbleNet-style weakly supervised segmentation model (Valvano et al.) with a ResNeSt U-Net.
        Arguments as in the reference (note the reference's spelling ``pretrian``)."""
        super().__init__()
        if mask_input_shape[1] != num_classes:
            warn('Number channels in mask input is not same as number of classes. Can cause an error when model discriminator is in use.')
        self.segmentor = ResnestUNet(num_classes=num_classes, pretrain=pretrian, weight_path=weight_path,
                                     gating_level=segmentor_gating_level, encoder_gating=encoder_gating)
        if discriminator_depth > 0:
            self.discriminator = DiscriminatorBlock(input_shape=mask_input_shape, is_training=is_training, depth=discriminator_depth,
                                                    num_filters=num_filters, instance_noise=instance_noise, label_noise=label_noise)
        if weakly_supervise:
            self.supervised_loss = WeightedPartialCE(num_classes=num_classes, manual=True)
        else:
            self.supervised_loss = DiceLoss()
        self.discriminatorial_loss = LSDiscriminatorialLoss()
        self.generator_loss = LSGeneratorLoss()
        self.is_train = is_training

    def forward(self, x: Tensor, y: Optional[Tensor] = None) -> Dict[str, Any]:
        raise NotImplementedError
