"""octave_amd: MI355X-native (gfx950) kernels + host glue for the OCTAve segmentor/discriminator
training hot path.  The user-facing modules live in the drop-in ``architectures`` package."""
from ._lib import OctaError, lib  # noqa: F401

__version__ = "0.1.0"
