import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Train-mode end-to-end band rule: |hip - ref64| <= BAND x |ref32 - ref64| + 1e-4 x scale.  The reference's own
# fp32 result is one draw of the rounding noise 93 train-mode BatchNorms amplify; an independent fp32 implementation is another
# draw.  Round 4 tightened the factor from 4 to 2.5 (measured ratios: profiles/r04_band_ratios.txt).
BAND = float(os.environ.get("OCTA_BAND", "2.5"))
# Gradient NORMS are a different matter.  Per parameter they are dominated by a few shared, heavy-tailed factors (the KL term's
# P / Q at the pixels where an attention probability is tiny; the LS-GAN term behind the discriminator's saturating tanh stack),
# so the reference's own fp32 deviation from float64 is ONE draw per fixture -- median 0.06 % at 48 x 48 (B = 3), 0.6 % (B = 6),
# 18 % at 64 x 64, 15 % at 304 x 304, 1.3 % at 400 x 400 -- and the HIP path's is another: measured ratios HIP / reference from
# 0.23 to 10 (profiles/r04_band_ratios.txt).  A ratio of two such draws exceeds 2.5 a quarter of the time even for identical
# distributions, so gradient-norm statistics keep the factor 4 next to absolute floors, and every gradient test also pins the
# well-conditioned gradients (the head, the discriminator's own step) tightly.
BAND_GRAD = float(os.environ.get("OCTA_BAND_GRAD", "4.0"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
