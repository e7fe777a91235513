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
