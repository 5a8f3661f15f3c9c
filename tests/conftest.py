import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"
HAS_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "TFRecModel"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


needs_reference = pytest.mark.skipif(not HAS_REFERENCE, reason="/root/reference not mounted (GPU box)")


@pytest.fixture(scope="session")
def samples():
    """First 256 rows of the reference's testSamples.csv as raw string columns (golden fixture)."""
    z = np.load(os.path.join(GOLDEN, "samples_256.npz"))
    return {k: z[k].astype(object) for k in z.files}


@pytest.fixture(scope="session")
def lib():
    from sparrowrecsys_amd import _lib
    _lib.build_library()
    return _lib.load_library()
