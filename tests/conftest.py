import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity gate on O against the CPU baseline on identical fp16 inputs.
MAX_ABS_TOL = 2e-3
REL_L2_TOL = 1e-3


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def fa():
    import __graft_entry__ as ge

    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle as o

    o.build()
    return o


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def normal_fp16(seed, shape):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(shape, dtype=np.float32).astype(np.float16) for _ in range(3)]
