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


SWITCHES = ("FA_W64", "FA_PERSISTENT", "FA_MASKED_W64", "FA_BLOCK_TUNED", "FA_SPLIT", "FA_RESERVED_CUS", "FA_NWAVES",
            "FA_RING", "FA_HD", "FA_VARLEN_W64")


@pytest.fixture
def fa_env(fa):
    """Set the library's environment switches for one test: ``fa_env(FA_W64="1", FA_SPLIT=None)`` (None = unset).
    The library reads them once per process, so every change is followed by a reload; the test's end restores
    the environment it found."""
    saved = {k: os.environ.get(k) for k in SWITCHES}

    def set_(**kv):
        for k, v in kv.items():
            assert k in SWITCHES, k
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        fa.reload_env()

    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    fa.reload_env()
    fa.set_reserved_cus(int(os.environ.get("FA_RESERVED_CUS", "0")))


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
