import os
import sys

import pytest

# the library reads its test / developer switches (LT_TEST_*, LT_GEN_ROW_SLOTS, LT_SCORE_FUSED, LT_TAIL_HOST, ...) only in a
# process that opts in: the tests that compare a fast path with its plain form need them
os.environ["LT_ENABLE_TEST_SWITCHES"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as ora
    ora.build()
    return ora


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP extension, loaded; fails loudly (no fallback) if it is missing on a GPU box."""
    from limap_amd import _capi
    from limap_amd.build import build_extension
    if not os.path.exists(_capi.LIB_PATH):
        build_extension()
    return _capi.load_library()
