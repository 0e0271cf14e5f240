import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# the device front end's scratch buffers start as 0xA5 bytes under test (csrc/gam_kernels.hip: GBuf, csrc/hc_flatten_kernels.hip: DBuf): a
# kernel that leaves an entry unwritten then fails here, not in the one run whose fresh memory happens not to be zero.  Inherited by the
# `vgan` binaries the tests start.
os.environ.setdefault("VGAN_POISON_ALLOCS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")
