import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liblvk_oracle.so), built on demand with gcc. Test infrastructure only."""
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback for the HIP path)")
    import livevisionkit_amd as lvk
    c = lvk.Context(0)
    yield c
    c.close()
