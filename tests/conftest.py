import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The PyTorch wheel brings its own copy of the HIP runtime; libntedit_hip.so links the system one.  Both can live
    # in one process as long as torch's is initialised FIRST (a torch.cuda call after the library has opened the
    # device reports "no ROCm-capable device").  Tests that use torch only as plumbing (synthetic data in HBM,
    # torch.distributed) must not depend on which test ran before them.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_build():
    import helpers
    return helpers.build_oracle()
