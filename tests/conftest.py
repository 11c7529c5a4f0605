import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)


# TrainStepper(backward="auto") picks the atomic adjoint below 256 bricks (measured faster on the first grids of a progressive schedule).
# The tests train on small grids BECAUSE they are fast to check against the oracle, and they are there to exercise the binned
# machinery (records, brick pass, optimizer in the flush): they keep it on every grid.  The policy itself is tested with the variable
# removed (tests/test_hip_training.py::test_auto_backward_policy).
os.environ.setdefault("RF_AUTO_BINNED_MIN_BRICKS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
