import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# MIOpen's exhaustive first-call search costs ~3.5 minutes for the 6 x 900x1600 convolutions of the configs[2] full-size test alone
# (210 of the suite's 730 s); the immediate-mode choice is as correct, only (possibly) slower, and nothing in the tests times MIOpen.
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip():
    """The C-ABI library wrapper; GPU tests call the product only through it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from u2mkd_amd import _lib
    return _lib
