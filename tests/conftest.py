import os
import sys
# ROCm runtime: kernel arguments in device memory -- 2-3 us less launch latency per kernel; with ~4000 dependent launches
# per training step that is -4 % (f32) / -5 % (bf16) step time on MI355X.  Must be set before the HIP runtime initialises.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
