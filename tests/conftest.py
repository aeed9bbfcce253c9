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
    # The CPU oracle is most of the GPU suite's wall time, and torch's CPU convolutions of this model peak at 16 threads on
    # the 256-thread GPU host (profiles/r02_cpu_threads.txt: 16 threads 0.81 clips/s, 128 threads 0.05): cap the pool.
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _reset_tune_knobs(request):
    """The fami_*_tune knobs are process-wide: after every GPU test put them back to the library defaults
    (fami_tune_reset), so a test that forgets its `finally` cannot change the route of the tests behind it."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from fami_pose_amd._lib import loaded, lib
        if loaded():
            lib().bind(None)              # (a test that bound a route of its own: back to the process default, then reset THAT)
            lib().cdll.fami_tune_reset()


@pytest.fixture(autouse=True)
def _release_device_objects(request):
    """After every GPU test: collect the garbage NOW (Trainers hold hipGraph executables, and each executable owns the
    HIP streams its parallel branches run on) and let the device drain.  Without it the graph executables of a dozen
    earlier tests are still alive when a later test instantiates and launches its own."""
    yield
    # only the whole-model / trainer modules build graph executables; a full collection after each of the ~200 kernel
    # tests cost 0.4 s apiece (a quarter of the suite's wall time)
    heavy = any(k in request.node.nodeid for k in ('test_model_gpu', 'test_train_gpu', 'test_ddp_gpu'))
    if heavy and (request.node.get_closest_marker('gpu') is not None or 'dev' in request.fixturenames):
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.synchronize()
            gc.collect()
