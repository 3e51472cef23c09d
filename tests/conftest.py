import os
import sys

import pytest

# the tests' scenes are small: keep their weight gradients on the SIDE stream (the path the 8-scene batch takes: fork / join
# events, bucket-slot events of BucketedDDP) instead of the small-batch inline path; test_small_batches_inline_their_weight_gradients
# covers the latter
os.environ.setdefault("LGS_WGRAD_INLINE_BELOW", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as languagegroundedsemseg_amd/__init__.py sets it (before the HIP runtime starts)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# ---- dispatch coverage (tests/test_gpu_dispatch_coverage.py): every GPU test's kernel launch sites are recorded from the engine's
# dispatch counters; the coverage test runs LAST and compares what the benchmarked steps dispatch against the union
DISPATCHED = {}          # test node id -> set of launch sites it hit


@pytest.fixture(autouse=True)
def _record_dispatch(request):
    if "gpu" not in request.keywords or not _has_gpu():
        yield
        return
    from languagegroundedsemseg_amd import engine
    engine.dispatch_counts(reset=True)
    yield
    DISPATCHED[request.node.nodeid] = set(engine.dispatch_counts(reset=True))


def pytest_collection_modifyitems(config, items):
    last = [it for it in items if "test_gpu_dispatch_coverage" in it.nodeid]
    if last:
        items[:] = [it for it in items if "test_gpu_dispatch_coverage" not in it.nodeid] + last
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
