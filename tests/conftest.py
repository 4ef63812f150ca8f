import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Builds libplayrender.so if it is missing (hipcc cross-compiles without a GPU)."""
    from playableenvironments_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """GPU tests start on NaN-poisoned allocator blocks (tests/helpers.poison_device_memory): the renderer's workspaces are
    ``torch.empty``, and scratch that a kernel reads without anybody having written it must not pass by luck on fresh pages."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from tests.helpers import poison_device_memory
            poison_device_memory()
    yield


@pytest.fixture(autouse=True)
def _settlement_log(request):
    """Every forward-field excess the gradient harness SETTLED (float64 arbitration / divergence kink) during a test goes to the
    file named by PR_SETTLEMENT_LOG with the test's id - how profiles/r05_settlements.log was recorded."""
    yield
    path = os.environ.get("PR_SETTLEMENT_LOG")
    module = sys.modules.get("tests.test_gpu")
    if module is None or not getattr(module, "SETTLEMENTS", None):
        return
    entries = module.drain_settlements()
    if path:
        with open(path, "a") as f:
            for e in entries:
                f.write(f"{request.node.nodeid} {e}\n")
