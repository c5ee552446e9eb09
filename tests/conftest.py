import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_lib():
    import dto_amd
    return dto_amd.load_library()


@pytest.fixture(autouse=True, scope="session")
def _host_xfer_self_check():
    """Every handle the tests create checks its host-pointer Jacobian / Hessian against the whole device slab (option
    host_xfer_check): a kernel that wrote outside the hand-off plan fails the call instead of being dropped silently."""
    import dto_amd
    dto_amd.Evaluator.default_options = {"host_xfer_check": 1}
    yield
    dto_amd.Evaluator.default_options = {}
