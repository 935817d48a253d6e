import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "steps_down: the test forces the resident kernel's self-healing step-down")


@pytest.fixture(scope="session")
def oracle():
    from oracle.rcn_oracle import COracle, build_oracle
    build_oracle()
    return COracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def no_silent_step_down(request):
    """A GPU test that passes because the library quietly stepped a context down from the resident one-XCD kernel to the two-kernel
    pipeline would test the wrong kernel: every context closed during a test must report zero step-downs, unless the test is about
    them (marker `steps_down`)."""
    if "gpu" not in request.keywords:
        yield
        return
    from mercer_research_amd import _lib
    before = _lib.FALLBACKS_SEEN
    yield
    if "steps_down" not in request.keywords:
        assert _lib.FALLBACKS_SEEN == before, "a context stepped down from the resident kernel during this test (rcn_hip_fallbacks_taken)"
