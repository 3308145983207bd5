import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _rlcf_test_stream(request):
    """Stream-hygiene mode of the GPU suite.  RLCF_TEST_STREAM=nonblocking runs every `-m gpu` test — every engine call, every op-level
    call, every torch expression the test compares with — on a NON-BLOCKING side stream (torch's pool streams are created with
    hipStreamNonBlocking) instead of the NULL stream the default run uses: nothing then orders the test's queue against NULL-stream
    work, so an engine path that still fills or copies on the NULL stream (a first-call workspace fill once did) shows up as a wrong
    result.  Engines are built per test (make_engine & friends), so first-call allocation paths run under it too.  The default
    (unset / "null") leaves torch's current stream alone.  profiles/r6_pytest_gpu_nonblocking_tail.txt keeps the run of the round."""
    mode = os.environ.get("RLCF_TEST_STREAM", "null")
    if mode not in ("null", "nonblocking"):
        raise pytest.UsageError(f"RLCF_TEST_STREAM={mode!r}: 'null' or 'nonblocking'")
    if mode == "null" or request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.default_stream())        # module-scoped fixtures may have produced tensors on the default stream
    with torch.cuda.stream(side):
        yield
    torch.cuda.default_stream().wait_stream(side)
    torch.cuda.synchronize()
