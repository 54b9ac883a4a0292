import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


@pytest.fixture(scope='session')
def built():
    """Build the HIP library and the C oracle once per session (both are plain compiler calls)."""
    import __graft_entry__ as entry
    entry.build()
    return True


@pytest.fixture(scope='session')
def dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from boxinstseg_amd import _lib
    assert _lib.load().bxi_check_device(0) == 0, 'cuda:0 is not gfx950'
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _fresh_eval_state(request):
    """Every GPU test starts from the library's own choices: a test that used several streams leaves this host thread's evaluations
    flagged BXI_EVAL_SHARED_DEVICE (boxinstseg_amd/functional.py: sticky by design), and a test that forced a fault leaves the two-launch
    preference; neither may leak into the next test.  Workspaces are dropped, so the next evaluation allocates a zeroed one."""
    yield
    if 'gpu' in request.keywords:
        from boxinstseg_amd import functional as Fh
        Fh.reset_eval_state()
