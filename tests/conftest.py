import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # torch's default of one CPU thread per logical CPU is slower than 16 threads on the GPU pool's 256-thread hosts (the oracle's
    # B = 2 step: 1.6 s at 16 threads, minutes at 256): cap the host-side thread pool of the test process
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), 16))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory vl-merging_amd/, imported as vl_merging_amd)."""
    import __graft_entry__ as ge
    return ge.import_package()
