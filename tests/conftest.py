import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: regenerates config-dim (2.6 GB) weights on the CPU")
    # The CPU oracle is plain torch: on a box that shows hundreds of logical CPUs under a container quota, torch's default intra-op
    # thread count (all of them) makes its GEMMs and copies many times SLOWER (measured: the GPU suite took 6 min with the default and
    # under 3 with 16 threads).  Bound it once for the whole session (the build container has 8: nothing changes for the bit-exact fixture tests there).
    try:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
