import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C oracle is test infrastructure; (re)build it once per session (gcc only, no GPU)."""
    from oracle import oracle
    oracle.build()


def pytest_terminal_summary(terminalreporter):
    """OccupancyGrid(as_image=True) engine-vs-oracle comparisons allow one uint8 step where ((v + 1) / 2) * 255 is an integer
    up to the last bit of the libm in use (golden_util.assert_obs_close): say how many cells that was (per xdist worker)."""
    from tests import golden_util
    off, total = golden_util.IMAGE_CELLS
    if total:
        terminalreporter.write_line(f"as_image cells off by one uint8 step (engine vs oracle): {off} of {total}")
