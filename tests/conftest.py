import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def kat_rows():
    from oracle import oracle
    return oracle.load_kat(os.path.join(GOLDEN, "pairhmm-testdata.txt"))


@pytest.fixture(scope="session")
def hip_engine():
    """The HIP engine.  No fallback: on a box without a GPU this raises, it never degrades."""
    from lorikeet_amd import HipPairHMMEngine
    eng = HipPairHMMEngine(0)
    yield eng
    eng.close()
