import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Past four of the caller's handles alive on a device the library hands the one-shot calls of PRIVATE handles to the device's
# shared lanes (phmm_host::route_shared).  Tests hold their handles to what THEY do (per-handle statistics, which stream a call
# ran on, which way a region call went), and a session creates dozens of engines: routing is off for engines made under this
# process's environment, and the tests of the routing itself (tests/test_submit_wait.py) and the C++ callers
# (tools/threads_bench children) switch it back on for theirs.
os.environ.setdefault("PHMM_ROUTE_SHARED", "0")


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
