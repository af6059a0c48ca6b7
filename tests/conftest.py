import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def test_bmp():
    """Test.bmp (the reference's round-trip corpus), recovered by decoding the committed
    Test.lz fixture with the oracle; its sha256 is pinned to the reference's file."""
    import hashlib
    import oracle_lib as O
    data = open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()
    out, st = O.container_decompress(O.A.C_LZSS, data, lz=O.A.LzProperties.from_bits(10, 6, 2))
    assert st == 0
    assert hashlib.sha256(out).hexdigest() == "5c8809e6059937c47544839bfa9f8d70a878574a0442c903e8353b2456757ccd"
    return out
