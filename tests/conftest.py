import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# ---- the clean launcher (tests/_launcher.py): started here, at session start, while this process has certainly not touched the GPU
_LAUNCHER = {"proc": None}


def pytest_sessionstart(session):
    import subprocess
    if "not gpu" in (session.config.getoption("-m") or ""):
        return                                     # the CPU suite starts children itself (nothing there initialises HIP)
    _LAUNCHER["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_launcher.py")], stdin=subprocess.PIPE,
                                         stdout=subprocess.PIPE, text=True, cwd=ROOT)


def pytest_sessionfinish(session, exitstatus):
    p = _LAUNCHER["proc"]
    if p is not None:
        try:
            p.stdin.close()
            p.wait(timeout=10)
        except Exception:
            p.kill()
        _LAUNCHER["proc"] = None


@pytest.fixture(scope="session")
def clean_launcher():
    """run(cmd, env=None, cwd=None, timeout=900) -> (returncode, stdout, stderr), executed by a helper process that was created before any
    GPU call of this session and never makes one itself -- independent of test order, -k filters or --lf."""
    import json
    p = _LAUNCHER["proc"]
    if p is None or p.poll() is not None:
        pytest.skip("no clean launcher process in this session (started only when GPU tests are selected)")

    def run(cmd, env=None, cwd=None, timeout=900):
        p.stdin.write(json.dumps({"cmd": cmd, "env": env, "cwd": cwd, "timeout": timeout}) + "\n")
        p.stdin.flush()
        rep = json.loads(p.stdout.readline())
        return rep["rc"], rep["stdout"], rep["stderr"]
    return run


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
