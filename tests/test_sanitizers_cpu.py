"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): the oracle against every golden / known-answer
suite, and the host-side header parsers of csrc/alz_container.cpp against the committed container files and seeded mutations of them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_env():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so beside gcc")
    env = dict(os.environ)
    env["LD_PRELOAD"] = libasan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=77"      # (CPython itself is not leak-clean)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    return env


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)


def test_oracle_suites_under_asan_ubsan():
    """tests/test_oracle_golden.py, test_oracle_semantics.py and test_kat.py in a child interpreter that loads oracle/liboracle_asan.so."""
    _build()
    env = _san_env()
    env["ALZ_ORACLE_SO"] = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_semantics.py"),
                        os.path.join(ROOT, "tests", "test_kat.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT, timeout=1500)
    out = p.stdout
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0 and " passed" in out, out[-4000:]
    # the child really ran on the sanitized library
    q = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tests'); import oracle_lib as O; print(O.lib._name)"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT, timeout=120)
    assert q.stdout.strip().endswith("liboracle_asan.so"), q.stdout


def test_container_header_parsers_fuzz_under_asan_ubsan():
    """alz_container_is_match / alz_container_decompressed_size of every container on exact-size heap buffers: the committed container
    files, every prefix of them and 2 000 seeded mutations of each (tests/fuzz_container.cpp)."""
    _build()
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_containers.json")))["cases"]
    seeds = "\n".join(c["file"] for c in cases) + "\n"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:exitcode=77", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([os.path.join(ROOT, "oracle", "fuzz_container"), "2000"], input=seeds, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=env, timeout=600)
    assert p.returncode == 0 and "no report" in p.stdout and "runtime error" not in p.stdout, p.stdout[-4000:]
    assert "%d seeds" % len(cases) in p.stdout
