"""The table behind the PRS group walk (csrc/alz_prs_table.h: one entry per (entry state, flag byte)) checked on the host: a C++
program parses random streams token by token, as Sega/PRS.cs:59-102 reads them, and group by group through the table -- the scalar
walk's word and the token lanes' word -- and compares every token and every flag-byte position (tests/prs_table_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_table_reproduces_the_token_by_token_parse(tmp_path):
    exe = str(tmp_path / "prs_table_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "auroralib", "compression_amd", "csrc"),
                           os.path.join(ROOT, "tests", "prs_table_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.strip() == "ok", out
