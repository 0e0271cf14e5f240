"""The table-driven fp64 log of the HaploCart column kernel (vgan_amd/csrc/log_tab.h), compiled for the host and checked
against logl; also that the committed table is what tools/gen_log_table.py generates."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_log_tab_matches_logl(tmp_path):
    exe = str(tmp_path / "log_tab_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "vgan_amd/csrc"),
                           os.path.join(ROOT, "tests/native/log_tab_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True, check=True).stdout.split()
    vals = dict(zip(out[0::2], map(float, out[1::2])))
    assert vals["worst_ulp"] < 4.0      # documented bound in log_tab.h
    assert vals["worst_abs"] < 4.5e-16  # error / max(1, |ln x|): what the per-read sums see
    assert vals["log1"] == 0.0          # a column that does not count contributes exactly nothing


def test_committed_table_is_reproducible(tmp_path):
    out = str(tmp_path / "log_table.h")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/gen_log_table.py"), out])
    assert open(out).read() == open(os.path.join(ROOT, "vgan_amd/csrc/log_table.h")).read()
