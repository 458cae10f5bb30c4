"""INTEGRATION.md's edit of the reference's main() as a program: examples/reference_patch_example.c declares the
reference's own types (MMRawData, struct _time_data_) and calls libsmvp_amd.so exactly as the document shows."""
import os
import subprocess

import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import ROOT


def build(tmp_path):
    exe = str(tmp_path / "patch_example")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "examples", "reference_patch_example.c"),
           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(sm.LIB_PATH), "-lsmvp_amd",
           "-Wl,-rpath," + os.path.dirname(sm.LIB_PATH), "-lm", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_example_compiles_and_links_against_the_abi(tmp_path):
    """CPU: the layout static-asserts hold, every call matches the header, the program links; without a GPU it stops
    with the engine's error (no CPU fallback)."""
    exe = build(tmp_path)
    p = subprocess.run([exe, ob.fixture_path("ibm32.mtx"), "3", str(tmp_path)], capture_output=True, text=True)
    if sm.device_count() == 0:
        assert p.returncode == 1 and "[ERROR]" in p.stdout and "no HIP device" in p.stdout


@pytest.mark.gpu
def test_example_reproduces_the_committed_reports(tmp_path):
    exe = build(tmp_path)
    p = subprocess.run([exe, ob.fixture_path("ibm32.mtx"), "50", str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))
    reports = sorted(f for f in os.listdir(tmp_path) if f.startswith("smvp-toolbox_report_"))
    assert len(reports) == 2
    for f in reports:
        assert ob.report_y_lines(open(tmp_path / f).read()) == want       # ibm32: CSR and the corrected TJDS agree
