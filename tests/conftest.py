"""Shared pytest setup: marker registration, import paths, one-time native builds."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_builds():
    """Build the engine and the oracle when their shared objects are missing."""
    lib = os.path.join(ROOT, "smvp-toolkit_amd", "lib", "libsmvp_amd.so")
    cli = os.path.join(ROOT, "smvp-toolkit_amd", "bin", "smvp-toolkit-cli")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "smvp-toolkit_amd")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    yield


PLAN_OPTIONS = ("csr_col16", "csr_rowrel", "binned_near", "binned_overlap", "tjds_index", "sharded_threads", "mm_threads")


@pytest.fixture(autouse=True)
def plan_options_back_to_default():
    """Plan options (smvp_set_option) are process-wide: whatever a test sets is taken back after it."""
    yield
    import smvp_toolkit_amd as sm

    for name in PLAN_OPTIONS:
        sm.set_option(name, None)


SAMPLES = ["ibm32.mtx", "curtis54.mtx", "pdp08-pg4.mtx", "memplus.mtx", "pwt.mtx"]

# committed reference reports: matrix -> (CSR report stamp, TJDS report stamp or None)
REPORTS = {
    "ibm32.mtx": ("1615284655", "1615284655"),
    "memplus.mtx": ("1615284663", "1615284665"),
    "pwt.mtx": ("1615284671", "1615284679"),
    "curtis54.mtx": ("1615284695", "1615284695"),
    "pdp08-pg4.mtx": ("1619162887", None),   # the reference crashes in TJDS on this one
}
