import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The HIP library is git-ignored (built in-tree): build it if a fresh checkout has none, so the C-ABI
    export test and the GPU tests never run against a missing or silently absent extension."""
    lib = os.path.join(ROOT, "wavtokenizer_amd", "libwavtok_hip.so")
    lab = os.path.join(ROOT, "tools", "lib", "libwavtok_hip_lab.so")
    if not os.path.exists(lib) or not os.path.exists(lab):
        subprocess.run(["make", "-C", os.path.join(ROOT, "wavtokenizer_amd", "csrc"), "-j8", "all", "lab"], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line with the session's parity numbers (tests/parity_log.py) so that a -q log still carries them."""
    from tests import parity_log
    line = parity_log.dump()
    if line:
        terminalreporter.write_line(line)
