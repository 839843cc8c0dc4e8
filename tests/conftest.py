"""pytest configuration: `gpu` marker, oracle loader, golden-fixture loader.

The oracle (oracle/liboracle.so) is test infrastructure: it is built on demand here and used only
as the checker.  Nothing in this file reads /root/reference.
"""
import ctypes
import json
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


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def limbs(hexlist):
    return [int(v, 16) for v in hexlist]


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_binding import load_oracle
    return load_oracle()
