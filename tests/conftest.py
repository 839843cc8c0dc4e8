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
    """a fixture of tests/golden/, plain JSON or xz-compressed JSON (tests/golden/gio.py)"""
    from tests.golden import gio
    return gio.load(name)


def limbs(hexlist):
    return [int(v, 16) for v in hexlist]


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_binding import load_oracle
    return load_oracle()


@pytest.fixture(scope="session", autouse=True)
def generated_fields():
    """On a GPU box: the plug-ins of the generator mode's example moduli exist before any test binds a Field to one (a no-op when
    they travelled with the tree and are current; otherwise one hipcc unit each, about ten seconds).  Not done on CPU-only runs:
    the CPU suite tests generation in a scratch directory (tests/test_generate_cpu.py)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        from modarith_amd import generate as gen
        for arg, fam in gen.EXAMPLES:
            gen.generate(arg, family=fam)
        for c in gen.EXAMPLE_CURVES:
            gen.generate_curve(**c)
        for c in gen.EXAMPLE_LADDERS:
            gen.generate_ladder(**c)
    yield
