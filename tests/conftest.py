import os
import sys

import pytest

# the parity tests of the closed experiments (alternative kernel forms, solver variants: DESIGN 9.1) select them through their switches, which the
# library and the package read only under this master switch; a test that sets none of them runs the defaults, as every user does
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build(ref=True)
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
