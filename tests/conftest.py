import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/a3_oracle.c through ctypes."""
    from oracle import a3oracle

    a3oracle.build()
    a3oracle.lib()
    return a3oracle


@pytest.fixture(scope="session")
def dicts():
    from aruco3_amd.dictionaries import ARDictionary

    return ARDictionary
