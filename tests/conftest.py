import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "auto_mode: test_gpu_select_modes.py — leave the choice of the select passes to the library")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build_library()
    return oracle
