import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import orc
    return orc.lib()


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from svin_amd import estimator
    return estimator.load_library()


@pytest.fixture
def debug_option():
    """set a process-wide debug / A-B option of libsvin_ba.so for one test (svin_ba_debug_set_option) and restore it afterwards;
    the library reads its environment switches once, so a monkeypatched environment variable would not be seen"""
    from svin_amd.estimator import Estimator
    saved = {}

    def setter(name, value=1):
        if name not in saved:
            saved[name] = Estimator.debug_get_option(name)
        Estimator.debug_set_option(name, value)
    yield setter
    for name, value in saved.items():
        Estimator.debug_set_option(name, value)
