import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _default_wgrad_mode():
    """vmg_amd.train.TrainStep switches the weight-gradient mode to 'deferred' for the process; tests that follow one that built a
    TrainStep must see the default ('autograd': every parameter gradient flows through autograd) again."""
    yield
    mod = sys.modules.get("vmg_amd.functional")
    if mod is not None:
        mod.set_wgrad_mode("autograd")
