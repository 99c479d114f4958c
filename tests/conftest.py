import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    # small world banks are the rule in tests: the "no more worlds than auto-resetting envs" hint is noise there
    for it in items:
        it.add_marker(pytest.mark.filterwarnings("ignore:BatchedAuvEnv.*auto-resetting envs"))
