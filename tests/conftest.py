import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip silently) on a GPU box when the extension is
    # missing; on a CPU-only box they are deselected by `-m "not gpu"`.
    pass


@pytest.fixture(scope="session")
def ext():
    import windflow_amd

    return windflow_amd.load_ext()
