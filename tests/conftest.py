import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg(sub: str = ""):
    """The product package directory is `metavoice-src_amd` (hyphenated), so it is imported by name."""
    return importlib.import_module("metavoice-src_amd" + (("." + sub) if sub else ""))


@pytest.fixture(scope="session")
def mv():
    return load_pkg()


@pytest.fixture(scope="session")
def synthetic():
    return load_pkg("synthetic")
