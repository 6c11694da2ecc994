import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the speaker-reference / embedding caches default to ~/.cache like the reference's; tests keep out of the home directory
if "METAVOICE_CACHE_DIR" not in os.environ:
    import tempfile

    os.environ["METAVOICE_CACHE_DIR"] = tempfile.mkdtemp(prefix="mv_test_cache_")


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
