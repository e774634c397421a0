import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """Default GPU context (gpu-marked tests only).  The HIP library must load and a
    device must be present: there is no CPU fallback to hide behind."""
    import bitnuc_amd
    from bitnuc_amd import build
    build.ensure_built()  # a fresh checkout has no .so (git-ignored): compile the product, never substitute it
    c = bitnuc_amd.Context(0)
    yield c
    c.close()
