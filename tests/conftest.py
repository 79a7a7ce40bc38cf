import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_ffi import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle_ffi import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libdbde_ref.so not built (needs /root/reference at build time)")
    return Reference()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        manifest = json.load(f)
    arrays = dict(np.load(os.path.join(HERE, "golden", "frames.npz")))
    return manifest, arrays
