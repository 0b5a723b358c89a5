import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gl-radix-sort_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Makes sure libglu_hip.so, the C++ test programs and the oracle exist (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as entry
    import glu_hip

    cpp_bin = os.path.join(ROOT, "tests", "cpp", "bin", "test_radix_sort_api")
    if not os.path.exists(glu_hip.LIB_PATH) or not os.path.exists(cpp_bin):
        entry.build()
    return glu_hip


@pytest.fixture(scope="session")
def golden():
    import json

    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "reference_vectors.json")) as f:
        ref = json.load(f)
    with open(os.path.join(gdir, "oracle_checksums.json")) as f:
        sums = json.load(f)
    return {"reference": ref, "checksums": sums}


def fnv1a64(a):
    import numpy as np

    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a, dtype=np.uint32).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h
