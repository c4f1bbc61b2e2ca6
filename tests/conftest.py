import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_window(g):
    """Rebuild the `window` argument a CAF golden was generated with."""
    spec = str(g["wspec"])
    if spec == "none":
        return None
    if spec == "array":
        return g["window"]
    return ("kaiser", float(g["window"][0]))


def rel_err(a, b):
    """max|a-b| / max|b| -- the peak-normalised error of the north-star parity statement."""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.abs(a - b).max() / np.abs(b).max())


CAF_SMALL = ["p2", "p2_kaiser_arr", "p2_kaiser_tuple", "oddq", "oddq_p1", "nondiv", "small",
             "bigq", "padded", "longfilt", "srv128", "lags_gt_q"]


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build product (git-ignored): build it once if it is missing."""
    from passiveradar_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def gpu_ready():
    from passiveradar_amd import _lib
    _lib.require_gpu()
    return True
