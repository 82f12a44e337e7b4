import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests fill every weight synthetically (oracle/fill.py): the offline fallback of BertWrapper is wanted here
os.environ.setdefault("PTPP_ALLOW_RANDOM_BERT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k]) for k in d.files}


def key_shapes(arr):
    """decode the 'name|d0,d1' strings stored by gen_golden._keys"""
    out = []
    for s in arr.tolist():
        n, sh = s.split("|")
        out.append((n, tuple(int(v) for v in sh.split(",")) if sh else ()))
    return out


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
