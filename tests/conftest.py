import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def pairs(flat):
    return [tuple(flat[i:i + 2]) for i in range(0, len(flat), 2)]


def grid_from_bits(hexbits, shape):
    W, H = shape
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(hexbits), dtype=np.uint8))[:W * H]
    return bits.reshape(W, H).astype(np.uint8)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def map_grids():
    z = np.load(os.path.join(GOLDEN, "maps_png.npz"))
    return z
