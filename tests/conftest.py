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
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "meta_json" in d:
        d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    if "images_u8" in d and "images" not in d:  # decoded 8-bit photographs: float32 u8 / 255 is exactly what the loader yields
        d["images"] = d["images_u8"].astype(np.float32) / np.float32(255.0)
    return d


@pytest.fixture(scope="session")
def golden():
    return load_golden
