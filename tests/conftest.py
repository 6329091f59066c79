"""pytest wiring: the `gpu` marker, import paths, and the golden-vector loader."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "dlmc-quant_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """tests/golden/golden_v1.{npz,json}: vectors produced by the reference's own code."""

    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        self.arr = np.load(os.path.join(d, "golden_v1.npz"))
        with open(os.path.join(d, "golden_v1.json")) as f:
            self.meta = json.load(f)
        self.cases = self.meta["cases"]

    def of_kind(self, kind):
        return [c for c in self.cases if c["kind"] == kind]

    def get(self, case, field):
        import torch
        return torch.from_numpy(np.array(self.arr[f"{case['name']}.{field}"]))

    def has(self, case, field):
        return f"{case['name']}.{field}" in self.arr.files


@pytest.fixture(scope="session")
def golden():
    return Golden()
