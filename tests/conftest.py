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
        self.arrs = [np.load(os.path.join(d, "golden_v1.npz"))]
        with open(os.path.join(d, "golden_v1.json")) as f:
            self.meta = json.load(f)
        self.cases = list(self.meta["cases"])
        # cases added after golden_v1 was frozen (make_golden.py --supplement): same generator, own pair of files
        for extra in ("golden_v1_grad", "golden_v2"):      # --supplement (round 2), --supplement2 (round 4: LSQ init, minmax_pixel)
            if os.path.exists(os.path.join(d, extra + ".npz")):
                self.arrs.append(np.load(os.path.join(d, extra + ".npz")))
                with open(os.path.join(d, extra + ".json")) as f:
                    self.cases += json.load(f)["cases"]
        self.arr = self.arrs[0]

    def of_kind(self, kind):
        return [c for c in self.cases if c["kind"] == kind]

    def get(self, case, field):
        import torch
        key = f"{case['name']}.{field}"
        return torch.from_numpy(np.array(next(a for a in self.arrs if key in a.files)[key]))

    def has(self, case, field):
        return any(f"{case['name']}.{field}" in a.files for a in self.arrs)


@pytest.fixture(scope="session")
def golden():
    return Golden()
