import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One fixture file written by oracle/make_golden.py from the REAL reference."""

    def __init__(self, path):
        self.path = path
        self.name = os.path.splitext(os.path.basename(path))[0]
        self.z = np.load(path)
        self.meta = json.loads(str(self.z["meta"]))

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z.files

    @property
    def T(self):
        return len(self.z["seed"])

    @property
    def variant(self):
        return self.meta["variant"]

    @property
    def max_it(self):
        """0 = unlimited (the fixtures were generated with 1000000, never reached)."""
        return 0 if self.meta["max_it"] >= 1000000 else self.meta["max_it"]

    def rows_of(self, t):
        off = self.z["rows_off"]
        return self.z["rows"][off[t]:off[t + 1]]


def golden_names(prefixes=None, variants=None, whole_run=False, uncapped=False):
    """Fixture names; `uncapped` keeps only those generated without a binding iteration cap (the fixpoint kernels'
    domain), so that no test has to skip at run time."""
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.splitext(os.path.basename(path))[0]
        is_whole = name.endswith("_wholerun")
        if is_whole != whole_run:
            continue
        if prefixes and not any(name.startswith(p) for p in prefixes):
            continue
        if variants and name.split("_")[1] not in variants:
            continue
        if uncapped and Golden(path).max_it:
            continue
        out.append(name)
    return out


def load_golden(name):
    return Golden(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(with_reference=False)
    return O


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a HIP device; there is no CPU fallback")
