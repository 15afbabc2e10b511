import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))

        @property
        def meta(self):
            with open(os.path.join(GOLDEN, "meta.json")) as f:
                return json.load(f)
    return G()


@pytest.fixture(scope="session")
def parity_log():
    """Collects the numbers the parity tests measure (max |dp|, mismatch rates, ...) and writes them to
    gpurun_out/parity_report.json at the end of the session (`pytest -q` drops prints; tools/refresh_profiles.sh
    copies the file to profiles/)."""
    import json
    rec = {}

    def log(name, **values):
        rec.setdefault(name, {}).update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in values.items()})
    yield log
    if rec:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_report.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(rec)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
