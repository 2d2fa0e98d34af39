import os
import sys

import pytest

# The oracle is OpenMP code; the GPU box has 256 hardware threads.  Tests use small inputs: cap the team size so
# fork/join overhead does not dominate (bench.py's cpu_baseline leg uses all cores and is not affected).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_golden.json")) as f:
        return json.load(f)


def hex_f32(h):
    import numpy as np
    return np.frombuffer(bytes.fromhex(h), dtype="<f4").copy()
