import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def record_measured(name: str, value: float) -> None:
    """measured worst-case errors of the end-to-end parity tests -> gpurun_out/e2e_measured.json (copied to profiles/ per round;
    the gates in the tests are set at 2x these)"""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "e2e_measured.json")
    try:
        cur = json.load(open(p))
    except Exception:
        cur = {}
    cur[name] = float(value)
    with open(p, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
