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


def e2e_gate(name: str, floor: float = 6e-3, default: float = 2e-2) -> float:
    """Gate of an end-to-end f32 gradient check = 2 x the worst error measured for it on the MI355X (profiles/r03_e2e_measured.json,
    written by record_measured), but not below the one-flip level: ONE activation of this 4-utterance BN stack landing on the other
    side of zero moves the worst tensor by a few 1e-3 (profiles/r02_golden_f32_threads.txt: the CPU oracle itself moved by 6.8e-3
    between thread counts), so a kernel change that reorders a sum may legitimately move a tiny measured value up to that level."""
    import json
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "r03_e2e_measured.json")))
    except Exception:
        return default
    return max(2.0 * m[name], floor) if name in m else default


# ONE tolerance for the log-mel front-end (north_star: "mel features within 1e-4 relative"): the HIP kernel against the oracle
# evaluated in f64, relative to the feature scale.  Measured worst on the MI355X: 2.9e-5 (profiles/r03_e2e_measured.json, mel_*).
# Against the oracle's f32 evaluation (what the reference's torch.stft computes in) the distance is the f32 oracle's OWN round-off in
# the weak bins of high-dynamic-range frames (up to 5e-4 on a pure tone), so that comparison is allowed MEL_TOL + that spread.
MEL_TOL = 1e-4
