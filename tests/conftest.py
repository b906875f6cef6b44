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


E2E_CAPS = {
    # fixed upper bounds of the end-to-end f32 gradient checks (worst relative L2 over the parameter tensors vs the f64 oracle): the
    # tolerances these tests carried before the gates were derived from measurements.  A gate never exceeds its cap, so re-recording
    # the measured json after a regression cannot loosen a test silently.
    "plain_golden_f32_grad_rel_l2_vs_f64_oracle": 1e-2,
    "context_golden_f32_grad_rel_l2_vs_f64_oracle": 1e-2,
    "context_se_golden_f32_grad_rel_l2_vs_f64_oracle": 1e-2,
    "host_training_step_f32_grad_rel_l2_vs_f64_oracle": 1e-2,
    # the 2-utterance T' = 33 SE edge shape is the one case whose measured value (1.99e-2, profiles/r03_e2e_measured.json) sits at the
    # old fixed 2e-2: its BatchNorm statistics run over 66 rows of which 13 are padding, one ReLU flip there is ~1e-2 of the worst
    # tensor.  Its cap is deliberately 3e-2 (1.5 x measured), stated here instead of hidden in a 2 x rule.
    "edge_context_se_B2_T65_f32_grad_rel_l2_vs_f64_oracle": 3e-2,
}


def e2e_gate(name: str, floor: float = 6e-3, cap: float = 2e-2) -> float:
    """Gate of an end-to-end f32 gradient check = min(cap, max(2 x the worst error measured for it on the MI355X, floor)).
    measured: profiles/r03_e2e_measured.json (written by record_measured in an earlier round; the file is never re-recorded by a
    test run).  floor = the one-flip level: ONE activation of this 4-utterance BN stack landing on the other side of zero moves the
    worst tensor by a few 1e-3 (profiles/r02_golden_f32_threads.txt: the CPU oracle itself moved by 6.8e-3 between thread counts).
    cap = E2E_CAPS[name] (default 2e-2, the old fixed tolerance of the edge-shape tests); a name without a measurement gets the cap."""
    import json
    cap = E2E_CAPS.get(name, cap)
    m = json.load(open(os.path.join(ROOT, "profiles", "r03_e2e_measured.json")))
    if name not in m:
        return cap
    return min(cap, max(2.0 * m[name], floor))


# ONE tolerance for the log-mel front-end (north_star: "mel features within 1e-4 relative"): the HIP kernel against the oracle
# evaluated in f64, relative to the feature scale.  Measured worst on the MI355X: 2.9e-5 (profiles/r03_e2e_measured.json, mel_*).
# Against the oracle's f32 evaluation (what the reference's torch.stft computes in) the distance is the f32 oracle's OWN round-off in
# the weak bins of high-dynamic-range frames (up to 5e-4 on a pure tone), so that comparison is allowed MEL_TOL + that spread.
MEL_TOL = 1e-4
