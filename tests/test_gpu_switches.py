"""GPU tier: the A/B switches that select alternative kernels inside liblasr (DESIGN.md §4 "Switches") are branches of the shipped
library; the default suite only walks the defaults.  Each switch here runs the same bf16 training step + eval forward in a
subprocess (the switches are read once per process) and must agree with the default build of the step: same loss, same gradient
norms per layer group, same eval log-probs - to bf16-path tolerances (different kernels round differently)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {},                                             # defaults (the yardstick)
    {"LASR_DW_UNI": "0"}, {"LASR_DW_UNI": "64"},   # depthwise backward: two-kind grid / 64-channel unified form
    {"LASR_DWCONV_FMA": "1"}, {"LASR_DWCONV_DOT2": "1"}, {"LASR_DWWGRAD_VALU": "1"},   # VALU forms of the depthwise kernels
    {"LASR_DW_NO_FUSED_BWD": "1"},
    {"LASR_NO_FUSE": "1"}, {"LASR_NO_DEFER": "1"},  # unfused BN statistics / inline weight-gradient reductions
    {"LASR_BN_SLICED": "0"},                        # row-major BN backward pair
    {"LASR_WGRAD_SMALL_TILE": "1"}, {"LASR_NO_GEMM_BATCH": "1"}, {"LASR_GEMM_BIG_MIN_TILES": "100000"},   # 128 x 128 GEMM forms
    {"LASR_NO_MEL_CTC": "1"}, {"LASR_CTC_NO_LDS": "1"},   # separate lattice / feature launches; emissions through the register ring
    {"LASR_NO_EVAL_FOLD": "1"},
    {"LASR_BN_DW_FUSE": "0"},                       # BN + add + activation as its own launch instead of inside the next depthwise forward
    {"LASR_BN_DW_FUSE": "2"},                       # ... inside every stride-1 depthwise forward, full tiles too
    {"LASR_BN_APPLY_SPLIT": "1"},                   # BN backward apply pass on the statistics pass's chunks (one workgroup per CU)
]


def _run(env_extra, variant="plain", n_class=28, shape=()):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("LASR_") and k not in ("LASR_LIB_PATH",):
            del env[k]
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py"), variant, str(n_class)] + [str(a) for a in shape],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (env_extra, out.stderr[-2000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def _close(a, b, env):
    assert abs(a["loss"] - b["loss"]) <= 2e-3 * abs(b["loss"]), (env, a["loss"], b["loss"])
    for k, v in b["grad_norm"].items():
        assert abs(a["grad_norm"][k] - v) <= 5e-2 * v + 1e-6, (env, k, a["grad_norm"][k], v)
    assert abs(a["eval_checksum"] - b["eval_checksum"]) <= 5e-3 * abs(b["eval_checksum"]), (env, a["eval_checksum"], b["eval_checksum"])


def test_kernel_switches_agree_with_the_default_paths(dev):
    base = _run({})
    assert base["loss"] > 0 and all(v > 0 for v in base["grad_norm"].values())
    for env in SWITCHES[1:]:
        _close(_run(env), base, env)


def test_bn_inside_the_depthwise_forward_is_bit_identical(dev):
    """the BatchNorm + residual add + ReLU of a unit made in the staging loop of the next unit's depthwise forward (csrc/fused.h)
    performs bn_act_fwd_kernel's arithmetic operation for operation: loss, gradient norms and eval log-probs are the SAME numbers"""
    # T' = 501 (the BASELINE clip length), ragged utterances, B = 8: every stride-1 depthwise layer takes the HALF-tile kernel
    # (dwconv_s1_mfma_bn_kernel<*, 1>, the default path of the 256-channel layers at cfg2 / cfg5) - at the switch probe's default
    # T' = 201 dwconv_fwd_bn declines (T <= 256) and both runs would make the same unfused launches
    for act in ("relu", "swish"):
        shape = (8, 160000, 1, act)
        fused, unfused = _run({}, shape=shape), _run({"LASR_BN_DW_FUSE": "0"}, shape=shape)
        # the fused kernel really ran: 13 units (first_cnn ... block43) lose the BN-apply bracket of their own
        assert unfused["prof_brackets"]["bn"] - fused["prof_brackets"]["bn"] == 13, (fused["prof_brackets"], unfused["prof_brackets"])
        assert unfused["prof_brackets"]["dwconv"] == fused["prof_brackets"]["dwconv"]
        fused.pop("prof_brackets"); unfused.pop("prof_brackets")
        assert fused == unfused, (act, fused, unfused)
    # full 512-frame tiles too (LASR_BN_DW_FUSE=2, B = 32: the 512-channel layers leave the half-tile form), same bits
    shape = (32, 160000, 1, "relu")
    a, b = _run({"LASR_BN_DW_FUSE": "2"}, shape=shape), _run({"LASR_BN_DW_FUSE": "0"}, shape=shape)
    assert b["prof_brackets"]["bn"] - a["prof_brackets"]["bn"] == 13
    a.pop("prof_brackets"); b.pop("prof_brackets")
    assert a == b
    # and at the probe's own short shape nothing is fused (T' = 201): the two builds of the step make the same launches
    base = _run({})
    assert base == _run({"LASR_BN_DW_FUSE": "0"})
    # the dense head's tail (padded bf16 copy of d(logits), bias-gradient column sums, loss mean) in one launch or in three: same numbers
    sep = _run({"LASR_HEAD_TAIL_MERGED": "0"})
    sep.pop("prof_brackets"); base.pop("prof_brackets")
    assert sep == base, (sep, base)
    # the depthwise kernels' tap tables copied from the step's precomputed tables or built by every workgroup: the same bf16 pairs
    own = _run({"LASR_DW_TAPS": "0"})
    own.pop("prof_brackets")
    assert own == base, (own, base)
    # the decoder's split-K slabs summed inside the log_softmax launch or by the GEMM's own reduction: the same sums in the same order
    sep = _run({"LASR_LOGSOFTMAX_SPLIT": "0"})
    sep.pop("prof_brackets")
    assert sep == base, (sep, base)
    # round 5: the depthwise weight gradients' partial sums reduced by rider workgroups of the stage's weight-gradient launch (on the
    # CUs its tiles leave idle) or by lasr_reduce_many behind it: reduce_body.h either way - the same sums in the same order.  At the
    # BASELINE batch (32 x 10 s: 71 tiles x 3 slices leave 43 CUs) the riders really take them
    shape = (32, 160000, 1, "relu")
    rid, norid = _run({}, shape=shape), _run({"LASR_WGRAD_RIDERS": "0"}, shape=shape)
    rid.pop("prof_brackets"); norid.pop("prof_brackets")
    assert rid == norid, (rid, norid)


def test_large_vocabulary_head_switch(dev):
    """dense head (LASR_NO_LEAN_HEAD=1) against the lean head at C = 300"""
    base = _run({}, n_class=300)
    _close(_run({"LASR_NO_LEAN_HEAD": "1"}, n_class=300), base, "LASR_NO_LEAN_HEAD")


def test_context_se_switches(dev):
    base = _run({}, variant="context_se")
    for env in ({"LASR_SE_UNFUSED_BWD": "1"}, {"LASR_NO_FUSE": "1"}):
        _close(_run(env, variant="context_se"), base, env)
    # the BiLSTM backward recurrence inside the grid of the stage's weight-gradient launch (default) against the two launches one
    # after the other: the same kernels' arithmetic on the same operands - the same numbers
    # the SE squeeze's column sums inside the BN finalize launch (default) against the two launches: the same numbers
    two = _run({"LASR_SE_SEQSUM_IN_FINALIZE": "0"}, variant="context_se")
    assert two == base, (two, base)
    # round 5: the excite MLP inside the BN + SE + add + activation pass (one launch, every workgroup recomputing its utterance's hidden
    # vector) against lasr_se_fwd's two launches + the apply pass: the same arithmetic in the same order - the same numbers
    three = _run({"LASR_SE_FWD_FOLD": "0"}, variant="context_se")
    assert three["prof_brackets"]["bn"] - base["prof_brackets"]["bn"] == 15     # the fold really ran: 15 SE units lose a bracket each
    assert {k: v for k, v in three.items() if k != "prof_brackets"} == {k: v for k, v in base.items() if k != "prof_brackets"}
    sep = _run({"LASR_LSTM_BESIDE_WGRAD": "0"}, variant="context_se")
    _close(sep, base, "LASR_LSTM_BESIDE_WGRAD=0")
    # the recurrence's arithmetic is the same code (lstm_body.h): the forward is untouched and the loss identical; the gradients
    # differ in the last bits only - the split-K slice count of the launch the recurrences share leaves room for their workgroups,
    # and the separate form sums dg's columns (bias gradients) and casts it to bf16 in launches of their own
    assert sep["loss"] == base["loss"]
    for k, v in base["grad_norm"].items():
        assert abs(sep["grad_norm"][k] - v) <= 1e-5 * v, (k, sep["grad_norm"][k], v)
    # all of the stage's weight-gradient problems in the recurrences' launch, not only the tiles that end when the recurrences do
    allp = _run({"LASR_LSTM_WGRAD_BUDGET": "0"}, variant="context_se")
    _close(allp, base, "LASR_LSTM_WGRAD_BUDGET=0")
    assert allp["loss"] == base["loss"]


def test_roctx_ranges_change_nothing_but_are_live(dev):
    """LASR_ROCTX=1: the plan's roctx ranges (host-side markers for `rocprofv3 --marker-trace`) are live - the roctx library was found
    and bound - and the step's numbers are the numbers without them."""
    import ctypes
    base = _run({})
    traced = _run({"LASR_ROCTX": "1"})
    assert traced == base
    code = ("import os, sys; sys.path.insert(0, %r); from lightning_asr_amd import _lib; lib = _lib.load(); "
            "print(int(lib.lasr_roctx_enabled()), lib.lasr_roctx_range_push(b'lasr:test'), lib.lasr_roctx_range_pop())" % ROOT)
    on = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LASR_ROCTX="1"), capture_output=True, text=True, timeout=120)
    off = subprocess.run([sys.executable, "-c", code], env={k: v for k, v in os.environ.items() if k != "LASR_ROCTX"}, capture_output=True,
                         text=True, timeout=120)
    assert on.returncode == 0 and on.stdout.split() == ["1", "0", "0"], (on.stdout, on.stderr[-500:])
    assert off.returncode == 0 and off.stdout.split() == ["0", "0", "0"], (off.stdout, off.stderr[-500:])
