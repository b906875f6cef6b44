"""Whole-model parity of the native plan (HIP, via the C ABI) against the golden fixtures captured
from the reference and against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.make_golden import golden_inputs, checksum

pytestmark = pytest.mark.gpu


# End-to-end f32 gradient gates (worst relative L2 over the parameter tensors, GPU f32 path vs the f64 oracle): conftest.e2e_gate =
# 2 x the worst measured on the MI355X (profiles/r03_e2e_measured.json), floored at the one-ReLU-flip level (6e-3).  The GPU result is
# bit-reproducible and the f64 oracle does not move with the host's thread count; the tight bounds (2e-5 per unit) are test_gpu_units.py.

def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def argmax_report(tag, am_gpu, lp_gpu, gold, noise=None):
    """Bit-exactness of the token-id argmax against the reference fixture, with the numbers SURVEY 8d asks for: the count of
    mismatching frames, the reference's own top-1 / top-2 log-prob margin at each of them, and the smallest margin over all frames.
    A mismatch is only acceptable where the reference's margin is below twice `noise` (the measured f32 noise of the CPU oracle
    against the same fixture: two classes whose log-probs are closer than what one f32 evaluation order moves them by have no
    defined winner).  Written to gpurun_out/argmax_margins.json (copied to profiles/ per round)."""
    import json
    import os
    ref_lp = torch.from_numpy(gold["logprobs"]).double()
    top2 = ref_lp.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])                       # (B, T')
    t_len = torch.from_numpy(gold["t_lengths"]).long()
    valid = torch.arange(margin.shape[1]).unsqueeze(0) < t_len.unsqueeze(1)
    ref_am = torch.from_numpy(gold["argmax"].astype("int64"))
    mis = (am_gpu.cpu().long() != ref_am)
    rec = {"frames": int(mis.numel()), "valid_frames": int(valid.sum()), "mismatches": int(mis.sum()), "mismatches_in_valid_frames": int((mis & valid).sum()),
           "min_margin_all_frames": float(margin.min()), "min_margin_valid_frames": float(margin[valid].min()),
           "margins_at_mismatches": [float(v) for v in margin[mis]],
           "gpu_logp_max_abs_err": float((lp_gpu.cpu().double() - ref_lp).abs().max()), "oracle_f32_noise": noise}
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/argmax_margins.json"
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    cur[tag] = rec
    with open(path, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
    return rec


def _native(variant, n_class, dev, dtype=torch.float32, act="relu"):
    from lightning_asr_amd.engine import NativeModel
    m = NativeModel(variant, n_class, mask=True, act=act, dtype=dtype, device=dev)
    m.load_state_dict(R.formula_state(variant, n_class))
    return m


def test_state_dict_layout_matches_reference(dev):
    m = _native("plain", 28, dev)
    assert [(t.name, t.shape) for t in m.tensors] == [(k, tuple(s)) for k, s in R.state_shapes("plain", 28)]
    assert m.n_param == 5044572          # SURVEY §6 [probe]
    assert len(m.tensors) == 184


def test_plain_forward_matches_golden_f32(dev):
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_plain.npz")
    x, tg, pct, tsz = golden_inputs()
    m = _native("plain", 28, dev)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    # eval mode first (running stats are the formula ones: mean 0, var 1)
    lp_e, _ = m.forward(feats, pct.to(dev), training=False)
    assert np.abs(lp_e.cpu().numpy() - gold["eval_logprobs"]).max() < 2e-4
    lp, am = m.forward(feats, pct.to(dev), training=True)
    assert m.tap("lens").cpu().tolist() == gold["t_lengths"].tolist()
    err = np.abs(lp.cpu().numpy() - gold["logprobs"]).max()
    assert err < 2e-4, err
    # bit-exact token-id argmax decode vs the reference CPU path
    assert np.array_equal(am.cpu().numpy().astype(np.int16), gold["argmax"])
    rec = argmax_report("plain_golden_f32", am, lp, gold)
    assert rec["mismatches"] == 0 and rec["min_margin_all_frames"] > 2 * rec["gpu_logp_max_abs_err"]     # no frame is even close to a tie
    for name in ["first_cnn", "block1", "block23", "block3", "block43", "block5", "last_cnn2"]:
        got = checksum(m.tap(name).transpose(1, 2).contiguous().cpu())
        assert np.abs(got - gold["tap_" + name]).max() < 1e-4, name


def test_plain_loss_backward_matches_golden_f32(dev):
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_plain.npz")
    x, tg, pct, tsz = golden_inputs()
    m = _native("plain", 28, dev)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert np.abs(nll.cpu().numpy() - gold["nll"]).max() / np.abs(gold["nll"]).max() < 1e-4
    assert abs(loss.item() - gold["losses"][0]) / gold["losses"][0] < 1e-4
    norms = np.array([m.view(t, m.grads).norm().item() for t in m.param_infos()])
    assert np.abs(norms / gold["grad_norms"] - 1).max() < 5e-3
    # full gradients against the oracle run in f64.  The f32 oracle is NOT a stable yardstick for this check: through the
    # 24-block BN stack one activation landing on the other side of zero moves every upstream gradient by a few 1e-3, and
    # the CPU's own f32 result does that as a function of torch's thread count (profiles/r02_golden_f32_threads.txt:
    # same host, 128 vs <=32 threads, 6.8e-3 apart; f32 vs f64 oracle in the dev container 3.2e-3).  So the end-to-end
    # bound is that noise level; the tight f32 bound (2e-5 per unit, teacher-forced) is tests/test_gpu_units.py.
    from oracle import ref_bf16 as E
    o = E.Bf16OracleModel("plain", 28, mask=True, state=R.formula_state("plain", 28), dtype=torch.float64, emulate=False)
    _, _, _, grads = E.loss_and_grads(o, x, tg, pct, tsz)
    rels = {t.name: rel_l2(m.view(t, m.grads), g) for t, g in zip(m.param_infos(), grads)}
    worst = max(rels.values())
    print("worst grad rel-L2 vs f64 oracle %.3e" % worst)
    from conftest import e2e_gate, record_measured
    record_measured("plain_golden_f32_grad_rel_l2_vs_f64_oracle", worst)
    assert worst < e2e_gate("plain_golden_f32_grad_rel_l2_vs_f64_oracle"), sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    # running statistics after one training forward
    for t in m.tensors:
        if t.kind == 1:
            assert rel_l2(m.view(t), o.state[t.name]) < 1e-4, t.name


def test_generic_backward_equals_fused(dev):
    from lightning_asr_amd import ops
    x, tg, pct, tsz = golden_inputs()
    m = _native("plain", 28, dev)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    loss, nll, lp, _ = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    g_fused = m.grads.clone()
    m2 = _native("plain", 28, dev)
    lp2, _ = m2.forward(feats, pct.to(dev), training=True)
    lens = ops.mask_lengths(pct.to(dev), lp2.shape[1])
    _, g_lp = ops.ctc_loss(lp2, tg.to(dev), lens, tsz.to(dev), 27)
    m2.backward(g_lp)
    assert rel_l2(m2.grads, g_fused) < 1e-4      # log_softmax backward sees sum_c(grad) ~ 1e-7, not exactly 0


def test_aishell_vocab_large_C(dev):
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_plain_aishell.npz")
    V = 4333
    x, tg, pct, tsz = golden_inputs(B=2, T_in=81, S=6, V=V)
    pct = torch.tensor([1.0, 0.75])
    m = _native("plain", V + 1, dev)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert np.abs(nll.cpu().numpy() - gold["nll"]).max() / np.abs(gold["nll"]).max() < 1e-4
    assert np.abs(checksum(lp.cpu()) - gold["logprob_checksum"]).max() < 2e-4
    assert np.array_equal(am.cpu().numpy().astype(np.int16), gold["argmax"])
    norms = np.array([m.view(t, m.grads).norm().item() for t in m.param_infos()])
    assert np.abs(norms / gold["grad_norms"] - 1).max() < 5e-3


def test_plain_bf16_mode_tracks_f32(dev):
    """bf16 activation mode (bench dtype): same plan on the bf16-MFMA kernels.  Not a parity mode:
    reported against the f32 golden with its own tolerance."""
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_plain.npz")
    x, tg, pct, tsz = golden_inputs()
    m = _native("plain", 28, dev, torch.bfloat16)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev), torch.bfloat16)
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert abs(loss.item() - gold["losses"][0]) / gold["losses"][0] < 2e-2
    assert np.abs(lp.cpu().numpy() - gold["logprobs"]).max() < 0.25
    agree = (am.cpu().numpy().astype(np.int16) == gold["argmax"]).mean()
    assert agree > 0.9, agree
    norms = np.array([m.view(t, m.grads).norm().item() for t in m.param_infos()])
    assert np.median(np.abs(norms / gold["grad_norms"] - 1)) < 5e-2
    assert torch.isfinite(m.grads).all()


@pytest.mark.parametrize("variant", ["context", "context_se"])
def test_context_variants_match_golden_f32(dev, variant):
    """QuartNetContext (BiLSTM context branch, 336-ch block3, block6) and QuartNetContextSE (+SE) on the
    native plan vs the fixtures captured from the reference, and full gradients vs the CPU oracle."""
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_%s.npz" % variant)
    x, tg, pct, tsz = golden_inputs()
    m = _native(variant, 28, dev)
    assert [(t.name, t.shape) for t in m.tensors] == [(k, tuple(s)) for k, s in R.state_shapes(variant, 28)]
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    lp_e, _ = m.forward(feats, pct.to(dev), training=False)
    assert np.abs(lp_e.cpu().numpy() - gold["eval_logprobs"]).max() < 3e-4
    m2 = _native(variant, 28, dev)
    loss, nll, lp, am = m2.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert np.abs(lp.cpu().numpy() - gold["logprobs"]).max() < 3e-3       # oracle itself is 2.4e-3 from the reference here
    assert np.abs(nll.cpu().numpy() - gold["nll"]).max() / np.abs(gold["nll"]).max() < 1e-4
    # argmax: bit-exact, or different ONLY at frames where the reference's own top-1 / top-2 margin is inside the f32 noise of this
    # model's CPU evaluation - measured right here as the distance between the pinned f32 oracle (explicit LSTM loop) and the
    # reference fixture (ATen's LSTM) on the same weights and inputs (`oracle itself is 2.4e-3 from the reference`)
    om = R.OracleModel(variant, 28, mask=True, state=R.formula_state(variant, 28))
    om.training = True
    with torch.no_grad():
        noise = float((om(x, pct).double() - torch.from_numpy(gold["logprobs"]).double()).abs().max())
    rec = argmax_report("%s_golden_f32" % variant, am, lp, gold, noise=noise)
    # measured on the MI355X (profiles/r04_argmax_margins.json): 0 mismatching frames of 404 for both variants (the gate was
    # "> 99.5 % agreement" before); smallest reference margin 2.1e-4 / 4.8e-4 against a GPU log-prob error of 1.5e-5: bit-exact
    assert rec["mismatches"] == 0, rec
    assert rec["min_margin_all_frames"] > 2 * rec["gpu_logp_max_abs_err"], rec
    ctx = m2.tap("ctx_in")[:, :, 256:].contiguous().cpu()            # the reference module returns (B, T, 80)
    assert np.abs(checksum(ctx) - gold["tap_context_rnn"]).max() < 1e-5
    norms = np.array([m2.view(t, m2.grads).norm().item() for t in m2.param_infos()])
    assert np.abs(norms / gold["grad_norms"] - 1).max() < 2e-2
    # gradients against the f64 oracle (a stable yardstick: the f32 oracle's own result moves with torch's thread count)
    from oracle import ref_bf16 as E
    o = E.Bf16OracleModel(variant, 28, mask=True, state=R.formula_state(variant, 28), dtype=torch.float64, emulate=False)
    _, _, _, grads = E.loss_and_grads(o, x.double(), tg, pct, tsz)
    rels = {t.name: rel_l2(m2.view(t, m2.grads), g) for t, g in zip(m2.param_infos(), grads)}
    worst = max(rels.values())
    from conftest import e2e_gate, record_measured
    record_measured("%s_golden_f32_grad_rel_l2_vs_f64_oracle" % variant, worst)
    assert worst < e2e_gate("%s_golden_f32_grad_rel_l2_vs_f64_oracle" % variant), sorted(rels.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize("variant", ["plain", "context_se"])
def test_swish_whole_plan_matches_golden_f32(dev, variant):
    """model.act = swish (north_star "BatchNorm + Swish"; activate_fun/Swish.py:9-10) through the WHOLE native plan - every fused
    epilogue, the SE units, the BN backward that rebuilds the pre-activation - against the fixture captured from the reference model
    with its own Swish module in the epilogues, and full gradients against the f64 oracle.  Swish has no kink: no activation can
    land on the other side of zero, so the end-to-end gradient bound is far below the ReLU models' one-flip level."""
    from lightning_asr_amd import ops
    from oracle import ref_bf16 as E
    gold = np.load("tests/golden/model_%s_swish.npz" % variant)
    x, tg, pct, tsz = golden_inputs()
    m = _native(variant, 28, dev, act="swish")
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    lp_e, _ = m.forward(feats, pct.to(dev), training=False)
    assert np.abs(lp_e.cpu().numpy() - gold["eval_logprobs"]).max() < 3e-4
    m2 = _native(variant, 28, dev, act="swish")
    loss, nll, lp, am = m2.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert np.abs(lp.cpu().numpy() - gold["logprobs"]).max() < 3e-4
    assert np.abs(nll.cpu().numpy() - gold["nll"]).max() / np.abs(gold["nll"]).max() < 1e-4
    assert abs(loss.item() - gold["losses"][0]) / gold["losses"][0] < 1e-4
    mism = int((am.cpu().numpy().astype(np.int16) != gold["argmax"]).sum())
    assert mism == 0, mism                                    # bit-exact token-id argmax decode
    norms = np.array([m2.view(t, m2.grads).norm().item() for t in m2.param_infos()])
    assert np.abs(norms / gold["grad_norms"] - 1).max() < 5e-3
    o = E.Bf16OracleModel(variant, 28, mask=True, act="swish", state=R.formula_state(variant, 28), dtype=torch.float64, emulate=False)
    _, _, _, grads = E.loss_and_grads(o, x.double(), tg, pct, tsz)
    rels = {t.name: rel_l2(m2.view(t, m2.grads), g) for t, g in zip(m2.param_infos(), grads)}
    worst = max(rels.values())
    from conftest import record_measured
    record_measured("%s_swish_golden_f32_grad_rel_l2_vs_f64_oracle" % variant, worst)
    assert worst < 2e-3, sorted(rels.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize("variant", ["context", "context_se"])
def test_context_variants_bf16_mode_runs(dev, variant):
    from lightning_asr_amd import ops
    gold = np.load("tests/golden/model_%s.npz" % variant)
    x, tg, pct, tsz = golden_inputs()
    m = _native(variant, 28, dev, torch.bfloat16)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev), torch.bfloat16)
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert abs(loss.item() - gold["losses"][0]) / gold["losses"][0] < 3e-2
    assert torch.isfinite(m.grads).all()
    norms = np.array([m.view(t, m.grads).norm().item() for t in m.param_infos()])
    assert np.median(np.abs(norms / gold["grad_norms"] - 1)) < 8e-2


@pytest.mark.parametrize("n_buckets", [4, 2, 1])
def test_staged_backward_equals_single_call(dev, n_buckets):
    """The bucketed (overlap-ready) backward produces the same gradients, bucket by bucket, as the single call."""
    from lightning_asr_amd import ops
    x, tg, pct, tsz = golden_inputs()
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    m1 = _native("plain", 28, dev)
    m1.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    m2 = _native("plain", 28, dev)
    m2.grads.fill_(float("nan"))
    seen = []

    def on_bucket(ranges):
        torch.cuda.synchronize()
        assert len(ranges) == 1                                   # plain variant: every bucket is one contiguous piece
        lo, hi = ranges[0]
        assert torch.isfinite(m2.grads[lo:hi]).all()             # this bucket is final ...
        seen.append((lo, hi))
    m2.loss_backward_staged(feats, pct.to(dev), tg.to(dev), tsz.to(dev), on_bucket, n_buckets=n_buckets)
    assert seen[0][1] == m2.n_param and seen[-1][0] == 0 and len(seen) == n_buckets
    assert all(a[0] == b[1] for a, b in zip(seen[:-1], seen[1:]))  # contiguous, reverse layer order
    assert torch.equal(m1.grads, m2.grads)
    assert m2.unit_names()[0] == "first_cnn" and m2.unit_names()[-1] == "last_cnn2"


def test_dropout_kwarg_trains_and_is_off_in_eval(dev):
    """MyModel2(drop_rate=0.2) (models/QuartNet.py:265): training forwards differ from step to step and from the p = 0 model,
    the same seed reproduces the same trajectory, eval mode ignores dropout."""
    from lightning_asr_amd import ops
    from lightning_asr_amd.models.QuartNet import MyModel2
    labels = [c.strip() for c in open("data/labels.txt").readlines()]
    x, tg, pct, tsz = golden_inputs()

    def make(p):
        torch.manual_seed(3)
        mm = MyModel2(labels, drop_rate=p, mask=True, device=str(dev))
        mm.load_state_dict(R.formula_state("plain", 28))
        return mm
    m0, m1, m2 = make(0.0), make(0.2), make(0.2)
    m0.train(); m1.train(); m2.train()
    a0 = m0(x.to(dev), pct.to(dev))
    a1 = m1(x.to(dev), pct.to(dev))
    a1.sum().backward()                          # backward through the masked plan runs (same masks: tests/test_gpu_units.py)
    assert torch.isfinite(m1.native.grads).all() and m1.native.grads.abs().max() > 0
    a1 = a1.detach()
    a1b = m1(x.to(dev), pct.to(dev)).detach()
    a2 = m2(x.to(dev), pct.to(dev))
    assert torch.isfinite(a1).all()
    assert (a1 - a0).abs().max() > 1e-3          # dropout changes the forward
    assert (a1 - a1b).abs().max() > 1e-3         # a fresh mask per training forward (the device step counter moved)
    assert torch.equal(a1, a2)                   # same seed, same step index: same masks
    m0.eval(); m1.eval()
    # (running statistics differ after the training forwards above: compare eval outputs of two p = 0.2 models instead)
    m2.eval()
    m1.load_state_dict(m2.state_dict())
    e1, e2 = m1(x.to(dev), pct.to(dev)), m2(x.to(dev), pct.to(dev))
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("variant,B,T_in,pcts,S", [("plain", 1, 41, [1.0], 3), ("plain", 2, 33, [1.0, 0.3], 2),
                                                  ("plain", 3, 201, [1.0, 0.1, 0.55], 6), ("context_se", 2, 65, [1.0, 0.4], 4),
                                                  ("context", 1, 33, [0.8], 2)])
def test_small_and_ragged_shapes_match_oracle(dev, variant, B, T_in, pcts, S):
    """Edge shapes the kernels' tilings do not see at the BASELINE sizes: one utterance, T' = 17 (the shortest batch the library takes is T_in = 33
    frames, one first_cnn window: shorter than every depthwise kernel and than one BN slab), utterances cut to a tenth of the batch's length.  f32 mode against the
    f64 oracle end to end (forward log-probs tight; gradients at the end-to-end noise level of DESIGN §2), and the bf16 mode must
    stay finite and close on the loss."""
    from lightning_asr_amd import ops
    from oracle import ref_bf16 as E
    g = torch.Generator().manual_seed(B * 100 + T_in)
    x = torch.randn(B, 1, 64, T_in, generator=g)
    pct = torch.tensor(pcts, dtype=torch.float32)
    lens_in = (T_in * pct).int()
    x = x * (torch.arange(T_in).view(1, 1, 1, T_in) < lens_in.view(B, 1, 1, 1))
    tg = torch.randint(0, 27, (B, S), generator=g)
    for b in range(B):
        for s in range(1, S):
            if tg[b, s] == tg[b, s - 1]:
                tg[b, s] = (tg[b, s] + 1) % 27
    Tp = (T_in - 1) // 2 + 1
    tl = (Tp * pct).int()
    tsz = torch.minimum(torch.full((B,), S, dtype=torch.int32), torch.clamp(tl // 2, min=1).int())
    o = E.Bf16OracleModel(variant, 28, mask=True, state=R.formula_state(variant, 28), dtype=torch.float64, emulate=False)
    loss_ref, nll_ref, lp_ref, grads = E.loss_and_grads(o, x.double(), tg, pct, tsz)
    m = _native(variant, 28, dev)
    feats = ops.bct_to_btc(x[:, 0].contiguous().to(dev))
    loss, nll, lp, am = m.loss_backward(feats, pct.to(dev), tg.to(dev), tsz.to(dev))
    assert torch.isfinite(lp).all() and torch.isfinite(m.grads).all()
    assert (lp.cpu().double() - lp_ref).abs().max() < 2e-3
    assert abs(loss.item() - loss_ref) / abs(loss_ref) < 1e-4
    rels = {t.name: rel_l2(m.view(t, m.grads), gr) for t, gr in zip(m.param_infos(), grads)}
    worst = max(rels.values())
    from conftest import e2e_gate, record_measured
    record_measured("edge_%s_B%d_T%d_f32_grad_rel_l2_vs_f64_oracle" % (variant, B, T_in), worst)
    assert worst < e2e_gate("edge_%s_B%d_T%d_f32_grad_rel_l2_vs_f64_oracle" % (variant, B, T_in)), sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    mb = _native(variant, 28, dev, dtype=torch.bfloat16)
    lossb, _, lpb, _ = mb.loss_backward(feats.to(torch.bfloat16), pct.to(dev), tg.to(dev), tsz.to(dev))
    assert torch.isfinite(lpb).all() and torch.isfinite(mb.grads).all()
    assert abs(lossb.item() - loss_ref) / abs(loss_ref) < 5e-2
