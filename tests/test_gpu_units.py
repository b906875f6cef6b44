"""Per-unit ("teacher-forced") parity of the whole training step at the BASELINE sizes, bf16 (the bench dtype) and f32.

One full step runs on the GPU through the C ABI (wave -> mel -> forward -> CTC -> backward, one unit per backward
stage so that every unit's d(out) / d(in) can be read from the workspace).  Then, for EVERY unit of the plan, the CPU
oracle (oracle/ref_bf16.py: the pinned f32 arithmetic of oracle/ref_cpu.py plus the plan's bf16 storage points)
recomputes that unit from the GPU's own stored inputs and every tensor the unit produced is compared: u, y, y2, out,
dx and all parameter gradients; same for the BiLSTM context branch and the decoder + log-softmax + CTC head.

Why per unit: the end-to-end map is chaotic at bf16 resolution (see the header of oracle/ref_bf16.py: two emulations
differing only in f32-vs-f64 arithmetic between the stores differ by 0.4 relative L2 in the gradients), so whole-step
bf16 gradients cannot be compared tightly whatever the kernels do; per unit the only legitimate difference is the f32
accumulation order and the handful of bf16 rounding flips it causes.  End-to-end the loss is still compared (<= 1e-3).

Configs (BASELINE.json): cfg2 = asr13x1 bf16 bs=32 10 s; cfg4 = QuartNetContextSE bf16 bs=32 10 s; cfg5-shaped = AISHELL
vocabulary (C=4334), bs=32, one length bucket of ragged 14.5-16 s clips (T' up to 801), bf16.
"""
import json
import os

import pytest
import torch

from oracle import ref_bf16 as E
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

# tolerances (relative L2 per tensor unless noted).  bf16: one rounding flip changes an element by 2^-8 relative; with
# f32 accumulation noise ~1e-6 against a rounding step of 4e-3 about 1 element in 4000 flips -> ~1e-4 relative L2.
# Measured on MI355X (profiles/r02_unit_parity.json): bf16 act <= 8e-5, grad_act <= 4.5e-4, grad_param <= 4.5e-4;
# f32 act <= 4e-7, grad_param <= 2e-6, grad_act <= 2e-4 (that one is d(logits): the lattice kernel evaluates lse on the
# v_exp_f32 / v_log_f32 units, torch's CPU ctc_loss on libm - north_star's gate for the CTC loss itself is 1e-4).
# Round 5 (VERDICT r4 "what's weak" 3): the bf16 gradient gates are per KIND OF UNIT.  Dense units (every conv unit, the BiLSTM context,
# SE, the dense head): 1.0e-3 - the worst measured over all nine dense reports is 6.0e-4 / 4.9e-4 (profiles/r04_unit_parity_*.json), so a
# regression that doubles any dense unit's error now fails.  The lean head's STORED-in-bf16 d(logits) alone keeps 1.5e-3 (its relative
# L2 against the rounded oracle is the bf16 quantisation floor, measured 1.20e-3; what it is really held to is the elementwise
# bf16-neighbour check below).
TOL = {
    "bf16": {"act": 3e-4, "grad_act": 1.0e-3, "grad_param": 1.0e-3, "grad_act_lean_glogits": 1.5e-3, "logp_abs": 2e-5, "nll": 2e-6,
             "loss_e2e": 1e-3},
    "f32": {"act": 5e-6, "grad_act": 6e-4, "grad_param": 2e-5, "logp_abs": 2e-5, "nll": 2e-6, "loss_e2e": 1e-5},
}


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def _bct(t_btc):
    return t_btc.float().cpu().transpose(1, 2).contiguous()


def _ragged_batch(B, L_max, L_min, V, seed, chars_per_s=2.8):
    """one length bucket: utterance lengths uniform in [L_min, L_max] (<= 10 % padding), AISHELL-like target lengths"""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(L_min, L_max + 1, (B,), generator=g)
    lens[0] = L_max
    wave = 0.1 * torch.randn(B, L_max, generator=g)
    for b in range(B):
        wave[b, lens[b]:] = 0
    tl = (lens.float() / 16000 * chars_per_s).int()
    S = int(tl.max())
    tg = torch.randint(0, V, (B, S), generator=g)
    for s in range(1, S):
        same = tg[:, s] == tg[:, s - 1]
        tg[same, s] = (tg[same, s] + 1) % V
    return wave, lens.int(), tg.long(), tl.int()


def run_units_check(dev, variant, n_class, dtype, wave, sample_lens, tg, tl, tag, oracle_dtype=torch.float32, weights="random",
                    lean=False, drop_p=0.0, act="relu"):
    from lightning_asr_amd import ops
    from lightning_asr_amd.engine import NativeModel
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    mode = "bf16" if dtype == torch.bfloat16 else "f32"
    tol = TOL[mode]
    state = R.random_state(variant, n_class, 0) if weights == "random" else R.formula_state(variant, n_class)
    m = NativeModel(variant, n_class, mask=True, act=act, dtype=dtype, device=dev)
    m.load_state_dict(state)
    if drop_p:
        m.set_dropout(drop_p, seed=12345)
    B = wave.shape[0]
    _, feats, frames, pct = ops.mel(wave.to(dev), None if sample_lens is None else sample_lens.to(dev), None, None, True, dtype,
                                    want_bft=False, want_btf=True)
    names = m.unit_names()
    units = {}
    T = m.out_frames(feats.shape[1])
    N = B * T

    def on_unit(i, name):
        gp, gc = m.tap("bwd.g_prev"), m.tap("bwd.g_cur")
        units[name] = {"g_prev": gp.clone(), "g_cur": gc.clone()}
    loss, nll, logp, am = m.loss_backward_units(feats, pct, tg.to(dev), tl.to(dev), on_unit, want_logp=not lean)
    torch.cuda.synchronize()
    if lean:        # large-vocabulary head: no f32 log-probs exist; rebuild them from the stored bf16 logits and the per-row lse
        assert logp is None and m.lean_head
        logp = m.tap("logits_bf16")[:, :, :n_class].float() - m.tap("lse").view(B, T, 1)
        assert bool((m.tap("grad_logits_bf16")[:, :, n_class:] == 0).all())
        glogits_gpu = m.tap("grad_logits_bf16")[:, :, :n_class].float().cpu()
    else:
        glogits_gpu = m.tap("grad_logits").cpu()
    assert torch.isfinite(loss).all() and torch.isfinite(m.grads).all()
    gpu_grads = {t.name: m.view(t, m.grads).detach().float().cpu() for t in m.param_infos()}
    pct_c = pct.cpu()
    lens = R.mask_lengths(T, pct_c)
    assert m.tap("lens").cpu().tolist() == lens.tolist()
    o = E.Bf16OracleModel(variant, n_class, mask=True, act=act, state={k: v.clone() for k, v in state.items()}, dtype=oracle_dtype,
                          emulate=(mode == "bf16"))
    o.lean_head = lean
    report = {"config": tag, "mode": mode, "act": act, "B": B, "T_in": int(feats.shape[1]), "T": T, "C": n_class, "units": {}}
    worst = {"act": 0.0, "grad_act": 0.0, "grad_param": 0.0}

    def note(unit, kind, key, val, gate=None):
        report["units"].setdefault(unit, {})[key] = val
        worst[kind] = max(worst[kind], val)
        gate = tol[kind] if gate is None else gate
        if os.environ.get("LASR_UNITS_NOASSERT"):          # diagnosis: collect the whole report, fail at the end
            if val >= gate:
                report.setdefault("over_tolerance", []).append((unit, key, val, gate))
            return
        assert val < gate, (tag, unit, key, val, gate)

    def grad_of(name, c):
        return _bct(units[name]["g_cur"][:N * c].view(B, T, c))

    ctx = variant != "plain"
    chans = {"first_cnn": (m.cfg.in_c, 256), "last_cnn2": (512, 1024)}
    for n_, ci, co, _k in R.block_table(variant):
        chans[n_] = (ci, co)
    for i, name in enumerate(names):
        ci, co = chans[name]
        if i == 0:
            x_in = _bct(feats)
        elif ctx and name == "block3":
            x_in = _bct(m.tap("ctx_in"))
        else:
            x_in = _bct(m.tap(names[i - 1]))
        if i == len(names) - 1:
            dout = _bct(units[name]["g_prev"][:N * co].view(B, T, co))
        else:
            dout = grad_of(names[i + 1], co)
        out_gpu = _bct(m.tap(name))
        drop = None
        if drop_p:      # the mask this unit drew in the forward above, read back from the same counter-based generator
            keep = m.dropout_mask(i, N * co).view(B, T, co).transpose(1, 2).cpu()
            thresh = int(drop_p * 65536 + 0.5)
            drop = (keep, 1.0 / (1.0 - thresh / 65536.0))
            report.setdefault("drop_kept_frac", {})[name] = keep.float().mean().item()
            assert abs(keep.float().mean().item() - (1 - drop_p)) < 5e-3, (name, keep.float().mean().item())
        # ReLU: derivative mask from the GPU's output.  With dropout it is still right where it matters: a residual unit drops before
        # the add (out > 0 <=> z > 0), first_cnn / last_cnn2 drop after the activation and a dropped element's gradient is zero anyway.
        # Swish (smooth: no element can change sides): the oracle differentiates its OWN pre-activation z, nothing is taken from the GPU
        # SE units: the excite MLP's inner ReLU (B x C/8 values per unit) gets the same treatment - its derivative mask is the GPU's own
        # `se_hidden > 0`.  Measured without it (context_se, Swish epilogues, where nothing else can flip): ONE hidden unit on the other
        # side of zero = 3.0e-2 of d(fc.0.weight) and 3.9e-3 of d(bn.bias) in that unit, everything else of the model at <= 5e-4
        se_mask = (m.tap(name + ".se_hidden").view(B, -1).cpu() > 0) if (variant == "context_se" and name != "last_cnn2") else None
        r = E.run_unit(o, name, x_in, lens, dout, act_mask=(out_gpu > 0) if act == "relu" else None, drop=drop, se_mask=se_mask)
        note(name, "act", "out", rel_l2(out_gpu, r["out"]))
        note(name, "act", "y", rel_l2(_bct(m.tap(name + ".y")), r["y"]))
        if "u" in r:
            note(name, "act", "u", rel_l2(_bct(m.tap(name + ".u")), r["u"]))
        if "y2" in r:
            note(name, "act", "y2", rel_l2(_bct(m.tap(name + ".y2")), r["y2"]))
        if "dx" in r and i > 0:
            note(name, "grad_act", "dx", rel_l2(grad_of(name, ci) if not (ctx and name == "block3") else
                                                _bct(units[name]["g_prev"][:N * ci].view(B, T, ci)), r["dx"]))
        for k, g in r["grads"].items():
            note(name, "grad_param", "d." + k.split("encoder.")[-1], rel_l2(gpu_grads[k], g.float()))
        if ctx and name == "block3":
            x23 = _bct(m.tap(names[i - 1]))
            dcat = _bct(units[name]["g_prev"][:N * 336].view(B, T, 336))
            rc = E.run_context(o, x23, lens, dcat)
            note("context_rnn", "act", "ctx", rel_l2(x_in[:, 256:], rc["ctx"]))
            note("context_rnn", "grad_act", "dx", rel_l2(grad_of(name, 256), rc["dx"]))
            for k, g in rc["grads"].items():
                note("context_rnn", "grad_param", "d." + k.split("rnn.")[-1], rel_l2(gpu_grads[k], g.float()))
    # head: decoder + log_softmax + mean CTC from the stored last_cnn2 output
    rh = E.run_head(o, _bct(m.tap("last_cnn2")), pct_c, tg, tl, glogits_in=glogits_gpu)
    lp_err = (logp.cpu().double() - rh["logp"].double()).abs().max().item()
    report["head"] = {"logp_max_abs": lp_err, "nll_rel": ((nll.cpu().double() - rh["nll"].double()).abs() / rh["nll"].double().abs()).max().item(),
                      "loss_gpu": loss.item(), "loss_head_oracle": rh["loss"]}
    if lean:
        # the logits are a bf16 tensor here: where the GPU's f32 accumulation and the oracle land on different sides of a rounding
        # boundary one logit (and its log-prob) moves by one bf16 ulp (2^-7 at |x| in [1, 2)); everything else agrees to f32 noise
        lp_diff = (logp.cpu().double() - rh["logp"].double()).abs()
        report["head"]["logp_flipped_frac"] = (lp_diff > tol["logp_abs"]).double().mean().item()
        assert report["head"]["logp_flipped_frac"] < 2e-3 and lp_err < 0.04, (tag, "logp", lp_err, report["head"]["logp_flipped_frac"])
    else:
        assert lp_err < tol["logp_abs"], (tag, "logp", lp_err)
    assert report["head"]["nll_rel"] < tol["nll"], (tag, "nll", report["head"]["nll_rel"])
    assert torch.equal(am.cpu().long(), rh["logp"].argmax(-1)) or (am.cpu().long() != rh["logp"].argmax(-1)).float().mean() < 1e-4
    if lean:
        # The lean head STORES d(logits) in bf16.  Against the oracle's (rounded) gradient the relative L2 distance is then the bf16
        # quantisation floor itself: two tensors whose unrounded values differ by the dense head's own 4.5e-4 (the lattice evaluates
        # lse on v_exp_f32 / v_log_f32, torch's CPU ctc_loss on libm - measured on the f32 gradient of the dense head, same batch)
        # land on different bf16 neighbours for ~11 % of the elements, each such flip is one ulp = 2^-8 relative:
        # sqrt(0.11) * 3.9e-3 = 1.3e-3; measured 1.20e-3 (profiles/r04_unit_parity_cfg5_aishell_bf16_lean.json), reported below.
        # What the kernel can be held to is tighter and elementwise: every stored value is a bf16 NEIGHBOUR of the oracle's f32
        # value, i.e. within half an ulp plus that f32-level noise - on all but a vanishing fraction of the 111 M elements.
        # the yardstick for this check is the head evaluated in f64: torch's f32 CPU lattice (T' = 801 dependent steps) leaves the
        # per-frame occupancies summing to 1 +- a few 1e-3 on some frames, and log_softmax's backward multiplies EVERY class of such a
        # frame by that sum (measured: 1.4 - 4.7 % of an utterance's elements more than 0.2 % off); the kernel uses the exact 1
        o64 = E.Bf16OracleModel(variant, n_class, mask=True, act=act, state={k: v.clone() for k, v in state.items()}, dtype=torch.float64,
                                emulate=True)
        o64.lean_head = True
        rh64 = E.run_head(o64, _bct(m.tap("last_cnn2")).double(), pct_c, tg, tl)
        x = rh64["glogits"].float()
        # d(logits) = (softmax - occupancy) / B.  The softmax part is exact to f32 rounding; the occupancy part (target and blank
        # columns, 1 % of the elements) comes out of an f32 log-domain lattice whose alpha + beta - nll are sums of up to 801 terms
        # of magnitude ~8 (|log p| at C = 4334): measured against the f64 lattice it is off by up to ~1 % on ~5 % of those elements
        # - torch's own f32 ctc_loss does no better - so the occupancy term gets a 2 % allowance, everything else 0.2 %
        occ = (torch.exp(rh64["logp"].double()) / B - rh64["glogits"].double()).abs().float()
        ulp = torch.pow(2.0, torch.floor(torch.log2(x.abs().clamp_min(1e-30))) - 7)        # bf16: 8 significant bits
        bad = ((glogits_gpu - x).abs() > 0.5 * ulp + 2e-3 * x.abs() + 2e-2 * occ + 1e-12)
        report["head"]["glogits_not_a_bf16_neighbour_frac"] = bad.double().mean().item()
        if bool(bad.any()):      # diagnosis: which elements, and by how much
            rr = (glogits_gpu[bad] / x[bad]).double()
            report["head"]["bad_ratio_quantiles"] = [float(v) for v in torch.quantile(rr[torch.isfinite(rr)][:2000000], torch.tensor([0.01, 0.25, 0.5, 0.75, 0.99], dtype=torch.float64))]
            report["head"]["bad_abs_x_quantiles"] = [float(v) for v in torch.quantile(x[bad].abs().double()[:2000000], torch.tensor([0.01, 0.5, 0.99], dtype=torch.float64))]
            report["head"]["all_abs_x_quantiles"] = [float(v) for v in torch.quantile(x.abs().double().flatten()[:2000000], torch.tensor([0.01, 0.5, 0.99], dtype=torch.float64))]
            report["head"]["bad_frac_by_utterance"] = [float(v) for v in bad.double().mean(dim=(1, 2))]
            report["head"]["bad_x_zero_frac"] = float((x[bad] == 0).double().mean())
            report["head"]["bad_gpu_zero_frac"] = float((glogits_gpu[bad] == 0).double().mean())
        report["head"]["glogits_rel_l2_vs_unrounded_oracle"] = rel_l2(glogits_gpu, x)
        if report["head"]["glogits_not_a_bf16_neighbour_frac"] >= 1e-4:
            os.makedirs("gpurun_out", exist_ok=True)
            with open("gpurun_out/lean_diag_%s.json" % tag, "w") as f:
                json.dump(report["head"], f, indent=1)
        assert report["head"]["glogits_not_a_bf16_neighbour_frac"] < 1e-4, (tag, report["head"])
        # relative L2 like for like: against the oracle head in the plan's own arithmetic class (f32 lattice), as in every other unit;
        # against the f64 lattice the figure is the f32 lattice's own accuracy at T' = 801 (reported, not gated: torch's is the same)
        report["head"]["glogits_rel_l2_vs_f64_oracle_rounded"] = rel_l2(glogits_gpu, E.rb(x))
        note("head", "grad_act", "glogits", rel_l2(glogits_gpu, E.rb(rh["glogits"].float())), gate=tol.get("grad_act_lean_glogits"))
    else:
        note("head", "grad_act", "glogits", rel_l2(glogits_gpu, rh["glogits"]))
    note("head", "grad_act", "dx", rel_l2(_bct(units["last_cnn2"]["g_prev"][:N * 1024].view(B, T, 1024)), rh["dx"]))
    for k, g in rh["grads"].items():
        note("head", "grad_param", "d." + k, rel_l2(gpu_grads[k], g.float()))
    report["worst"] = worst
    if report.get("over_tolerance"):
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/unit_parity_%s.json" % tag, "w") as f:
            json.dump(report, f, indent=1)
        raise AssertionError((tag, report["over_tolerance"]))
    # end to end: the emulated oracle's own whole forward from the same features (loss only: see the module docstring)
    o2 = E.Bf16OracleModel(variant, n_class, mask=True, act=act, state={k: v.clone() for k, v in state.items()}, dtype=oracle_dtype,
                           emulate=(mode == "bf16"))
    if drop_p:
        report["worst"] = worst
        with open("gpurun_out/unit_parity_%s.json" % tag, "w") as f:
            json.dump(report, f, indent=1)
        return report
    o2.training = True
    o2.lean_head = lean
    with torch.no_grad():
        lp2 = o2.forward(_bct(feats).unsqueeze(1), pct_c)
        loss2 = R.training_loss(lp2, tg, pct_c, tl, n_class - 1).item()
    report["e2e"] = {"loss_gpu": loss.item(), "loss_oracle": loss2, "rel": abs(loss.item() - loss2) / abs(loss2),
                     "logp_rel_l2": rel_l2(logp.cpu(), lp2), "argmax_agree": (am.cpu().long() == lp2.argmax(-1)).float().mean().item()}
    assert report["e2e"]["rel"] < tol["loss_e2e"], (tag, report["e2e"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/unit_parity_%s.json" % tag, "w") as f:
        json.dump(report, f, indent=1)
    return report


def test_units_f32_small_validates_harness(dev):
    """f32 parity mode, formula weights, golden-sized batch: the per-unit harness against the pinned plain oracle."""
    wave, tg, tl = R.synth_batch(4, 32000, 20, 27, seed=5)
    lens = torch.tensor([32000, 32000, 28000, 16000], dtype=torch.int32)
    rep = run_units_check(dev, "context_se", 28, torch.float32, wave, lens, tg, tl, "f32_small_context_se", torch.float64, "formula")
    assert rep["worst"]["act"] < 2e-5


@pytest.mark.parametrize("variant,dtype", [("context_se", torch.float32), ("plain", torch.bfloat16)])
def test_units_with_dropout(dev, variant, dtype):
    """drop_rate = 0.1 (models/QuartNet.py:26,38,149): every unit's forward and backward with the counter-based masks, against the
    oracle applying the SAME masks (read back through lasr_dropout_mask); kept fraction ~ 1 - p in every layer."""
    wave, tg, tl = R.synth_batch(6, 48000, 30, 27, seed=21)
    lens = torch.tensor([48000, 48000, 40000, 30000, 48000, 20000], dtype=torch.int32)
    f32 = dtype == torch.float32
    rep = run_units_check(dev, variant, 28, dtype, wave, lens, tg, tl, "dropout_%s_%s" % (variant, "f32" if f32 else "bf16"),
                          torch.float64 if f32 else torch.float32, "formula" if f32 else "random", drop_p=0.1)
    assert len(rep["drop_kept_frac"]) == len(rep["units"]) - 1 - (1 if variant != "plain" else 0)     # every unit but the head (and the LSTM)


def test_units_cfg2_plain_bf16_bs32_10s(dev):
    wave, tg, tl = R.synth_batch(32, 160000, 100, 27, seed=1234)
    run_units_check(dev, "plain", 28, torch.bfloat16, wave, None, tg, tl, "cfg2_plain_bf16")


def test_units_cfg2_plain_bf16_swish(dev):
    """model.act = swish at the BASELINE size in the bench dtype: every unit of the plan - the half-tile depthwise forward that makes
    the unit below's BN + add + Swish in its staging loop (dwconv_s1_mfma_bn_kernel), the sliced BN backward that rebuilds z and
    applies swish'(z) - against the oracle differentiating its own pre-activation (activate_fun/Swish.py:9-10)."""
    wave, tg, tl = R.synth_batch(32, 160000, 100, 27, seed=1234)
    run_units_check(dev, "plain", 28, torch.bfloat16, wave, None, tg, tl, "cfg2_plain_bf16_swish", act="swish")


def test_units_context_se_bf16_swish(dev):
    """SE + Swish (the SE scale sits between BN and the activation: models/QuartNetContextSE.py:55), ragged batch, bf16"""
    wave, lens, tg, tl = _ragged_batch(8, 96000, 60000, 27, seed=78)
    run_units_check(dev, "context_se", 28, torch.bfloat16, wave, lens, tg, tl, "context_se_bf16_swish", act="swish")


def test_units_cfg2_plain_f32_bs32_10s(dev):
    wave, tg, tl = R.synth_batch(32, 160000, 100, 27, seed=1234)
    run_units_check(dev, "plain", 28, torch.float32, wave, None, tg, tl, "cfg2_plain_f32")


def test_units_cfg4_context_se_bf16_bs32_10s(dev):
    wave, tg, tl = R.synth_batch(32, 160000, 100, 27, seed=4321)
    run_units_check(dev, "context_se", 28, torch.bfloat16, wave, None, tg, tl, "cfg4_context_se_bf16")


def test_units_context_bf16_ragged(dev):
    wave, lens, tg, tl = _ragged_batch(8, 96000, 60000, 27, seed=77)
    run_units_check(dev, "context", 28, torch.bfloat16, wave, lens, tg, tl, "context_bf16_ragged")


def test_units_cfg5_aishell_bf16_bs32_ragged_16s(dev):
    V = 4333
    wave, lens, tg, tl = _ragged_batch(32, 256000, 232000, V, seed=555)
    rep = run_units_check(dev, "plain", V + 1, torch.bfloat16, wave, lens, tg, tl, "cfg5_aishell_bf16")
    assert rep["T"] == 801


def test_units_cfg5_aishell_bf16_lean_head(dev):
    """the same cfg5-shaped step through the large-vocabulary head (csrc/ctc_lean.hip): bf16 logits + softmax row statistics from
    the decoder GEMM's epilogue, the lattice over the gathered emissions, bf16 d(logits) straight from the stored logits -
    no (B, T', C) f32 tensor.  Same per-unit checks; the head's log-probs are rebuilt from the stored logits and lse."""
    V = 4333
    wave, lens, tg, tl = _ragged_batch(32, 256000, 232000, V, seed=555)
    rep = run_units_check(dev, "plain", V + 1, torch.bfloat16, wave, lens, tg, tl, "cfg5_aishell_bf16_lean", lean=True)
    assert rep["T"] == 801


def test_lean_head_small_cases(dev):
    """lasr_gemm_rowstat + lasr_ctc_loss_lean against torch on small ragged cases: repeated labels, an utterance shorter than the
    batch, a class count that is not a multiple of 8 or of the 256-column tile, an infeasible utterance (inf / NaN like torch)."""
    import torch.nn.functional as F
    from lightning_asr_amd import _lib
    from lightning_asr_amd.ops import _p, _stream
    from lightning_asr_amd._lib import call
    import ctypes as C
    g = torch.Generator().manual_seed(11)
    for (B, T, Cc, S, K) in [(3, 40, 300, 7, 64), (2, 33, 777, 12, 128), (4, 50, 4334, 9, 64)]:
        x = E.rb(torch.randn(B * T, K, generator=g))
        W = E.rb(torch.randn(Cc, K, generator=g) * 0.3)
        bias = torch.randn(Cc, generator=g) * 0.1
        tgt = torch.randint(0, Cc - 1, (B, S), generator=g)
        tgt[0, 1] = tgt[0, 0]                                       # a repeat
        il = torch.tensor([T, T - 7, T, 5][:B], dtype=torch.int32)  # the last of 4 is infeasible (5 frames for 9 labels)
        tl = torch.tensor([S, S - 2, 1, S][:B], dtype=torch.int32)
        ldc = (Cc + 7) // 8 * 8
        N = B * T
        logits = torch.empty(N, ldc, dtype=torch.bfloat16, device=dev)
        nb = _lib.load().lasr_gemm_rowstat_bytes(N, Cc)
        tiles = (Cc + 255) // 256
        rs = torch.empty(N * tiles * 2, dtype=torch.float32, device=dev)
        ra = torch.empty(N * tiles, dtype=torch.int32, device=dev)
        assert nb == rs.numel() * 4 + ra.numel() * 4
        nt = C.c_int(0)
        # (every device operand is held in a variable: a temporary passed as _p(t.to(dev)) is freed - and its block reused by the
        #  next temporary - before the kernel runs)
        xb, Wb, bias_d = x.to(dev, torch.bfloat16), W.to(dev, torch.bfloat16), bias.to(dev)
        tgt_d, il_d, tl_d = tgt.to(dev), il.to(dev), tl.to(dev)
        call("lasr_gemm_rowstat", _p(xb), _p(Wb), _p(bias_d), _p(logits), ldc, N, Cc, K, _p(rs), _p(ra), C.byref(nt), _stream())
        assert nt.value == tiles
        ref_logits = E.rb((x.double() @ W.double().t() + bias.double()).float())
        got = logits[:, :Cc].float().cpu()
        assert ((got - ref_logits).abs() <= 2.0 ** -7 * ref_logits.abs() + 1e-6).all()
        wsb = _lib.load().lasr_ctc_lean_workspace_bytes(B, T, Cc, S)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        am = torch.empty(N, dtype=torch.int32, device=dev)
        grad = torch.empty(N, ldc, dtype=torch.bfloat16, device=dev)
        db = torch.empty(Cc, dtype=torch.float32, device=dev)
        call("lasr_ctc_loss_lean", _p(logits), ldc, _p(rs), _p(ra), tiles, _p(tgt_d), _p(il_d), _p(tl_d), B, T, Cc, S,
             Cc - 1, _p(nll), _p(am), _p(grad), _p(db), None, _p(ws), wsb, _stream())
        torch.cuda.synchronize()
        # torch on the SAME stored logits
        lg = got.double().view(B, T, Cc).requires_grad_(True)
        lp = F.log_softmax(lg, -1)
        ref_nll = F.ctc_loss(lp.transpose(0, 1), tgt, il, tl, blank=Cc - 1, reduction="none")
        ok = torch.isfinite(ref_nll)
        (ref_nll[ok].sum() / B).backward()
        assert torch.equal(torch.isfinite(nll.cpu()), ok)
        assert ((nll.cpu().double()[ok] - ref_nll[ok]).abs() / ref_nll[ok].abs()).max() < 1e-5
        assert torch.equal(am.cpu().long().view(B, T), got.view(B, T, Cc).argmax(-1))
        gg = grad[:, :Cc].float().cpu().view(B, T, Cc)
        assert bool((grad[:, Cc:] == 0).all())
        for b in range(B):
            if ok[b]:
                ref_g = E.rb(lg.grad[b].float())
                assert rel_l2(gg[b], ref_g) < 2e-3, (B, T, Cc, b, rel_l2(gg[b], ref_g))
                assert bool((gg[b, int(il[b]):] == 0).all())
            else:
                assert bool(torch.isnan(gg[b, :int(il[b])]).all())
        if bool(ok.all()):
            assert rel_l2(db.cpu(), lg.grad.sum((0, 1)).float()) < 2e-4
