"""CPU tier: the oracle (oracle/ref_cpu.py) against the golden vectors captured from the reference
(tests/golden/*.npz, written by oracle/make_golden.py in the dev container), plus the host-side
scalar logic of the product against the same vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from oracle.make_golden import checksum, golden_inputs

LABELS = ["'"] + [chr(ord("a") + i) for i in range(26)]


@pytest.mark.parametrize("variant", R.VARIANTS)
def test_oracle_forward_matches_reference_golden(variant):
    gold = np.load("tests/golden/model_%s.npz" % variant)
    x, tg, pct, tsz = golden_inputs()
    m = R.OracleModel(variant, 28, mask=True, state=R.formula_state(variant, 28))
    m.training = False
    with torch.no_grad():
        lp_e = m(x, pct)
    tol = 1e-6 if variant == "plain" else 5e-5
    assert np.abs(lp_e.numpy() - gold["eval_logprobs"]).max() < tol
    m.training = True
    m.keep_taps = True
    with torch.no_grad():
        lp = m(x, pct)
    assert np.abs(lp.numpy() - gold["logprobs"]).max() < (1e-6 if variant == "plain" else 2e-3)
    assert np.array_equal(R.mask_lengths(lp.size(1), pct).numpy(), gold["t_lengths"])
    nll = R.ctc_loss_per_sample(lp, tg, R.mask_lengths(lp.size(1), pct), tsz, 27)
    assert np.abs(nll.numpy() - gold["nll"]).max() / np.abs(gold["nll"]).max() < 1e-4
    if variant == "plain":
        assert np.array_equal(lp.argmax(-1).numpy().astype(np.int16), gold["argmax"])
        for k in gold.files:
            if k.startswith("tap_"):
                assert np.abs(checksum(m.taps[k[4:]]) - gold[k]).max() < 1e-6, k


@pytest.mark.parametrize("variant", ["plain", "context_se"])
def test_oracle_swish_matches_reference_golden(variant):
    """act="swish" (activate_fun/Swish.py:9-10): the oracle against the reference model with its own Swish module swapped into
    the unit epilogues (oracle/make_golden.py::swap_activation): forward, per-sample NLL, gradient norms, one NovoGrad step."""
    gold = np.load("tests/golden/model_%s_swish.npz" % variant)
    x, tg, pct, tsz = golden_inputs()
    m = R.OracleModel(variant, 28, mask=True, act="swish", state=R.formula_state(variant, 28))
    m.training = False
    with torch.no_grad():
        lp_e = m(x, pct)
    assert np.abs(lp_e.numpy() - gold["eval_logprobs"]).max() < (1e-6 if variant == "plain" else 5e-5)
    st = R.NovogradState(len(m.parameters()))
    loss, grads = R.train_step(m, st, x, tg, pct, tsz, 1e-2, 1e-3)
    assert abs(loss / gold["losses"][0] - 1) < 1e-5
    norms = np.array([g.norm().item() for g in grads])
    assert np.abs(norms / gold["grad_norms"] - 1).max() < (1e-5 if variant == "plain" else 2e-3)
    assert np.abs(np.stack([checksum(p) for p in m.parameters()]) - gold["params_after_1"]).max() < (1e-6 if variant == "plain" else 1e-4)
    # a different function from the ReLU model (the fixture is not a copy of model_<variant>.npz)
    assert np.abs(gold["logprobs"] - np.load("tests/golden/model_%s.npz" % variant)["logprobs"]).max() > 1e-2


def test_oracle_train_steps_match_reference_golden():
    gold = np.load("tests/golden/model_plain.npz")
    x, tg, pct, tsz = golden_inputs()
    m = R.OracleModel("plain", 28, mask=True, state=R.formula_state("plain", 28))
    st = R.NovogradState(len(m.parameters()))
    losses = []
    for step in range(3):
        loss, grads = R.train_step(m, st, x, tg, pct, tsz, 1e-2, 1e-3)
        losses.append(loss)
        if step == 0:
            norms = np.array([g.norm().item() for g in grads])
            assert np.abs(norms / gold["grad_norms"] - 1).max() < 1e-5
            assert np.abs(np.stack([checksum(g) for g in grads]) - gold["grad_sample"]).max() < 1e-5
            assert np.abs(np.stack([checksum(p) for p in m.parameters()]) - gold["params_after_1"]).max() < 1e-6
    assert np.abs(np.array(losses) / gold["losses"] - 1).max() < 1e-5
    assert np.abs(np.stack([checksum(p) for p in m.parameters()]) - gold["params_after_3"]).max() < 1e-5


def test_state_shapes_and_param_counts():
    # SURVEY §6 [probe]: parameter counts of the three reference models
    for variant, n_class, count in (("plain", 28, 5044572), ("plain", 4334, 9458222), ("context", 28, 5796812),
                                    ("context_se", 28, 6435788)):
        n = sum(int(np.prod(s)) if len(s) else 1 for k, s in R.state_shapes(variant, n_class) if not R.is_buffer(k))
        assert n == count, (variant, n_class, n)
    assert len(R.state_shapes("plain", 28)) == 184


def test_lr_schedule_matches_reference_golden():
    from lightning_asr_amd.schedule import CosineAnnealingWarmupRestarts
    gold = np.load("tests/golden/lr_schedule.npz")
    a = [int(gold["args"][0]), float(gold["args"][1]), float(gold["args"][2]), float(gold["args"][3]), int(gold["args"][4]), float(gold["args"][5])]
    o = R.CosineWarmupRestarts(*a)
    p = CosineAnnealingWarmupRestarts(None, first_cycle_steps=a[0], cycle_mult=a[1], max_lr=a[2], min_lr=a[3], warmup_steps=a[4], gamma=a[5])
    lo, lp_ = [], []
    for _ in range(len(gold["lr"])):
        lo.append(o.lr); lp_.append(p.lr)
        o.step(); p.step()
    assert np.allclose(lo, gold["lr"], rtol=1e-12, atol=0)
    assert np.allclose(lp_, gold["lr"], rtol=1e-12, atol=0)


def test_mask_lengths_golden():
    gold = np.load("tests/golden/mask_lengths.npz")
    for T, p, l in zip(gold["T"], gold["pct"], gold["lens"]):
        assert int(R.mask_lengths(int(T), torch.tensor([p], dtype=torch.float32))[0]) == int(l)
    assert int(R.mask_lengths(801, torch.tensor([0.3333333]))[0]) == 266


def test_ctc_numpy_restatement_matches_torch():
    g = torch.Generator().manual_seed(0)
    lp = F.log_softmax(torch.randn(3, 12, 6, generator=g), -1)
    tg = torch.tensor([[0, 0, 1, 2], [1, 2, 0, 0], [4, 4, 4, 4]])
    il = torch.tensor([12, 9, 5], dtype=torch.int32)
    tl = torch.tensor([4, 2, 4], dtype=torch.int32)
    lpr = lp.clone().requires_grad_(True)
    ref = F.ctc_loss(lpr.transpose(0, 1), tg, il, tl, blank=5, reduction="none")
    assert torch.isinf(ref[2])                                  # infeasible: 4 equal labels need 7 frames
    ref[:2].sum().backward()
    for b in range(2):
        nll, grad = R.ctc_numpy(lp[b, :il[b]].numpy(), tg[b, :tl[b]].tolist(), 5)
        assert abs(nll - ref[b].item()) < 1e-4
        full = np.exp(lp[b, :il[b]].double().numpy()) + grad
        assert np.abs(full - lpr.grad[b, :il[b]].numpy()).max() < 1e-5


def test_greedy_collapse_and_wer():
    assert R.greedy_collapse([3, 3, 27, 3, 4, 4, 27, 27, 5], 27) == [3, 3, 4, 5]
    assert R.greedy_collapse([27, 27], 27) == []
    assert R.levenshtein("kitten", "sitting") == 3
    assert R.word_error_rate(["a b c", "x"], ["a c", "x y"]) == pytest.approx(2 / 4)
    assert R.word_error_rate(["abc"], ["abd"], use_cer=True) == pytest.approx(1 / 3)
    am = torch.tensor([[0, 0, 27, 1], [2, 27, 27, 27]])
    assert R.greedy_decode(am, torch.tensor([4, 1]), LABELS) == ["'a", "b"]


def test_mel_front_end_shapes_and_collate():
    g = torch.Generator().manual_seed(1)
    y = 0.1 * torch.randn(1, 16000, generator=g)
    f = R.parse_wave(y)
    assert f.shape == (1, 64, R.num_frames(16000)) == (1, 64, 101)
    assert abs(f.mean().item()) < 1e-5 and abs(f.std().item() - 1) < 1e-5
    fb = R.mel_filterbank()
    assert fb.shape == (257, 64) and int((fb > 0).sum()) == 500         # SURVEY §8a a1 [probe]
    y2 = 0.1 * torch.randn(1, 8000, generator=g)
    inputs, targets, pct, tsz = R.collate([f, R.parse_wave(y2)], [[1, 2, 3], [4]])
    assert inputs.shape == (2, 1, 64, 101) and targets.tolist() == [[1, 2, 3], [4, 0, 0]]
    assert pct.tolist() == [1.0, pytest.approx(51 / 101)] and tsz.tolist() == [3, 1]
    # bug-compatible sub-sequence: END index is target_length
    w = torch.arange(100.).view(1, -1)
    assert R.sub_sequence(w, 0.5, 0.5, weight=0.98).shape[1] == 99 - int((100 - 99) * 0.5)


@pytest.mark.parametrize("variant", ["plain", "context_se"])
def test_bf16_oracle_without_rounding_is_the_pinned_oracle(variant):
    """oracle/ref_bf16.py restates the layers with storage-rounding hooks (and its own BiLSTM loop): with the hooks off it
    must be the pinned oracle, so what the per-unit GPU tests check in bf16 mode is that arithmetic plus roundings only."""
    from oracle import ref_bf16 as E
    from oracle.make_golden import golden_inputs
    x, tg, pct, tsz = golden_inputs()
    o = R.OracleModel(variant, 28, state=R.formula_state(variant, 28))
    l0, nll0, lp0, g0 = E.loss_and_grads(o, x, tg, pct, tsz)
    e = E.Bf16OracleModel(variant, 28, state=R.formula_state(variant, 28), dtype=torch.float64, emulate=False)
    l1, nll1, lp1, g1 = E.loss_and_grads(e, x, tg, pct, tsz)
    assert abs(l0 - l1) / abs(l1) < 1e-5
    assert (lp0.double() - lp1).abs().max() < 1e-4
    for a, b in zip(g0, g1):
        assert ((a.double() - b).norm() / b.norm()).item() < 5e-3      # f32 vs f64 through this BN stack (DESIGN: 100-400x noise gain)
    # and with the roundings on it stays a small perturbation of the forward (bf16 resolution ~4e-3 per store)
    eb = E.Bf16OracleModel(variant, 28, state=R.formula_state(variant, 28), dtype=torch.float64, emulate=True)
    l2, _, lp2, _ = E.loss_and_grads(eb, x, tg, pct, tsz)
    assert abs(l2 - l1) / abs(l1) < 5e-3


@pytest.mark.parametrize("L", [193, 1000, 16000 + 37, 160000, 255999])
def test_mel_restatement_has_an_independent_second_leg(L):
    """The log-mel front-end is UNPINNED (torchaudio 0.8.1 absent, the reference holds no vectors: DESIGN 2).  What can be had: two
    restatements that share nothing - oracle/ref_cpu.py (torch.stft) and oracle/mel_numpy.py (frames cut by hand with explicit
    reflect indices, numpy rfft, per-corner triangular filters; written from torchaudio 0.8.1's documented Spectrogram / MelScale /
    AmplitudeToDB with the arguments of /root/reference/data_module.py:68-71, chain of :150-174) - agree (a) to f64 round-off when
    both use the first one's f32-rounded window / filter tables, (b) to the tables' f32 round-off (1e-5 of the filter weights,
    2e-5 of the feature scale) when each builds its own, at the BASELINE clip lengths (10 s = 160 000 samples, 16 s)."""
    from oracle import mel_numpy as M
    g = torch.Generator().manual_seed(L)
    y = 0.1 * torch.randn(1, L, generator=g, dtype=torch.float64)
    noise = torch.randn(1, L, generator=g, dtype=torch.float64)
    win32 = R.hann_window_padded().double().numpy()[96:96 + 320]
    fb32 = R.mel_filterbank().double().numpy()
    assert np.abs(win32 - M.hann_periodic(320)).max() < 1e-7 and np.abs(R.hann_window_padded().numpy()[:96]).max() == 0
    assert np.abs(fb32 - M.filterbank()).max() < 1e-5
    assert ((fb32 > 0) == (M.filterbank() > 1e-6)).mean() > 0.999          # same support (up to bins sitting on a corner)
    # (a) power mel spectrum and the whole chain on shared tables: f64 round-off
    yp = R.preemphasis(y + 1e-5 * noise)
    a = R.mel_power(yp)[0].numpy()
    b = (M.power_spectrum(yp[0].numpy(), win32) @ fb32).T
    assert a.shape == b.shape == (64, R.num_frames(L))
    assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max()
    crop = (0.3, 0.7) if L > 4000 else None
    aug = (5, 11, 3, min(7, a.shape[1] - 4)) if a.shape[1] > 8 else None
    fa = R.parse_wave(y, noise, aug, crop=crop)[0].numpy()
    fb_ = M.parse_wave(y[0].numpy(), noise[0].numpy(), aug, crop=crop, window=win32, fb=fb32)
    assert fa.shape == fb_.shape and np.abs(fa - fb_).max() < 1e-9
    # (b) each with its own tables
    fc = M.parse_wave(y[0].numpy(), noise[0].numpy(), aug, crop=crop)
    assert np.abs(fa - fc).max() < 3e-5 * np.abs(fa).max()
