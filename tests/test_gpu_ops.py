"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle / plain torch fp32."""
import math
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ----------------------------------------------------------------------------------------- mel
@pytest.mark.parametrize("kind", ["noise", "tone"])
def test_mel_matches_oracle(dev, kind):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(3)
    L = 16000 * 2 + 123
    if kind == "noise":
        wave = 0.1 * torch.randn(3, L, generator=g)
    else:
        t = torch.arange(L) / 16000.0
        wave = torch.stack([0.3 * torch.sin(2 * math.pi * f * t) + 0.01 * torch.randn(L, generator=g) for f in (440., 1234., 3999.)])
    lens = torch.tensor([L, L - 4000, 8000], dtype=torch.int32)
    dither = torch.randn(3, L, generator=g)
    bft, btf, frames, pct = ops.mel(wave.to(dev), lens.to(dev), dither.to(dev), None, True)
    T = ops.mel_num_frames(L)
    assert bft.shape == (3, 64, T)
    for b in range(3):
        Lb = int(lens[b])
        ref = R.parse_wave(wave[b:b + 1, :Lb], dither[b:b + 1, :Lb])          # (1,64,Tb)
        Tb = ref.shape[2]
        assert int(frames[b]) == Tb == R.num_frames(Lb)
        assert abs(float(pct[b]) - Tb / T) < 1e-7
        got = bft[b, :, :Tb].cpu()
        # north_star tolerance: mel features within 1e-4 relative (to the feature scale, O(1) after
        # normalisation).  A pure tone has ~70 dB of dynamic range, where two f32 FFTs differ in the
        # weak bins by their own round-off: price that with the f64 evaluation of the same oracle.
        ref64 = R.parse_wave(wave[b:b + 1, :Lb].double(), dither[b:b + 1, :Lb].double())
        from conftest import MEL_TOL, record_measured
        assert max_rel(got, ref64[0]) < MEL_TOL, (kind, b, max_rel(got, ref64[0]))
        assert max_rel(got, ref[0]) < MEL_TOL + max_rel(ref[0], ref64[0]), (kind, b, max_rel(got, ref[0]), max_rel(ref[0], ref64[0]))
        record_measured("mel_%s_%d_vs_f64_oracle" % (kind, b), max_rel(got, ref64[0]))
        record_measured("mel_%s_%d_vs_f32_oracle" % (kind, b), max_rel(got, ref[0]))
        record_measured("mel_%s_%d_f32_oracle_vs_f64_oracle" % (kind, b), max_rel(ref[0], ref64[0]))
        assert torch.all(bft[b, :, Tb:] == 0)
        assert torch.equal(btf[b].t().contiguous().cpu(), bft[b].cpu())


def test_mel_shortest_clips(dev):
    """The shortest clips the reference can featurise: MelSpectrogram(pad=32, center=True, n_fft=512) reflect-pads 256 samples of a
    (L + 64)-sample signal, which torch refuses below L = 193 (data_module.py:68-71,160).  From 193 samples on the kernel matches
    the oracle; below (the reference raises) it still writes finite features for its 1 + (L + 64) // 160 frames and zeros behind
    them, so that a stray empty file cannot poison a batch's BatchNorm statistics."""
    from conftest import MEL_TOL
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(11)
    lens_l = [193, 257, 520, 192, 1, 0]
    L = 520
    wave = 0.1 * torch.randn(len(lens_l), L, generator=g)
    dither = torch.randn(len(lens_l), L, generator=g)
    lens = torch.tensor(lens_l, dtype=torch.int32)
    bft, btf, frames, pct = ops.mel(wave.to(dev), lens.to(dev), dither.to(dev), None, True)
    T = ops.mel_num_frames(L)
    for b, Lb in enumerate(lens_l):
        Tb = R.num_frames(Lb)
        assert int(frames[b]) == Tb and abs(float(pct[b]) - Tb / T) < 1e-7
        assert torch.isfinite(bft[b]).all() and torch.all(bft[b, :, Tb:] == 0), Lb
        if Lb >= 193:
            ref64 = R.parse_wave(wave[b:b + 1, :Lb].double(), dither[b:b + 1, :Lb].double())
            assert ref64.shape[2] == Tb
            assert max_rel(bft[b, :, :Tb], ref64[0]) < MEL_TOL, (Lb, max_rel(bft[b, :, :Tb], ref64[0]))
    with pytest.raises(RuntimeError):
        R.parse_wave(wave[3:4, :192], dither[3:4, :192])


def test_mel_crop_after_preemphasis_lead_in(dev, tmp_path):
    """The reference dithers and pre-emphasises the WHOLE clip and crops afterwards (data_module.py:155-159): a crop's first
    sample is y[loc] - 0.97 y[loc-1] (both dithered).  The build crops the raw waveform on the host; the sample before the crop
    travels as a lead-in (LASR_LEN_LEAD in the length word).  Three routes against R.parse_wave(y, dither, crop=(u_len, u_loc)):
    f32 rows with explicit noise, int16 rows straight from lasr_wav_read_batch(lead_in=1), and a crop starting at sample 0
    (no lead-in: y[0] stays unfiltered, as in the reference)."""
    import wave as wavmod
    import numpy as np
    from conftest import MEL_TOL
    from lightning_asr_amd import _lib, ops
    from lightning_asr_amd.ingest import read_wav_batch
    g = torch.Generator().manual_seed(23)
    n_files, L0 = 4, 24000
    pcm = (0.1 * torch.randn(n_files, L0, generator=g)).clamp(-1, 1).mul(32767).round().to(torch.int16)
    # a loud step right before / at the crop point makes the first sample's pre-emphasis term visible far above the tolerance
    crop_u = np.array([[0.3, 0.9], [0.8, 0.5], [0.5, 0.0], [0.0, 0.999]])         # (u_len, u_loc); row 2: loc = 0 -> no lead-in
    paths = []
    for i in range(n_files):
        tgt = int(L0 * (0.98 + 0.02 * crop_u[i, 0])); loc = int(crop_u[i, 1] * (L0 - tgt))
        if loc > 0:
            pcm[i, loc - 1] = 30000
        p = str(tmp_path / ("c%d.wav" % i))
        with wavmod.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm[i].numpy().tobytes())
        paths.append(p)
    out = torch.zeros(n_files * (L0 + 8), dtype=torch.int16)
    lens_w = torch.zeros(n_files, dtype=torch.int32)
    ld = read_wav_batch(paths, out, lens_w, crop_u, 0.98, n_threads=2, lead_in=True)
    rows = out[:n_files * ld].view(n_files, ld)
    leads = [(int(v) >> 30) & 1 for v in lens_w]
    ns = [int(v) & (_lib.LEN_LEAD - 1) for v in lens_w]
    assert leads == [1, 1, 0, 1]
    y_full = pcm.float() / 32768.0
    noise_full = torch.randn(n_files, L0, generator=g)
    # the rows' explicit noise: the full clip's noise, cropped like the samples (lead-in included)
    noise_rows = torch.zeros(n_files, ld)
    for i in range(n_files):
        tgt = int(L0 * (0.98 + 0.02 * crop_u[i, 0])); loc = int(crop_u[i, 1] * (L0 - tgt))
        assert ns[i] == tgt - loc
        assert torch.equal(rows[i, :ns[i] + leads[i]], pcm[i, loc - leads[i]:tgt])
        noise_rows[i, :ns[i] + leads[i]] = noise_full[i, loc - leads[i]:tgt]
    for wave_dev in (rows.to(dev), (rows.float() / 32768.0).to(dev)):          # int16 PCM and f32 rows
        bft, _, frames, _ = ops.mel(wave_dev, lens_w.to(dev), noise_rows.to(dev), None, True)
        for i in range(n_files):
            ref64 = R.parse_wave(y_full[i:i + 1].double(), noise_full[i:i + 1].double(), crop=(float(crop_u[i, 0]), float(crop_u[i, 1])))
            Tb = ref64.shape[2]
            assert int(frames[i]) == Tb
            err = max_rel(bft[i, :, :Tb], ref64[0])
            assert err < MEL_TOL, (i, err)
            # and the crop-first order (what the build did before: the first sample left unfiltered) is measurably different
            if leads[i]:
                tgt = int(L0 * (0.98 + 0.02 * crop_u[i, 0])); loc = int(crop_u[i, 1] * (L0 - tgt))
                wrong = R.parse_wave(y_full[i:i + 1, loc:tgt].double(), noise_full[i:i + 1, loc:tgt].double())
                assert max_rel(wrong[0], ref64[0]) > 10 * MEL_TOL


def test_mel_db_and_specaugment(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(5)
    L = 16000
    wave = 0.05 * torch.randn(2, L, generator=g)
    rng = random.Random(7)
    T = R.num_frames(L)
    augs = [R.spec_augment_draw(rng, 64, T) for _ in range(2)]
    aug = torch.tensor(augs, dtype=torch.int32)
    # dB only (normalize=0), no aug
    bft, _, _, _ = ops.mel(wave.to(dev), None, None, None, False)
    ref_db = torch.stack([R.parse_wave(wave[b:b + 1], None, None, normalize=False)[0] for b in range(2)])
    assert (bft.cpu() - ref_db).abs().max() < 2e-3          # dB units (values ~ -60..0)
    # full chain with SpecAugment zeros applied before the statistics
    bft2, _, _, _ = ops.mel(wave.to(dev), None, None, aug.to(dev), True)
    for b in range(2):
        ref = R.parse_wave(wave[b:b + 1], None, augs[b])
        assert max_rel(bft2[b].cpu(), ref[0]) < 1e-4


def test_layout_roundtrip_and_mask_lengths(dev):
    from lightning_asr_amd import ops
    x = torch.randn(3, 70, 45)
    y = ops.bct_to_btc(x.to(dev))
    assert torch.equal(y.cpu(), x.transpose(1, 2).contiguous())
    assert torch.equal(ops.btc_to_bct(y).cpu(), x)
    yb = ops.bct_to_btc(x.to(dev), torch.bfloat16)
    assert torch.equal(yb.cpu(), x.transpose(1, 2).contiguous().bfloat16())
    gold = np.load("tests/golden/mask_lengths.npz")
    for T, p, l in zip(gold["T"], gold["pct"], gold["lens"]):
        got = ops.mask_lengths(torch.tensor([p], dtype=torch.float32, device=dev), int(T))
        assert int(got[0]) == int(l)


# ----------------------------------------------------------------------------------------- depthwise conv
@pytest.mark.parametrize("C,k,stride,T", [(64, 33, 2, 201), (256, 39, 1, 101), (512, 75, 1, 260), (336, 51, 1, 77), (512, 87, 1, 130)])
def test_dwconv_fwd_bwd(dev, C, k, stride, T):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C + k)
    B = 3
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    w = (torch.randn(C, 1, k, generator=g) / math.sqrt(k)).requires_grad_(True)
    y = F.conv1d(x, w, None, stride, k // 2, 1, C)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg = x.detach().transpose(1, 2).contiguous().to(dev)
    got = ops.dwconv(xg, w.detach().to(dev), stride)
    assert max_rel(got.transpose(1, 2), y.detach()) < 2e-6
    dyg = dy.transpose(1, 2).contiguous().to(dev)
    dw = ops.dwconv_wgrad(xg, dyg, k, stride)
    assert max_rel(dw, w.grad[:, 0]) < 2e-5
    if stride == 1:
        add = torch.randn(B, T, C, generator=g)
        dx = ops.dwconv(dyg, w.detach().to(dev), 1, flip=True, addend=add.to(dev))
        assert max_rel(dx.cpu() - add, x.grad.transpose(1, 2)) < 2e-6


def test_dwconv_bf16(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 256, 90, generator=g).bfloat16()
    w = torch.randn(256, 1, 33, generator=g) / 6
    ref = F.conv1d(x.float(), w, None, 1, 16, 1, 256)
    got = ops.dwconv(x.transpose(1, 2).contiguous().to(dev), w.to(dev))
    assert got.dtype == torch.bfloat16
    assert max_rel(got.float().transpose(1, 2), ref) < 1e-2


@pytest.mark.parametrize("C,k,T", [(256, 33, 501), (512, 75, 260), (336, 51, 77), (512, 87, 130), (64, 1, 40), (72, 5, 300),
                                   (64, 63, 801), (128, 39, 1300), (64, 101, 513)])
def test_dwconv_bf16_dot2_exact_taps(dev, C, k, T):
    """bf16 stride-1 kernels (Toeplitz MFMA form; tap pairs on v_dot2c_f32_bf16 with LASR_DWCONV_DOT2=1): activations
    and taps are bf16, products exact in f32, so against an f64 convolution of the same bf16 operands only the f32
    accumulation order and the final bf16 rounding differ.  Forward, and the flipped form with an addend (data
    gradient); T > 512 runs several time tiles, k = 101 is the widest window the MFMA form takes."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C * 3 + k)
    B = 3
    x = torch.randn(B, C, T, generator=g).bfloat16()
    w = (torch.randn(C, 1, k, generator=g) / math.sqrt(k))
    wq = w.bfloat16().double()
    ref = F.conv1d(x.double(), wq, None, 1, k // 2, 1, C)
    xg = x.transpose(1, 2).contiguous().to(dev)
    got = ops.dwconv(xg, w.to(dev))
    assert got.dtype == torch.bfloat16
    assert max_rel(got.double().cpu().transpose(1, 2), ref) < 4e-3          # one bf16 rounding (2^-9) of the result
    add = torch.randn(B, T, C, generator=g).bfloat16()
    ref_f = F.conv1d(x.double(), wq.flip(2), None, 1, k // 2, 1, C) + add.double().transpose(1, 2)
    got_f = ops.dwconv(xg, w.to(dev), 1, flip=True, addend=add.to(dev))
    assert max_rel(got_f.double().cpu().transpose(1, 2), ref_f) < 4e-3


# ----------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tA,tB", [(300, 256, 64, 0, 0), (257, 130, 100, 0, 1), (96, 200, 515, 1, 1), (128, 28, 1024, 1, 0),
                                         (1000, 28, 1024, 0, 0), (70, 1024, 28, 0, 1), (33, 50, 4334, 0, 0), (200, 4334, 64, 0, 0)])
def test_gemm_layouts(dev, M, N, K, tA, tB):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    Bm = torch.randn((K, N) if tB else (N, K), generator=g)
    ref = (A.t() if tA else A).double() @ (Bm if tB else Bm.t()).double()
    got, _ = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, tA, tB)
    assert max_rel(got, ref) < 2e-6
    got2, _ = ops.gemm(A.to(dev), Bm.to(dev), M, N, K, tA, tB, split_k=4)
    assert max_rel(got2, ref) < 2e-6


def test_gemm_epilogue_bias_addend_mask_stats(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(11)
    B, T, K, N = 3, 50, 96, 200
    M = B * T
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    add = torch.randn(M, N, generator=g)
    lens = torch.tensor([50, 31, 0], dtype=torch.int32)
    ref = A @ W.t() + bias + add
    keep = (torch.arange(T).view(1, T) < lens.view(B, 1)).view(M, 1)
    ref = ref * keep
    got, stats = ops.gemm(A.to(dev), W.to(dev), M, N, K, bias=bias.to(dev), addend=add.to(dev), row_lens=lens.to(dev),
                          rows_per_seq=T, want_stats=True)
    assert max_rel(got, ref) < 2e-6
    assert max_rel(stats[:N], ref.sum(0)) < 1e-5
    assert max_rel(stats[N:], (ref * ref).sum(0)) < 1e-5


def test_gemm_bf16_inputs(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(12)
    M, N, K = 260, 136, 256
    A = torch.randn(M, K, generator=g).bfloat16()
    W = torch.randn(N, K, generator=g).bfloat16()
    ref = A.double() @ W.double().t()
    got, stats = ops.gemm(A.to(dev), W.to(dev), M, N, K, want_stats=True)
    assert got.dtype == torch.bfloat16
    assert max_rel(got.float(), ref) < 6e-3
    assert max_rel(stats[:N], got.float().sum(0)) < 1e-4
    got32, _ = ops.gemm(A.to(dev), W.to(dev), M, N, K, out_dtype=torch.float32)
    assert max_rel(got32, ref) < 1e-5


# ----------------------------------------------------------------------------------------- BN + residual + act
@pytest.mark.parametrize("C,has_res,act,masked", [(256, True, "relu", True), (512, False, "relu", False), (1024, False, "relu", False),
                                                  (336, True, "swish", True), (64, True, "none", False)])
def test_bn_act_fwd_bwd(dev, C, has_res, act, masked):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C)
    B, T = 3, 37
    y = torch.randn(B, C, T, generator=g)
    lens = torch.tensor([37, 20, 5], dtype=torch.int32)
    keep = (torch.arange(T).view(1, 1, T) < lens.view(B, 1, 1)).float()
    if masked:
        y = y * keep
    y = y.requires_grad_(True)
    y2 = torch.randn(B, C, T, generator=g, requires_grad=True) if has_res else None
    gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True), (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    gam2, bet2 = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True), (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    rm2, rv2 = torch.zeros(C), torch.ones(C)
    ym = y * keep if masked else y          # mask is part of the graph: its zeros block the gradient
    z = F.batch_norm(ym, rm, rv, gam, bet, True, 0.1, 1e-3)
    if has_res:
        z = z + F.batch_norm(y2, rm2, rv2, gam2, bet2, True, 0.1, 1e-3)
    out = {"relu": F.relu, "swish": lambda v: v * torch.sigmoid(v), "none": lambda v: v}[act](z)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)

    def cl(t):
        return t.detach().transpose(1, 2).contiguous().to(dev)
    yg, y2g = cl(ym), (cl(y2) if has_res else None)
    N = B * T

    def stats_of(t):
        f = t.reshape(N, C).double()
        return torch.cat([f.sum(0), (f * f).sum(0)]).float()
    grm, grv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    coef, saved = ops.bn_finalize(stats_of(yg), gam.detach().to(dev), bet.detach().to(dev), grm, grv, N)
    assert max_rel(grm, rm) < 1e-5 and max_rel(grv, rv) < 1e-5
    coef2 = saved2 = None
    if has_res:
        coef2, saved2 = ops.bn_finalize(stats_of(y2g), gam2.detach().to(dev), bet2.detach().to(dev), torch.zeros(C, device=dev),
                                        torch.ones(C, device=dev), N)
    got = ops.bn_act(yg, coef, y2g, coef2, None, act)
    assert max_rel(got.transpose(1, 2), out.detach()) < 5e-6
    dy, dy2, dg, db, dg2, db2 = ops.bn_act_bwd(cl(dout), yg, coef, saved, gam.detach().to(dev), y2g, coef2, saved2,
                                               gam2.detach().to(dev) if has_res else None, row_lens=lens.to(dev) if masked else None,
                                               act=act)
    assert max_rel(dy.transpose(1, 2), y.grad) < 2e-5
    assert max_rel(dg, gam.grad) < 2e-5 and max_rel(db, bet.grad) < 2e-5
    if has_res:
        assert max_rel(dy2.transpose(1, 2), y2.grad) < 2e-5
        assert max_rel(dg2, gam2.grad) < 2e-5 and max_rel(db2, bet2.grad) < 2e-5
    # fused hand-over (pass 2 reduces pass 1's partial sums itself): bit-identical to the two-launch path
    fz = ops.bn_act_bwd(cl(dout), yg, coef, saved, gam.detach().to(dev), y2g, coef2, saved2,
                        gam2.detach().to(dev) if has_res else None, row_lens=lens.to(dev) if masked else None, act=act, fused=True)
    for a_, b_ in zip(fz, (dy, dy2, dg, db, dg2, db2)):
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)
    # eval mode: coefficients from the running statistics
    coef_e, _ = ops.bn_finalize(None, gam.detach().to(dev), bet.detach().to(dev), grm, grv, N, training=False)
    ref_e = F.batch_norm(ym.detach(), grm.cpu(), grv.cpu(), gam.detach(), bet.detach(), False, 0.1, 1e-3)
    got_e = ops.bn_act(yg, coef_e, None, None, None, "none")
    assert max_rel(got_e.transpose(1, 2), ref_e) < 5e-6


# ----------------------------------------------------------------------------------------- log_softmax / CTC / decode
@pytest.mark.parametrize("C", [28, 4334])
def test_log_softmax_argmax(dev, C):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C)
    x = torch.randn(5, 41, C, generator=g) * 3
    x[0, 0, 3] = x[0, 0, 7] = 50.0          # exact tie -> lowest index
    lp, am = ops.log_softmax(x.to(dev))
    ref = F.log_softmax(x, -1)
    assert (lp.cpu() - ref).abs().max() < 2e-6
    assert torch.equal(am.cpu().long(), ref.argmax(-1))
    assert int(am[0, 0]) == 3


def _ctc_case(B, T, C, S, seed, repeats=False):
    g = torch.Generator().manual_seed(seed)
    lp = F.log_softmax(torch.randn(B, T, C, generator=g) * 2, -1)
    tg = torch.randint(0, C - 1, (B, S), generator=g)
    if repeats:
        tg[:, 1::2] = tg[:, 0::2][:, :tg[:, 1::2].shape[1]]
    il = torch.randint(max(2 * S + 1, T // 2), T + 1, (B,), generator=g, dtype=torch.int32)
    il[0] = T
    tl = torch.randint(1, S + 1, (B,), generator=g, dtype=torch.int32)
    tl[0] = S
    return lp, tg, il, tl


@pytest.mark.parametrize("B,T,C,S,rep", [(3, 20, 5, 4, True), (4, 101, 28, 12, False), (2, 60, 28, 25, True), (2, 120, 4334, 45, False),
                                         (2, 501, 28, 100, True), (1, 300, 28, 140, False), (1, 600, 40, 290, False)])
def test_ctc_matches_torch(dev, B, T, C, S, rep):
    from lightning_asr_amd import ops
    lp, tg, il, tl = _ctc_case(B, T, C, S, B * 1000 + T, rep)
    lpr = lp.clone().requires_grad_(True)
    ref = F.ctc_loss(lpr.transpose(0, 1), tg, il, tl, blank=C - 1, reduction="none")
    gs = torch.rand(B) + 0.5
    (ref * gs).sum().backward()
    nll, grad = ops.ctc_loss(lp.to(dev), tg.to(dev), il.to(dev), tl.to(dev), C - 1, True, gs.to(dev))
    assert torch.isfinite(ref).all()
    # north_star: fp32 CTC loss within 1e-4 relative of the CPU reference
    assert ((nll.cpu() - ref.detach()).abs() / ref.detach().abs()).max() < 1e-4
    gref = lpr.grad
    assert (grad.cpu() - gref).abs().max() < 2e-3 * gref.abs().max() + 1e-6
    assert rel_l2(grad, gref) < 2e-3
    # default scale = 1/B (batch mean, train.py:77)
    _, grad_m = ops.ctc_loss(lp.to(dev), tg.to(dev), il.to(dev), tl.to(dev), C - 1, True, None)
    lpr2 = lp.clone().requires_grad_(True)
    F.ctc_loss(lpr2.transpose(0, 1), tg, il, tl, blank=C - 1, reduction="none").mean().backward()
    assert rel_l2(grad_m, lpr2.grad) < 2e-3


def test_ctc_small_vs_numpy_and_edge_cases(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(0)
    B, T, C = 4, 9, 5
    lp = F.log_softmax(torch.randn(B, T, C, generator=g), -1)
    tg = torch.tensor([[0, 0, 1], [1, 2, 0], [3, 3, 3], [2, 0, 0]])
    il = torch.tensor([9, 6, 3, 0], dtype=torch.int32)       # sample 2 infeasible (needs 5 frames), sample 3 empty input
    tl = torch.tensor([3, 2, 3, 0], dtype=torch.int32)
    nll, grad = ops.ctc_loss(lp.to(dev), tg.to(dev), il.to(dev), tl.to(dev), C - 1, True, torch.ones(B, device=dev))
    nll, grad = nll.cpu(), grad.cpu()
    for b in (0, 1):
        n_ref, g_ref = R.ctc_numpy(lp[b, :il[b]].numpy(), tg[b, :tl[b]].tolist(), C - 1)
        assert abs(nll[b].item() - n_ref) < 1e-4 * abs(n_ref)
        # torch convention: grad = exp(lp) - occupancy; ctc_numpy returns the true derivative -occupancy
        full = np.exp(lp[b, :il[b]].numpy().astype(np.float64)) + g_ref
        assert np.abs(grad[b, :il[b]].numpy() - full).max() < 1e-4
        assert torch.all(grad[b, il[b]:] == 0)
    assert math.isinf(nll[2].item()) and nll[2] > 0           # zero_infinity=False (train.py:196)
    assert torch.isnan(grad[2, :3]).all() and torch.all(grad[2, 3:] == 0)
    assert nll[3].item() == 0.0 and torch.all(grad[3] == 0)


def test_greedy_decode(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(4)
    B, T, blank = 5, 300, 27
    ids = torch.randint(0, 28, (B, T), generator=g, dtype=torch.int32)
    ids[0, :40] = 3
    ids[1] = blank
    lens = torch.tensor([300, 300, 150, 1, 0], dtype=torch.int32)
    tok, n = ops.greedy_decode(ids.to(dev), lens.to(dev), blank)
    for b in range(B):
        ref = R.greedy_collapse(ids[b, :lens[b]].tolist(), blank)
        assert int(n[b]) == len(ref)
        assert tok[b, :len(ref)].cpu().tolist() == ref
    tok2, n2 = ops.greedy_decode(ids.to(dev), None, blank)
    assert tok2[0, :int(n2[0])].cpu().tolist() == R.greedy_collapse(ids[0].tolist(), blank)


# ----------------------------------------------------------------------------------------- optimiser
def test_novograd_matches_oracle(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(9)
    shapes = [(256, 1, 33), (256, 64, 1), (256,), (3,), (1024, 512, 1), (28,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    offs = np.cumsum([0] + [p.numel() for p in ps])
    flat = torch.cat([p.flatten() for p in ps]).to(dev)
    m = torch.zeros_like(flat)
    v = torch.zeros(len(ps), device=dev)
    offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
    lr = torch.tensor([0.01], device=dev)
    st = R.NovogradState(len(ps))
    ref_p = [p.clone() for p in ps]
    for step in range(3):
        gr = [torch.randn(s, generator=g) * (step + 1) for s in shapes]
        R.novograd_step(ref_p, [x.clone() for x in gr], st, 0.01, 0.8, 0.5, 1e-8, 1e-3)
        gflat = torch.cat([x.flatten() for x in gr]).to(dev) * 4.0
        ops.novograd_step(flat, gflat, m, v, offsets, lr, 0.8, 0.5, 1e-8, 1e-3, grad_scale=0.25)
    ref_flat = torch.cat([p.flatten() for p in ref_p])
    assert max_rel(flat, ref_flat) < 2e-6
    assert max_rel(v, torch.stack([x for x in st.exp_avg_sq])) < 5e-5     # oracle sums ||g||^2 in f32
    # the kept-workspace form (lasr_novograd_step_keep: zeroed once by the caller, left zeroed by every step, no memset launch):
    # the same three steps land on the same bits as the form that zeroes per call
    g2 = torch.Generator().manual_seed(9)
    ps2 = [torch.randn(s, generator=g2) for s in shapes]
    flat2 = torch.cat([p.flatten() for p in ps2]).to(dev)
    m2, v2 = torch.zeros_like(flat2), torch.zeros(len(ps2), device=dev)
    ws = ops.novograd_workspace(len(ps2), flat2.numel(), dev)
    for step in range(3):
        gr = [torch.randn(s, generator=g2) * (step + 1) for s in shapes]
        gflat = torch.cat([x.flatten() for x in gr]).to(dev) * 4.0
        ops.novograd_step(flat2, gflat, m2, v2, offsets, lr, 0.8, 0.5, 1e-8, 1e-3, grad_scale=0.25, ws=ws)
    assert torch.equal(flat2, flat) and torch.equal(v2, v) and torch.equal(m2, m)
    assert int(ws.view(torch.int32).abs().sum().item()) == 0 or bool((ws[256 * ((4 * len(ps2) + 255) // 256):] == 0).all())   # norm accumulators left zeroed


# ----------------------------------------------------------------------------------------- bf16 MFMA GEMM
@pytest.mark.parametrize("M,N,K,tA,tB", [(300, 256, 64, 0, 0), (257, 130, 100, 0, 1), (96, 200, 515, 1, 1), (128, 28, 1024, 1, 0),
                                         (1000, 28, 1024, 0, 0), (70, 1024, 28, 0, 1), (512, 512, 1603, 1, 1), (640, 336, 512, 0, 1)])
def test_gemm_bf16_mfma_exact_integers(dev, M, N, K, tA, tB):
    """Small-integer operands: every product and partial sum is exact in bf16 x bf16 -> f32, so the
    bf16-MFMA kernel (b128 and transposed ds_read_b64_tr_b16 fragments) must match bit for bit."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randint(-4, 5, (K, M) if tA else (M, K), generator=g).float()
    Bm = torch.randint(-4, 5, (K, N) if tB else (N, K), generator=g).float()
    ref = (A.t() if tA else A).double() @ (Bm if tB else Bm.t()).double()
    got, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, tA, tB, out_dtype=torch.float32)
    assert torch.equal(got.cpu().double(), ref)
    got2, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, tA, tB, out_dtype=torch.float32, split_k=5)
    assert torch.equal(got2.cpu().double(), ref)


def test_gemm_bf16_mfma_epilogue(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(21)
    B, T, K, N = 3, 77, 256, 336
    M = B * T
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / 16).bfloat16()
    bias = torch.randn(N, generator=g)
    lens = torch.tensor([77, 40, 0], dtype=torch.int32)
    keep = (torch.arange(T).view(1, T) < lens.view(B, 1)).view(M, 1)
    ref = ((A.double() @ W.double().t() + bias.double()) * keep).float()
    got, stats = ops.gemm(A.to(dev), W.to(dev), M, N, K, bias=bias.to(dev), row_lens=lens.to(dev), rows_per_seq=T, want_stats=True)
    assert got.dtype == torch.bfloat16
    assert torch.equal(got.cpu(), ref.bfloat16()) or max_rel(got.float(), ref) < 4e-3
    gf = got.float().cpu().double()
    assert max_rel(stats[:N], gf.sum(0)) < 1e-5           # statistics of the values as stored
    assert max_rel(stats[N:], (gf * gf).sum(0)) < 1e-5
    assert torch.all(got[T + 40:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,tB", [(16032, 256, 160, True), (300, 256, 64, False), (257, 132, 100, True), (130, 28, 512, False)])
def test_gemm_bf16_mfma_addend(dev, M, N, K, tB):
    """C = round_bf16(A B^T + bias + addend) in the bf16-MFMA kernel's epilogue (the accumulating data-gradient GEMMs of the context
    branch, models/QuartNetContext.py:171-173 backward: d(x) += dG W_ih).  Small integers: exact, so bit for bit; the in-place form
    (addend == C) is the one the model uses."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (K, N) if tB else (N, K), generator=g).float()
    add = torch.randint(-8, 9, (M, N), generator=g).float()
    bias = torch.randint(-2, 3, (N,), generator=g).float()
    ref = A.double() @ (Bm if tB else Bm.t()).double() + add.double() + bias.double()
    got, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, False, tB, bias=bias.to(dev), addend=add.bfloat16().to(dev))
    assert got.dtype == torch.bfloat16
    assert torch.equal(got.float().cpu().double(), ref.float().bfloat16().double())
    # in place
    Cm = add.bfloat16().to(dev)
    nb = ops._lib.load().lasr_gemm_workspace_bytes(M, N, 1, 0)
    ws = ops._ws(nb, dev)
    Ag, Bg = A.bfloat16().to(dev), Bm.bfloat16().to(dev)
    ops.call("lasr_gemm", ops._p(Ag), ops._p(Bg), ops._p(Cm), ops.BF16, ops.BF16, M, N, K, 0, int(tB), None, ops._p(Cm), None, 0, None, 1,
             ops._p(ws), nb, ops._stream())
    assert torch.equal(Cm.float().cpu().double(), (ref - bias.double()).float().bfloat16().double())


# ----------------------------------------------------------------------------------------- 256x256-tile bf16 GEMM
@pytest.mark.parametrize("M,N,K,tA,tB", [(16032, 512, 512, False, False), (16032, 512, 512, False, True),
                                         (8100, 768, 320, False, False), (8100, 776, 200, False, True),
                                         (8192, 1024, 256, True, False), (7000, 520, 136, True, True)])
def test_gemm_bf16_big_tile_exact_integers(dev, M, N, K, tA, tB):
    """Shapes large enough for the 256x256x64 one-workgroup-per-CU kernel (>= 120 tiles), with ragged M/N/K
    edges.  Small-integer operands make bf16 x bf16 -> f32 exact, and the bf16 result is exact while
    |value| <= 256, so the comparison is bit for bit."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randint(-2, 3, (K, M) if tA else (M, K), generator=g).float()
    Bm = (torch.rand((K, N) if tB else (N, K), generator=g) < 0.05).float()
    ref = (A.t() if tA else A) @ (Bm if tB else Bm.t())
    assert ref.abs().max() <= 256
    got, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, tA, tB)
    assert got.dtype == torch.bfloat16
    assert torch.equal(got.cpu().float(), ref)


def test_gemm_bf16_big_tile_epilogue(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(33)
    B, T, K, N = 32, 501, 512, 512
    M = B * T
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / 16).bfloat16()
    bias = torch.randn(N, generator=g)
    lens = torch.randint(0, T + 1, (B,), generator=g).to(torch.int32)
    lens[0], lens[1] = T, 0
    keep = (torch.arange(T).view(1, T) < lens.view(B, 1)).view(M, 1)
    ref = ((A.double() @ W.double().t() + bias.double()) * keep).float()
    got, stats = ops.gemm(A.to(dev), W.to(dev), M, N, K, bias=bias.to(dev), row_lens=lens.to(dev), rows_per_seq=T, want_stats=True)
    assert max_rel(got.float(), ref) < 4e-3
    gf = got.float().cpu().double()
    assert max_rel(stats[:N], gf.sum(0)) < 1e-5
    assert max_rel(stats[N:], (gf * gf).sum(0)) < 1e-5
    assert torch.all(got.view(B, T, N)[1] == 0)


@pytest.mark.parametrize("dtype,M,T", [(torch.bfloat16, 16032, 501), (torch.float32, 600, 200), (torch.bfloat16, 777, 259)])
def test_gemm_bn_fused_matches_unfused(dev, dtype, M, T):
    """lasr_gemm_batch_partials + lasr_bn_finalize_partials (one reduce+finalize launch for the main and the
    residual branch) against lasr_gemm(stats) + lasr_bn_finalize per branch: same sums, same coefficients."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(M)
    K, N, B = 256, 512, M // T
    xs = [torch.randn(M, K, generator=g).to(dtype).to(dev) for _ in range(2)]
    ws_ = [(torch.randn(N, K, generator=g) / 16).to(dtype).to(dev) for _ in range(2)]
    lens = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32).to(dev)
    mk = lambda: (torch.rand(N, generator=g).to(dev) + 0.5, torch.randn(N, generator=g).to(dev), torch.zeros(N, device=dev), torch.ones(N, device=dev))
    bns, bns_ref = [mk(), mk()], None
    bns_ref = [(a, b, torch.zeros(N, device=dev), torch.ones(N, device=dev)) for (a, b, _, _) in bns]
    ys, coefs, saveds, stats = ops.gemm_bn_fused(xs, ws_, bns, row_lens=lens, rows_per_seq=T)
    for i in range(2):
        y_ref, st_ref = ops.gemm(xs[i], ws_[i], M, N, K, row_lens=lens if i == 0 else None, rows_per_seq=T if i == 0 else 0,
                                 want_stats=True)
        coef_ref, saved_ref = ops.bn_finalize(st_ref, bns_ref[i][0], bns_ref[i][1], bns_ref[i][2], bns_ref[i][3], M)
        assert torch.equal(ys[i], y_ref)
        assert torch.equal(stats[i], st_ref)
        assert torch.equal(coefs[i], coef_ref) and torch.equal(saveds[i], saved_ref)
        assert torch.equal(bns[i][2], bns_ref[i][2]) and torch.equal(bns[i][3], bns_ref[i][3])


def test_gemm_ld_padded_narrow_operand(dev):
    """The decoder's backward GEMMs: a 28-column matrix kept in a 32-pitch bf16 buffer whose pad columns hold
    NaN must give exact results (pad elements are masked / only feed rows that are never stored), on the
    aligned 16-byte operand path, with K = 28 (not a multiple of 8) and M = 28."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(77)
    rows, C, H = 16032, 28, 1024
    gl = torch.randint(-2, 3, (rows, C), generator=g).float()
    glp = torch.full((rows, 32), float("nan"))
    glp[:, :C] = gl
    W = (torch.rand(C, H, generator=g) < 0.1).float()
    h = torch.randint(-1, 2, (rows, H), generator=g).float()
    glp_d, W_d, h_d = glp.bfloat16().to(dev), W.bfloat16().to(dev), h.bfloat16().to(dev)
    # dh = gl W : A K-contiguous with a ragged K and a NaN tail, B row-contiguous
    dh = ops.gemm_ld(glp_d, 32, W_d, H, rows, H, C, transA=False, transB=True)
    assert torch.equal(dh.float().cpu(), gl @ W)
    # dW = gl^T h : A row-contiguous with 28 of 32 columns valid, split-K, f32 result
    dW = ops.gemm_ld(glp_d, 32, h_d, H, C, H, rows, transA=True, transB=True, split_k=16, out_dtype=torch.float32)
    assert torch.equal(dW.cpu(), gl.t() @ h)


@pytest.mark.parametrize("C,k,T,B", [(512, 63, 501, 4), (256, 33, 501, 3), (336, 51, 77, 3), (512, 87, 130, 2), (64, 1, 40, 2),
                                     (72, 5, 300, 2), (64, 75, 801, 2), (128, 101, 513, 2)])
def test_dwconv_wgrad_bf16_mfma(dev, C, k, T, B):
    """bf16 stride-1 depthwise weight gradient (Toeplitz / strided-window MFMA form; LASR_DWWGRAD_VALU=1 selects the
    VALU form): bf16 operands, exact products, f32 accumulation over B*T terms, against an f64 correlation."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C + 7 * k + T)
    x = torch.randn(B, T, C, generator=g).bfloat16()
    dy = torch.randn(B, T, C, generator=g).bfloat16()
    xd, dyd = x.double(), dy.double()
    pad = k // 2
    xp = torch.nn.functional.pad(xd, (0, 0, pad, pad))                       # (B, T + 2 pad, C)
    ref = torch.stack([(dyd * xp[:, j:j + T, :]).sum(dim=(0, 1)) for j in range(k)], dim=1)   # (C, k)
    got = ops.dwconv_wgrad(x.to(dev), dy.to(dev), k, 1)
    scale = ref.abs().max()
    assert (got.cpu().double() - ref).abs().max() < 2e-5 * scale + 1e-6


def test_reduce_many_segments(dev):
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 512 * 512), (1, 7), (32, 512 * 75), (5, 1000), (64, 33), (3, 2048), (7, 4096), (6, 100), (2, 1028), (4, 12),
              (3, 512 * 512), (3, 1024 * 5 + 17), (4, 1023), (2, 1024), (1, 4097)]
    segs, refs = [], []
    for (P, n) in shapes:
        pt = torch.randn(P, n, generator=g)
        segs.append((pt.to(dev), torch.empty(n, device=dev)))
        refs.append(pt.double().sum(0))
    ops.reduce_many(segs)
    for (pt, out), ref in zip(segs, refs):
        assert max_rel(out, ref) < 1e-6
        P = pt.shape[0]
        if P <= 4:     # split-K slabs (round 5: four elements per thread): the fixed-order f64 sum, bit for bit
            v = [pt[u].cpu().double() if u < P else torch.zeros(pt.shape[1], dtype=torch.float64) for u in range(4)]
            assert torch.equal(out.cpu(), ((v[0] + v[1]) + (v[2] + v[3])).float())


def test_wgrad_multi_exact_integers(dev):
    """A backward stage's 1x1 weight gradients in one split-K launch: mixed shapes (256- and 512-wide, 256 -> 512),
    small-integer operands so every product and partial sum is exact in f32, result compared bit for bit."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(9)
    rows = 16032
    shapes = [(512, 512), (512, 512), (256, 256), (512, 256), (256, 256), (1024, 512), (160, 256), (256, 64), (512, 336), (264, 520)]
    dys = [torch.randint(-2, 3, (rows, co), generator=g).float() for co, _ in shapes]
    xs = [(torch.rand(rows, ci, generator=g) < 0.05).float() * torch.randint(-1, 2, (rows, ci), generator=g).float() for _, ci in shapes]
    for split in (1, 3, 6, 16):    # the cap; the library picks the slice count for its tile form (256 x 256 here)
        outs = ops.wgrad_multi([d.bfloat16().to(dev) for d in dys], [x.bfloat16().to(dev) for x in xs], split_k=split)
        for d, x, o in zip(dys, xs, outs):
            assert torch.equal(o.cpu(), d.t() @ x)


def test_gemm_bf16_random_shapes_exact(dev):
    """Seeded sweep over ragged shapes and all four operand layouts: small tiles, 256x256 and 256x128 one-per-CU tiles,
    aligned (16-byte) and unaligned pitches, K tails that are not multiples of 8 or 64.  Small-integer operands make
    every product and partial sum exact, so bf16 results (|value| <= 256) must match bit for bit."""
    from lightning_asr_amd import ops
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(2024)
    for case in range(28):
        big = case % 2 == 0
        M = int(rng.integers(15000, 40000)) if big else int(rng.integers(1, 3000))
        N = int(rng.integers(1, 80)) * 8 if case % 4 != 3 else int(rng.integers(1, 600))
        K = int(rng.integers(1, 90)) * 8 if case % 4 != 1 else int(rng.integers(1, 700))
        tA, tB = bool(case & 1), bool(case & 2)
        if big:
            tA = False                      # an [K][M] operand with M ~ 30 000 rows is not a shape of this model; keep the sweep fast
        A = torch.randint(-2, 3, (K, M) if tA else (M, K), generator=g).float()
        Bm = (torch.rand((K, N) if tB else (N, K), generator=g) < 0.06).float()
        ref = (A.t() if tA else A) @ (Bm if tB else Bm.t())
        assert ref.abs().max() <= 256
        got, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, tA, tB)
        assert torch.equal(got.cpu().float(), ref), (case, M, N, K, tA, tB)
        got32, _ = ops.gemm(A.bfloat16().to(dev), Bm.bfloat16().to(dev), M, N, K, tA, tB, out_dtype=torch.float32, split_k=3)
        assert torch.equal(got32.cpu(), ref), (case, M, N, K, tA, tB, "f32 split")


@pytest.mark.parametrize("C,k,T,B", [(512, 75, 501, 32), (64, 33, 2300, 2)])
def test_dwconv_bf16_mfma_tile_forms(dev, C, k, T, B):
    """The two time-tile forms of the stride-1 MFMA kernels: (512, 32 utterances) fills the chip with one 512-frame
    tile per workgroup and no time split of the weight gradient (the bench shape); (64, T = 2300) runs 256-frame half
    tiles with more tiles than gridDim.z, so workgroups loop over strided tiles."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C + k + T)
    x = torch.randn(B, C, T, generator=g).bfloat16()
    w = torch.randn(C, 1, k, generator=g) / math.sqrt(k)
    wq = w.bfloat16().double()
    ref = F.conv1d(x.double(), wq, None, 1, k // 2, 1, C)
    xg = x.transpose(1, 2).contiguous().to(dev)
    got = ops.dwconv(xg, w.to(dev))
    assert max_rel(got.double().cpu().transpose(1, 2), ref) < 4e-3
    dy = torch.randn(B, T, C, generator=g).bfloat16()
    xp = torch.nn.functional.pad(x.double().transpose(1, 2), (0, 0, k // 2, k // 2))
    dref = torch.stack([(dy.double() * xp[:, j:j + T, :]).sum(dim=(0, 1)) for j in range(k)], dim=1)
    dw = ops.dwconv_wgrad(xg, dy.to(dev), k, 1)
    assert (dw.cpu().double() - dref).abs().max() < 2e-5 * dref.abs().max() + 1e-6


@pytest.mark.parametrize("C,k,T,B,res", [(512, 63, 501, 32, True), (256, 33, 501, 8, True), (336, 51, 300, 3, False), (64, 5, 40, 2, True)])
def test_dwconv_bwd_fused_equals_separate_launches(dev, C, k, T, B, res):
    """lasr_dwconv_bwd_fused (weight-gradient and data-gradient workgroups in one grid) against the two separate launches
    on the same bf16 inputs: same kernels' bodies, so dx must be bit-identical and dW equal up to the order of the final f64 sum."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(C + k + T)
    x = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    dy = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    w = (torch.randn(C, 1, k, generator=g) / math.sqrt(k)).to(dev)
    add = torch.randn(B, T, C, generator=g).bfloat16().to(dev) if res else None
    dw_ref = ops.dwconv_wgrad(x, dy, k, 1)
    dx_ref = ops.dwconv(dy, w, 1, flip=True, addend=add)
    dw, dx = ops.dwconv_bwd_fused(x, dy, w, add)
    assert torch.equal(dx, dx_ref)
    assert max_rel(dw, dw_ref) < 1e-6            # same partials, summed by lasr_reduce_many instead of the kernel's own tail


def test_ctc_random_sweep_vs_torch(dev):
    """Seeded sweep of lattice shapes through both emission paths (LDS rows when C % 4 == 0, register ring otherwise):
    ragged input / target lengths, repeated labels, empty targets, tight alignments (T = S + repeats), blank-dominated
    emissions (the finite dead-state sentinel must behave like -inf)."""
    from lightning_asr_amd import ops
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    for case in range(24):
        C = int(rng.choice([5, 8, 28, 29, 40]))
        B = int(rng.integers(1, 6))
        S = int(rng.integers(0, 60))
        T = int(rng.integers(max(2 * S + 2, 4), 2 * S + 80))
        logits = torch.randn(B, T, C, generator=g) * float(rng.choice([1.0, 4.0]))
        if case % 3 == 0:
            logits[..., C - 1] += 12.0                       # blank-dominated: label paths are ~e^-12 per frame
        lp = F.log_softmax(logits, -1)
        tg = torch.randint(0, C - 1, (B, max(S, 1)), generator=g)
        if case % 2 == 0 and S > 1:
            tg[:, 1::2] = tg[:, 0:-1:2][:, :tg[:, 1::2].shape[1]]     # adjacent repeats
        tl = torch.randint(0, S + 1, (B,), generator=g, dtype=torch.int32)
        tl[0] = S
        reps = [(int((tg[b, 1:tl[b]] == tg[b, :max(tl[b] - 1, 0)]).sum()) if tl[b] > 1 else 0) for b in range(B)]
        il = torch.tensor([int(rng.integers(int(tl[b]) + reps[b], T + 1)) for b in range(B)], dtype=torch.int32)
        if case % 4 == 1:
            il[0] = int(tl[0]) + reps[0]                     # tightest feasible alignment
        il = il.clamp(min=1)
        lpr = lp.clone().requires_grad_(True)
        ref = F.ctc_loss(lpr.transpose(0, 1), tg, il.long(), tl.long(), blank=C - 1, reduction="none")
        assert torch.isfinite(ref).all(), case
        ref.sum().backward()
        nll, grad = ops.ctc_loss(lp.to(dev), tg.to(dev), il.to(dev), tl.to(dev), C - 1, True, torch.ones(B, device=dev))
        assert ((nll.cpu() - ref.detach()).abs() <= 1e-4 * ref.detach().abs() + 1e-5).all(), (case, nll.cpu(), ref)
        gref = lpr.grad
        assert (grad.cpu() - gref).abs().max() < 2e-3 * gref.abs().max() + 1e-6, case


@pytest.mark.parametrize("B,T,ci,co,act", [(32, 501, 256, 256, "relu"), (5, 77, 512, 512, "relu"), (3, 130, 256, 512, "swish"), (2, 40, 64, 72, "none")])
def test_folded_eval_unit_gemm(dev, B, T, ci, co, act):
    """Eval-mode BN folded into a residual unit's two 1x1 convs (lasr_fold_bn_weights_many + lasr_gemm_dual):
    act([mask(u) | x] . [a W | a2 Wr]^T + b + b2) against the same expression in f64 from the folded bf16 weights, and
    the folded weights against a*W in f32."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(ci + co + T)
    M = B * T
    u = torch.randn(M, ci, generator=g).bfloat16()
    x = torch.randn(M, ci, generator=g).bfloat16()
    W, Wr = torch.randn(co, ci, generator=g) / math.sqrt(ci), torch.randn(co, ci, generator=g) / math.sqrt(ci)
    coef = torch.cat([1 + 0.2 * torch.randn(co, generator=g), 0.3 * torch.randn(co, generator=g)])
    coef2 = torch.cat([1 + 0.2 * torch.randn(co, generator=g), 0.3 * torch.randn(co, generator=g)])
    lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    wcat, bias = ops.fold_bn_weights(W.to(dev), coef.to(dev), Wr.to(dev), coef2.to(dev))
    ref_w = torch.cat([coef[:co, None] * W, coef2[:co, None] * Wr], 1)
    assert (wcat.cpu().float() - ref_w).abs().max() <= 4e-3 * ref_w.abs().max()
    assert torch.allclose(bias.cpu(), coef[co:] + coef2[co:], atol=1e-6)
    out = ops.gemm_dual(u.to(dev), x.to(dev), wcat, bias, lens.to(dev), T, act)
    keep = (torch.arange(T).view(1, T) < lens.view(B, 1)).reshape(M, 1).double()
    z = torch.cat([u.double() * keep, x.double()], 1) @ wcat.cpu().double().t() + bias.cpu().double()
    ref = {"relu": torch.relu, "swish": lambda v: v * torch.sigmoid(v), "none": lambda v: v}[act](z)
    assert (out.cpu().double() - ref).abs().max() <= 4e-3 * ref.abs().max() + 1e-6


def test_eval_forward_folded_matches_unfolded(dev, tmp_path):
    """lasr_model_forward(training=0) with the folded units (default) against LASR_NO_EVAL_FOLD=1 in a child process: the
    same bf16 network up to where the per-channel scale is rounded (into the weights, or after the bf16 pre-activation)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from lightning_asr_amd.engine import NativeModel\n"
        "g = torch.Generator().manual_seed(5)\n"
        "m = NativeModel('plain', 28, mask=True, act='relu', dtype=torch.bfloat16, device=torch.device('cuda'))\n"
        "m.init_parameters(seed=1)\n"
        "m.buffers.copy_((0.5 + torch.rand(m.buffers.shape, generator=g)).to(m.buffers.device))\n"
        "feats = torch.randn(3, 401, 64, generator=g).bfloat16().cuda(); pct = torch.tensor([1.0, 0.8, 0.35]).cuda()\n"
        "out = m.forward(feats, pct, training=False)\n"
        "torch.save((out[0] if isinstance(out, (tuple, list)) else out).float().cpu(), sys.argv[1])\n" % ROOT)
    outs = []
    for fold in (True, False):
        env = dict(os.environ)
        if not fold:
            env["LASR_NO_EVAL_FOLD"] = "1"
        path = str(tmp_path / ("logp_%d.pt" % fold))
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        outs.append(torch.load(path))
    assert torch.isfinite(outs[0]).all()
    assert (outs[0] - outs[1]).abs().max() < 0.08              # log-probabilities of a 15-unit bf16 network
    assert (outs[0].argmax(-1) == outs[1].argmax(-1)).float().mean() > 0.97


# ----------------------------------------------------------------------------------------- bf16 storage mode of the same kernels
@pytest.mark.parametrize("C,has_res,act,masked", [(256, True, "relu", True), (512, True, "relu", False), (1024, False, "relu", False),
                                                  (336, True, "swish", True)])
def test_bn_act_fwd_bwd_bf16(dev, C, has_res, act, masked):
    """bn_act_fwd / bn_act_bwd_stats / bn_act_bwd_apply on bf16 tensors (31 % of the bench step) against an f64 torch
    reference evaluated ON THE SAME bf16-rounded inputs: what may differ is one final bf16 rounding per output element."""
    from lightning_asr_amd import ops
    from oracle.ref_bf16 import rb
    g = torch.Generator().manual_seed(C + 1)
    B, T = 4, 123
    lens = torch.tensor([123, 77, 50, 9], dtype=torch.int32)
    keep = (torch.arange(T).view(1, 1, T) < lens.view(B, 1, 1)).double()
    y = rb(torch.randn(B, C, T, generator=g)).double()
    if masked:
        y = y * keep
    y.requires_grad_(True)
    y2 = rb(torch.randn(B, C, T, generator=g)).double().requires_grad_(True) if has_res else None
    mk = lambda base: (base + 0.1 * torch.randn(C, generator=g)).double().requires_grad_(True)   # noqa: E731
    gam, bet, gam2, bet2 = mk(1.0), mk(0.0), mk(1.0), mk(0.0)
    ym = y * keep if masked else y
    z = F.batch_norm(ym, torch.zeros(C).double(), torch.ones(C).double(), gam, bet, True, 0.1, 1e-3)
    if has_res:
        z = z + F.batch_norm(y2, torch.zeros(C).double(), torch.ones(C).double(), gam2, bet2, True, 0.1, 1e-3)
    out = {"relu": F.relu, "swish": lambda v: v * torch.sigmoid(v)}[act](z)
    dout = rb(torch.randn(out.shape, generator=g)).double()
    out.backward(dout)

    def cl(t):
        return t.detach().transpose(1, 2).contiguous().to(dev, torch.bfloat16)
    yg, y2g = cl(ym), (cl(y2) if has_res else None)
    N = B * T

    def stats_of(t):
        f = t.reshape(N, C).double()
        return torch.cat([f.sum(0), (f * f).sum(0)]).float()
    f32 = lambda t: t.detach().float().to(dev)   # noqa: E731
    coef, saved = ops.bn_finalize(stats_of(yg), f32(gam), f32(bet), torch.zeros(C, device=dev), torch.ones(C, device=dev), N)
    coef2 = saved2 = None
    if has_res:
        coef2, saved2 = ops.bn_finalize(stats_of(y2g), f32(gam2), f32(bet2), torch.zeros(C, device=dev), torch.ones(C, device=dev), N)
    got = ops.bn_act(yg, coef, y2g, coef2, None, act)
    assert got.dtype == torch.bfloat16
    ref_out = rb(out.detach().float())
    # identical except where the f32 evaluation lands on the other side of a bf16 rounding boundary (1 ulp = 2^-8 relative)
    assert rel_l2(got.transpose(1, 2).float(), ref_out) < 2e-4
    assert ((got.transpose(1, 2).float().cpu() - ref_out).abs() <= 2.0 ** -7 * ref_out.abs() + 1e-30).all()
    dy, dy2, dg, db, dg2, db2 = ops.bn_act_bwd(cl(dout), yg, coef, saved, f32(gam), y2g, coef2, saved2, f32(gam2) if has_res else None,
                                               row_lens=lens.to(dev) if masked else None, act=act)
    assert dy.dtype == torch.bfloat16
    assert rel_l2(dy.transpose(1, 2).float(), rb(y.grad.float())) < 3e-4
    assert max_rel(dg, gam.grad) < 2e-5 and max_rel(db, bet.grad) < 2e-5
    if has_res:
        assert rel_l2(dy2.transpose(1, 2).float(), rb(y2.grad.float())) < 3e-4
        assert max_rel(dg2, gam2.grad) < 2e-5 and max_rel(db2, bet2.grad) < 2e-5


def test_mel_bf16_output_is_the_rounded_f32_output(dev):
    """the channels-last bf16 feature tensor the model consumes == round-to-nearest-even of the f32 features (parity-checked
    against the oracle above), and the padded frames are exact zeros"""
    from lightning_asr_amd import ops
    from oracle.ref_bf16 import rb
    g = torch.Generator().manual_seed(9)
    L = 16000 * 3 + 77
    wave = 0.1 * torch.randn(3, L, generator=g)
    lens = torch.tensor([L, L - 5000, 9000], dtype=torch.int32)
    bft, btf, frames, pct = ops.mel(wave.to(dev), lens.to(dev), None, None, True, torch.bfloat16)
    assert btf.dtype == torch.bfloat16 and bft.dtype == torch.float32
    ref = rb(bft.cpu()).transpose(1, 2)
    got = btf.float().cpu()
    # both come from one f64 value: f64 -> f32 -> bf16 (reference here) against f64 -> bf16 in the kernel differ only on exact ties
    assert (got != ref).float().mean() < 1e-4
    assert ((got - ref).abs() <= 2.0 ** -7 * ref.abs()).all()
    for b in range(3):
        assert torch.all(btf[b, int(frames[b]):] == 0)


# ----------------------------------------------------------------------------------------- on-device edit distance
@pytest.mark.parametrize("mode", ["cer", "wer"])
def test_edit_distance_batch_matches_host_levenshtein(dev, mode):
    """lasr_edit_distance_batch (one wave per utterance, prefix-min rows) against the oracle's plain DP
    (utils/asr_metrics.py:26-59): token units and str.split() word units, empty / identical / disjoint / long cases."""
    from lightning_asr_amd import ops
    rng = random.Random(5)
    V, space = 28, 0
    B, T, S = 24, 700, 300
    hyp = torch.full((B, T), -1, dtype=torch.int32)
    ref = torch.zeros(B, S, dtype=torch.int64)
    nh = torch.zeros(B, dtype=torch.int32)
    nr = torch.zeros(B, dtype=torch.int32)
    seqs = []
    for b in range(B):
        lr = [0, 1, 5, 64, 65, 130, 299, 300][b % 8] if b < 16 else rng.randint(0, S)
        r = [rng.randint(0, V - 1) for _ in range(lr)]
        if b == 3:
            h = list(r)                                   # identical
        elif b == 4:
            h = []                                        # empty hypothesis
        elif b % 3 == 0:
            h = [x for x in r if rng.random() > 0.2]      # deletions
            h = [x if rng.random() > 0.1 else rng.randint(0, V - 1) for x in h]
        else:
            h = [rng.randint(0, V - 1) for _ in range(rng.randint(0, T))]
        if mode == "wer" and b == 5:
            r = [space, space, 3, 4, space, space, 5, space]      # leading / repeated / trailing spaces
            h = [3, 4, space, 5]
        hyp[b, :len(h)] = torch.tensor(h, dtype=torch.int32) if h else hyp[b, :0]
        ref[b, :len(r)] = torch.tensor(r, dtype=torch.int64) if r else ref[b, :0]
        nh[b], nr[b] = len(h), len(r)
        seqs.append((h, r))
    totals = torch.tensor([7, 11], dtype=torch.int64, device=dev)
    dist, units = ops.edit_distance_batch(hyp.to(dev), nh.to(dev), ref.to(dev), nr.to(dev), space if mode == "wer" else -1, totals)

    def words(x):
        out, cur = [], []
        for t in x:
            if t == space:
                if cur:
                    out.append(tuple(cur))
                cur = []
            else:
                cur.append(t)
        if cur:
            out.append(tuple(cur))
        return out
    exp_d, exp_u = [], []
    for h, r in seqs:
        a, b_ = (h, r) if mode == "cer" else (words(h), words(r))
        exp_d.append(R.levenshtein(a, b_))
        exp_u.append(len(b_))
    assert dist.cpu().tolist() == exp_d
    assert units.cpu().tolist() == exp_u
    assert totals.cpu().tolist() == [7 + sum(exp_d), 11 + sum(exp_u)]


def test_wer_metric_device_path_equals_string_path(dev):
    """WER.update on the device (greedy collapse + edit distance) gives the reference's scores/words (utils/asr_metrics.py:187-228)"""
    from lightning_asr_amd.utils.asr_metrics import WER, word_error_rate
    labels_cer = [c.strip() for c in open("data/labels.txt").readlines()]
    labels_wer = [" ", "'"] + [chr(ord("a") + i) for i in range(26)]
    g = torch.Generator().manual_seed(2)
    for labels, use_cer in ((labels_cer, True), (labels_wer, False)):
        V = len(labels)
        B, T, S = 6, 120, 30
        pred = torch.randint(0, V + 1, (B, T), generator=g)
        pred[:, ::3] = V                                    # blanks in between
        tg = torch.randint(0, V, (B, S), generator=g)
        tl = torch.tensor([30, 12, 1, 25, 7, 30], dtype=torch.int32)
        t_len = torch.tensor([120, 100, 5, 64, 65, 119], dtype=torch.int32)
        m = WER(labels, use_cer=use_cer)
        assert m.device_ok
        val = m(pred.to(dev), tg.to(dev), tl.to(dev), t_len.to(dev))
        hyps = m.ctc_decoder_predictions_tensor(pred.to(dev), t_len.to(dev))
        refs = m.decode_reference(tg, tl)
        assert float(val) == pytest.approx(word_error_rate(hyps, refs, use_cer=use_cer), rel=1e-6)
        assert m.scores.is_cuda and m.words.is_cuda


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,dtype,has_res", [(32, 501, 512, torch.bfloat16, True), (5, 77, 256, torch.float32, True),
                                                 (3, 130, 64, torch.float32, False), (9, 40, 96, torch.bfloat16, True),
                                                 (70, 33, 512, torch.float32, True)])
def test_se_layer_fwd_bwd_batched(dev, B, T, C, dtype, has_res):
    """SELayer (models/QuartNetContextSE.py:8-23) around a BN output: out = relu(BN(y) * s + BN2(y2)), s = sigmoid(W2 relu(W1 mean_T BN(y))).
    The excite MLP runs for the whole batch in two launches per direction; checked against f64 autograd of the same formula
    (forward values and the three SE gradients: through-the-mean term seg, dW1, dW2)."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(B * 7 + C)
    H = C // 8
    y = torch.randn(B, T, C, generator=g).to(dtype)
    y2 = torch.randn(B, T, C, generator=g).to(dtype) if has_res else None
    a = 0.5 + torch.rand(C, generator=g); b = 0.2 * torch.randn(C, generator=g)
    a2 = 0.5 + torch.rand(C, generator=g); b2 = 0.2 * torch.randn(C, generator=g)
    W1 = torch.randn(H, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, H, generator=g) / H ** 0.5
    dout = torch.randn(B, T, C, generator=g).to(dtype)
    coef = torch.stack([a, b]).contiguous()
    coef2 = torch.stack([a2, b2]).contiguous() if has_res else None
    # reference in f64 on the values as stored
    yd = y.double(); W1d = W1.double().requires_grad_(True); W2d = W2.double().requires_grad_(True)
    z1 = (yd * a.double() + b.double()).requires_grad_(True)
    pooled = z1.mean(1)
    hid = torch.relu(pooled @ W1d.t())
    s = torch.sigmoid(hid @ W2d.t())
    z = z1 * s[:, None, :] + ((y2.double() * a2.double() + b2.double()) if has_res else 0.0)
    out = torch.relu(z)
    s.retain_grad()
    out.backward(dout.double())
    # d(loss)/d(z1) = dout*relu'(z)*s (direct) + seg (through the mean): seg = z1.grad - direct term
    direct = dout.double() * (z > 0) * s.detach()[:, None, :]
    seg_ref = (z1.grad - direct)[:, 0, :]
    ysum, pooled_g, hid_g, s_g = ops.se_fwd(y.to(dev), coef.to(dev), W1.to(dev), W2.to(dev))
    assert torch.allclose(ysum.cpu().double(), yd.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(pooled_g.cpu().double(), pooled.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(hid_g.cpu().double(), hid.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(s_g.cpu().double(), s.detach(), rtol=1e-5, atol=1e-6)
    seg, dW1, dW2 = ops.se_bwd(dout.to(dev), y.to(dev), coef.to(dev), s_g, hid_g, pooled_g, W1.to(dev), W2.to(dev),
                               y2.to(dev) if has_res else None, coef2.to(dev) if has_res else None, "relu")

    def rel(x, r):
        return ((x.cpu().double() - r).norm() / (r.norm() + 1e-30)).item()
    assert rel(seg, seg_ref) < 2e-5, rel(seg, seg_ref)
    assert rel(dW1, W1d.grad) < 2e-5, rel(dW1, W1d.grad)
    assert rel(dW2, W2d.grad) < 2e-5, rel(dW2, W2d.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,has_res,masked", [(32, 501, 512, True, True), (16, 300, 256, True, False), (8, 801, 1024, False, False),
                                                  (5, 1000, 64, True, True)])
def test_bn_bwd_channel_sliced_pair_equals_row_major_pair(dev, B, T, C, has_res, masked):
    """The channel-sliced BN backward (workgroup = 64 channels x a chunk of rows, constants folded in the apply pass's prologue: no
    table launch) against the row-major pair with its reduced sums, on the same bf16 tensors: same arithmetic per element, the
    channel sums only differ in summation order.  Shapes with >= 4096 rows take the sliced kernels (sums = NULL path)."""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(B + T + C)
    N = B * T
    assert N >= 4096
    y = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    y2 = torch.randn(B, T, C, generator=g).bfloat16().to(dev) if has_res else None
    dout = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    lens = torch.randint(T // 3, T + 1, (B,), generator=g).int().to(dev) if masked else None
    gam = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    bet = (0.1 * torch.randn(C, generator=g)).to(dev)
    gam2 = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    bet2 = (0.1 * torch.randn(C, generator=g)).to(dev)

    def stats_of(t):
        f = t.reshape(N, C).double()
        return torch.cat([f.sum(0), (f * f).sum(0)]).float()
    coef, saved = ops.bn_finalize(stats_of(y), gam, bet, torch.zeros(C, device=dev), torch.ones(C, device=dev), N)
    coef2 = saved2 = None
    if has_res:
        coef2, saved2 = ops.bn_finalize(stats_of(y2), gam2, bet2, torch.zeros(C, device=dev), torch.ones(C, device=dev), N)
    a = ops.bn_act_bwd(dout, y, coef, saved, gam, y2, coef2, saved2, gam2 if has_res else None, row_lens=lens, act="relu", fused=False)
    b = ops.bn_act_bwd(dout, y, coef, saved, gam, y2, coef2, saved2, gam2 if has_res else None, row_lens=lens, act="relu", fused=True)
    names = ["dy", "dy2", "dgamma", "dbeta", "dgamma2", "dbeta2"]
    for name, ta, tb in zip(names, a, b):
        if ta is None:
            assert tb is None
            continue
        r = rel_l2(tb.float(), ta.float().cpu())
        if name.startswith("dy"):
            # bf16 outputs: equal except where the constants' last bits move a value across a rounding boundary
            assert r < 3e-4, (name, r)
            if masked and name == "dy":
                keep = (torch.arange(T, device=dev).view(1, T) < lens.view(B, 1))
                assert (tb[~keep] == 0).all()
        else:
            assert r < 2e-6, (name, r)
